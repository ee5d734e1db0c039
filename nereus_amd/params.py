"""SphSimParams as a numpy structured dtype.

Field order and sizes follow the reference POD (common/sph_kernel.cuh:13-59): 132 B for
SReal=float, 240 B for SReal=double (offsets listed in SURVEY.md §8 a1).  The same bytes are
what `nrs_params_f32` / `nrs_params_f64` in include/nereus_hip.h describe.
"""
import numpy as np


def params_dtype(double: bool = False) -> np.dtype:
    r = np.float64 if double else np.float32
    dt = np.dtype(
        [
            ("gridSize", np.uint32, 3),
            ("numCells", np.uint32),
            ("worldOrigin", r, 3),
            ("cellSize", r, 3),
            ("numBodies", np.uint32),
            ("maxParticlesPerCell", np.uint32),
            ("gasStiffness", r),
            ("viscosity", r),
            ("surfaceTension", r),
            ("restDensity", r),
            ("particleMass", r),
            ("interactionRadius", r),
            ("timestep", r),
            ("particleRadius", r),
            ("gravity", r, 3),
            ("soundSpeed", r),
            ("beta", r),
            ("kpoly", r),
            ("kpoly_grad", r),
            ("kpress_grad", r),
            ("kvisc_grad", r),
            ("kvisc_denum", r),
            ("ksurf1", r),
            ("ksurf2", r),
            ("bpol", r),
        ],
        align=True,
    )
    assert dt.itemsize == (240 if double else 132), dt.itemsize
    return dt


def new_params(double: bool = False) -> np.ndarray:
    """A zeroed 1-element params record (use p[0] / p['field'][0])."""
    return np.zeros(1, dtype=params_dtype(double))


def _powf(x, y):
    """powf(x, y): float in, float out (evaluated in double and rounded once)."""
    return np.float32(np.power(np.float64(np.float32(x)), np.float64(y)))


def kernel_constants(p, ctor):
    """Pre-computed smoothing-kernel parts.  ctor 0: SPH::SPH() (sph/sph.cpp:76-90, (SReal)M_PI);
    ctor 1: IISPH::IISPH() (sph/iisph/iisph.cpp:70-80, double M_PI); ctor 2: the SphSimParams constructors
    (sph.cpp:105-116)."""
    R = p.dtype["kpoly"].type
    ir = p["interactionRadius"][0]
    f64 = np.float64
    pi_r = R(np.pi)
    if ctor != 1:
        p["kpoly"][0] = R(315.0 / (64.0 * f64(pi_r) * f64(_powf(ir, 9.0))))
        p["kpoly_grad"][0] = R(-945.0 / (32.0 * f64(pi_r) * f64(_powf(ir, 9.0))))
        p["kpress_grad"][0] = R(-45.0 / f64(R(pi_r * R(_powf(ir, 6.0)))))
        if ctor == 0:
            p["kvisc_grad"][0] = R(15.0 / (2.0 * f64(pi_r) * f64(_powf(ir, 3.0))))
        else:
            p["kvisc_grad"][0] = R(15.0 / f64(R(R(R(2) * pi_r) * R(_powf(ir, 3.0)))))
        p["ksurf1"][0] = R(32.0 / f64(R(pi_r * R(_powf(ir, 9.0)))))
    else:
        p["kpoly"][0] = R(315.0 / (64.0 * np.pi * f64(_powf(ir, 9.0))))
        p["kpoly_grad"][0] = R(-945.0 / (32.0 * np.pi * f64(_powf(ir, 9.0))))
        p["kpress_grad"][0] = R(-45.0 / (np.pi * f64(_powf(ir, 6.0))))
        p["kvisc_grad"][0] = R(15.0 / (2 * np.pi * f64(_powf(ir, 3.0))))
        p["ksurf1"][0] = R(32.0 / (np.pi * f64(_powf(ir, 9))))
    p["kvisc_denum"][0] = R(2.0 * f64(_powf(ir, 3.0)))
    p["ksurf2"][0] = R(f64(_powf(ir, 6)) / 64.0)
    p["bpol"][0] = R(np.float32(0.007) / _powf(ir, 3.25))
    return p


def default_params(solver: int = 0, double: bool = False) -> np.ndarray:
    """Constructor defaults of Nereus::SPH (solver 0, sph/sph.cpp:29-93) / Nereus::IISPH (solver 1,
    sph/iisph/iisph.cpp:28-87), evaluated with the reference's float/double mix."""
    p = new_params(double)
    R = p.dtype["kpoly"].type
    f64 = np.float64
    p["gasStiffness"][0] = 800
    p["restDensity"][0] = 1000
    p["particleRadius"][0] = R(0.02)
    p["timestep"][0] = R(1e-3)
    p["surfaceTension"][0] = R(0.01)
    eta, H = R(0.01), R(0.1)
    vf = R(np.sqrt(2.0 * 9.81 * f64(H)))
    p["soundSpeed"][0] = vf / R(np.sqrt(eta))
    if solver == 0:
        p["viscosity"][0] = R(0.005)
        p["gravity"][0] = (R(0.0), R(-9.81), R(0.0))
        ir = R(0.0457)
        p["beta"][0] = 450.0
        origin, grid = R(-1.1), 64
    else:
        p["viscosity"][0] = R(0.01)
        p["gravity"][0] = (R(0.0), R(np.float32(-9.81)), R(0.0))
        ir = R(0.0537)
        p["beta"][0] = 1050.0
        origin, grid = R(-1.2), 128
    p["interactionRadius"][0] = ir
    p["particleMass"][0] = R(0.5 * f64(_powf(ir, 3)) * f64(p["restDensity"][0]))
    p["worldOrigin"][0] = (origin, origin, origin)
    p["gridSize"][0] = (grid, grid, grid)
    p["cellSize"][0] = (ir, ir, ir)
    p["numCells"][0] = grid ** 3
    return kernel_constants(p, 0 if solver == 0 else 1)


def update_grid(p: np.ndarray, bbmin, bbmax) -> np.ndarray:
    """SPH::updateGrid (sph/sph.cpp:313-337) applied to a known boundary AABB: origin = min - 0.1,
    gridSize = nextPow2(ceil((extent + 0.1) / h)) per axis.  Used by the slab driver so that every rank has the
    same GLOBAL grid without seeing all boundary particles."""
    R = p.dtype["kpoly"].type
    h = np.float64(p["interactionRadius"][0])
    grid = []
    for a in range(3):
        lo, hi = R(bbmin[a]), R(bbmax[a])
        p["worldOrigin"][0][a] = R(np.float64(lo) - 0.1)
        size = int(np.ceil((np.float64(R(hi - lo)) + 0.1) / h))
        v = max(size, 1) - 1
        for s in (1, 2, 4, 8, 16):
            v |= v >> s
        grid.append(v + 1)
    p["gridSize"][0] = grid
    p["numCells"][0] = grid[0] * grid[1] * grid[2]
    return p
