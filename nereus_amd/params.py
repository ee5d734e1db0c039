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
