"""Randomised parity soak (GPU box): production kernels against the reference-order kernels, bit for bit, on random particle clouds
with random grid geometry (cell size != h, anisotropic, origins far from the particles), optional random wall sheets; every 7th
seed in fp64, every 3rd with the Monaghan kernels, every 5th IISPH; every 13th on a grid one or two cells wide in x (no quantised
scan), every 17th with the grid origin more than 4096 cells away (beyond the quanta's error budget), every 19th with NaN / inf
coordinates.
usage: python tools/fuzz_parity.py [seeds=100] [first=0] [oracle]   (oracle: compare with the CPU oracle instead: keys bit-exact, floats
within the parity tolerances)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nereus_amd import capi
from nereus_amd.params import default_params

def make_scene(seed):
    rng = np.random.default_rng(seed)
    solver = capi.IISPH if seed % 5 == 4 else capi.SESPH
    double, kset = (seed % 7 == 6), (0 if seed % 3 == 2 else 1)
    p = default_params(1 if solver == capi.IISPH else 0, double=double).copy()
    real = np.float64 if double else np.float32
    h = float(p["interactionRadius"][0])
    n = int(rng.integers(500, 30000))
    if seed % 11 == 10:  # large enough for the coherent re-sort (merge path) of the production steps
        n = int(rng.integers(40000, 150000))
    ext = rng.uniform(4, 30, 3) * h * np.array([1.0, rng.uniform(0.3, 1.0), rng.uniform(0.3, 1.0)])
    centre = rng.uniform(-0.5, 0.5, 3)
    pos = np.ones((n, 4), real)
    dens_mode = rng.integers(0, 3)
    if dens_mode == 0:
        pos[:, :3] = centre + rng.uniform(-0.5, 0.5, (n, 3)) * ext
    elif dens_mode == 1:   # clumps
        c = centre + rng.uniform(-0.5, 0.5, (8, 3)) * ext
        pos[:, :3] = c[rng.integers(0, 8, n)] + rng.normal(0, 1.2 * h, (n, 3))
    else:                  # jittered lattice
        m = int(round(n ** (1 / 3))) + 1
        g = np.stack(np.meshgrid(*[np.arange(m)] * 3, indexing="ij"), -1).reshape(-1, 3)[:n]
        pos[:, :3] = centre + (g - m / 2) * (0.85 * h) + rng.normal(0, 0.03 * h, (n, 3))
    vel = np.zeros_like(pos); vel[:, :3] = rng.normal(0, 0.5, (n, 3))
    # grid geometry
    cs = h * np.array([rng.choice([1.0, 1.0, 1.25, 0.62]), rng.choice([1.0, 1.0, 1.4]), rng.choice([1.0, 1.0, 0.8])])
    p["cellSize"][0] = cs.astype(real)
    lo = pos[:, :3].min(0) - rng.uniform(0.05, 40.0) * h
    far = seed % 17 == 16   # the cloud sits > 4096 cells from the grid origin on one axis: outside the quantised scan's error budget
    if far:                 # (QP_FAR) — owners are diverted to the wall workgroups' exact scan / the reference-order walk
        lo[int(rng.integers(0, 3))] -= rng.uniform(4200, 9000) * cs.max()
    p["worldOrigin"][0] = lo.astype(real)
    gs = [int(2 ** np.ceil(np.log2(max(4, (pos[:, a].max() - lo[a]) / cs[a] + 2)))) for a in range(3)]
    if rng.random() < 0.3: gs[int(rng.integers(0, 3))] //= 2      # particles beyond the grid: wrap
    gs = [min(max(g, 4 if a == 0 else 1), 1024) for a, g in enumerate(gs)]
    if seed % 13 == 12: gs[0] = int(rng.choice([1, 2]))          # x grids narrower than 4 cells: no quantised scan, no hit lists (qOk false)
    while gs[0] * gs[1] * gs[2] > 2 ** 27: gs[int(np.argmax(gs))] //= 2
    p["gridSize"][0] = gs; p["numCells"][0] = gs[0] * gs[1] * gs[2]
    bi = vbi = None
    if rng.random() < 0.6:
        nb = int(rng.integers(200, 6000))
        bi = np.ones((nb, 4), real)
        bi[:, :3] = centre + rng.uniform(-0.5, 0.5, (nb, 3)) * ext
        bi[:, int(rng.integers(0, 3))] = real(pos[:, :3].min() + rng.uniform(0, 3) * h)   # a sheet
        vbi = rng.uniform(1e-5, 4e-5, nb).astype(real)
    if seed % 19 == 18:   # a few particles with NaN / inf coordinates (a caller's bug must not take the device down)
        k = rng.integers(0, n, 5)
        pos[k[:3], int(rng.integers(0, 3))] = np.nan
        pos[k[3:], int(rng.integers(0, 3))] = np.inf
    return dict(p=p, n=n, pos=pos, vel=vel, bi=bi, vbi=vbi, solver=solver, double=double, kset=kset, gs=gs, cs=cs, h=h)


def one(seed):
    sc = make_scene(seed)
    p, n, pos, vel, bi, vbi, solver, double, kset, gs, cs, h = (sc[k] for k in ("p", "n", "pos", "vel", "bi", "vbi", "solver", "double", "kset", "gs", "cs", "h"))
    outs, iters = [], []
    for ref in (False, True):
        s = capi.Solver(p, n, solver=solver, double=double, kernel_set=kset, reference_order=ref)
        s.set_particles(pos, vel)
        s.set_boundaries(bi, vbi, update_grid=False)
        s.step_partial(capi.STAGE_I_PFORCE if solver == capi.IISPH else capi.STAGE_FORCES)
        o = [s.get("dens"), s.get("forcesP") if solver == capi.IISPH else s.get("forces")]
        iters.append(s.last_iterations if solver == capi.IISPH else 0)
        s.set_particles(pos, vel)
        s.step(6 if n >= 40000 else 3)
        o += list(s.download())
        outs.append(o)
        s.close()
    # A solve that has overflowed (inf pressures in a random clump that does not converge) is not comparable: the list kernels visit
    # only neighbours inside the kernel support, the reference order also multiplies the zero gradients beyond it with the inf
    # (0 * inf = NaN) — equal for finite operands only.
    # (The solver's max(p, 0) clamp can turn such a NaN into a finite 0, so even finite results may then differ: a solve that ran into
    # its iteration cap is skipped as well.)
    if os.environ.get("FUZZ_SKIP_NONFINITE", "0") == "1" and not all(np.isfinite(x).all() for x in outs[1]):
        return "diverged"
    for k, (a, b) in enumerate(zip(*outs)):
        if not np.array_equal(a, b, equal_nan=True):
            bad = np.argwhere(~((a == b) | (np.isnan(a) & np.isnan(b))))
            return "seed %d: array %d differs at %d places, first %s (n=%d grid=%s cs/h=%s solver=%d walls=%s double=%s kset=%d)" % (seed, k, len(bad), bad[0], n, gs, cs / h, solver, bi is not None, double, kset)
    return None

def one_vs_oracle(seed):
    """production kernels against the CPU oracle on the same random scene: keys bit-exact, floats within the parity tolerances"""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tests.oracle_lib import IISPH as O_IISPH, SESPH as O_SESPH, STOP_FORCES, STOP_I_PFORCE, Oracle
    sc = make_scene(seed)
    iis = sc["solver"] == capi.IISPH
    o = Oracle(sc["p"], sc["double"], sc["kset"], O_IISPH if iis else O_SESPH)
    o.set_particles(sc["pos"], sc["vel"]); o.set_boundaries(sc["bi"], sc["vbi"], update_grid=False)
    s = capi.Solver(sc["p"], sc["n"], solver=sc["solver"], double=sc["double"], kernel_set=sc["kset"])
    s.set_particles(sc["pos"], sc["vel"]); s.set_boundaries(sc["bi"], sc["vbi"], update_grid=False)
    o.step(1, stop=STOP_I_PFORCE if iis else STOP_FORCES); s.step_partial(capi.STAGE_I_PFORCE if iis else capi.STAGE_FORCES)
    def rel(x, y):
        x = np.asarray(x, np.float64); y = np.asarray(y, np.float64)
        sc_ = np.abs(y).max()
        return float(np.abs(x - y).max() / sc_) if sc_ > 0 else float(np.abs(x - y).max())
    fo = o.get("forcesP" if iis else "forces")
    # (IISPH on a random clump may not converge: after dozens of iterations intermediate pressures overflow, see the note in one())
    if not (np.isfinite(fo).all() and np.isfinite(o.get("dens")).all()):
        s.close(); return "diverged"
    msg = None
    if not np.array_equal(s.get("hash"), o.get("hash")) or not np.array_equal(s.get("index"), o.get("index")):
        msg = "hash/index differ"
    elif rel(s.get("dens"), o.get("dens")) > 2e-6:
        msg = "dens rel %.2e" % rel(s.get("dens"), o.get("dens"))
    elif rel(s.get("forcesP" if iis else "forces"), fo) > (1e-4 if iis else 2e-5):
        msg = "forces rel %.2e" % rel(s.get("forcesP" if iis else "forces"), fo)
    s.close()
    return None if msg is None else "seed %d vs oracle: %s (n=%d grid=%s solver=%d double=%s kset=%d walls=%s)" % (
        seed, msg, sc["n"], sc["gs"], sc["solver"], sc["double"], sc["kset"], sc["bi"] is not None)


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[3] == "oracle":
        one = one_vs_oracle
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    fails = div = 0
    for sd in range(first, first + seeds):
        r = one(sd)
        if r == "diverged": div += 1
        elif r: print(r); fails += 1
        if (sd - first) % 25 == 24: print("... %d seeds done, %d failures" % (sd - first + 1, fails), flush=True)
    print("fuzz: %d seeds, %d failures, %d skipped (reference result not finite)" % (seeds, fails, div))
    sys.exit(1 if fails else 0)
