"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Bars (BASELINE.json north_star): hash / index / cell tables bit-exact; fp32 state within 1e-5 relative
after N steps.  Per-stage float arrays are compared much tighter than that (see TOL_* below): the HIP
kernels form every sum in the reference's order with IEEE arithmetic, so differences can only come from
the last-bit behaviour of pow()/powf().
"""
import os

import numpy as np
import pytest

from nereus_amd import capi, scene
from tests.common import check_cell_tables, compressed_block, default_scene, rel_err, small_dam_break
from tests.oracle_lib import (IISPH, SESPH, STOP_DENSITY, STOP_FORCES, STOP_HASH, STOP_I_ADVECTION,
                              STOP_I_DISPLACEMENT, STOP_I_PFORCE, STOP_I_SOLVE, STOP_REORDER, STOP_SORT, Oracle)

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL_STAGE = 2e-6   # per-stage float arrays (density, pressure, forces, IISPH intermediates), array-relative
TOL_STEPS = 1e-5   # positions / velocities after N steps (the north_star bar)


def max_ulp(a, b):
    """largest distance in units in the last place between two float32 arrays"""
    a = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return int(np.abs(a - b).max()) if a.size else 0


def make_pair(p, pos, vel, bi=None, vbi=None, solver=SESPH, double=False, kset=1, ref=False, pres=None,
              capacity=None, surf=True, tait="powf", flags=0, threads=1):
    o = Oracle(p, double, kset, solver, threads=threads, surface_tension=surf, tait=tait)
    o.set_particles(pos, vel, pres)
    o.set_boundaries(bi, vbi, update_grid=True)
    s = capi.Solver(p, capacity or max(len(pos), 1), solver=solver, double=double, kernel_set=kset,
                    reference_order=ref, surface_tension=surf, flags=flags)
    s.set_particles(pos, vel, pres)
    s.set_boundaries(bi, vbi, update_grid=True)
    return o, s


@pytest.mark.parametrize("ref", [False, True], ids=["tiled", "reforder"])
def test_sesph_stages_default_scene(hip_lib, ref):
    p, pos, vel = default_scene(SESPH)
    o, s = make_pair(p, pos, vel, ref=ref)
    # hash (unsorted)
    o.step(1, stop=STOP_HASH); s.step_partial(capi.STAGE_HASH)
    np.testing.assert_array_equal(s.get("hash"), o.get("hash"))
    np.testing.assert_array_equal(s.get("index"), np.arange(len(pos), dtype=np.uint32))
    # sort
    s.set_particles(pos, vel)
    o.step(1, stop=STOP_SORT); s.step_partial(capi.STAGE_SORT)
    np.testing.assert_array_equal(s.get("hash"), o.get("hash"))
    np.testing.assert_array_equal(s.get("index"), o.get("index"))
    # reorder + cell tables
    s.set_particles(pos, vel)
    o.step(1, stop=STOP_REORDER); s.step_partial(capi.STAGE_REORDER)
    check_cell_tables(s.get("cellStart"), s.get("cellEnd"), o.get("cellStart"), o.get("cellEnd"))
    np.testing.assert_array_equal(s.get("sortedPos"), o.get("sortedPos"))
    np.testing.assert_array_equal(s.get("sortedVel"), o.get("sortedVel"))
    # density / pressure
    s.set_particles(pos, vel)
    o.step(1, stop=STOP_DENSITY); s.step_partial(capi.STAGE_DENSITY)
    assert rel_err(s.get("dens"), o.get("dens")) <= TOL_STAGE
    assert rel_err(s.get("pres"), o.get("pres")) <= TOL_STAGE
    # forces
    s.set_particles(pos, vel)
    o.step(1, stop=STOP_FORCES); s.step_partial(capi.STAGE_FORCES)
    assert rel_err(s.get("forces"), o.get("forces")) <= TOL_STAGE
    # stronger than the bar: every sum is formed in the reference's order with IEEE arithmetic, so density and
    # forces are BIT-EXACT against the oracle; the Tait pressure may differ by 1 ulp where glibc powf(x,7) is not
    # correctly rounded (x^7 is evaluated in double and rounded once on the device)
    np.testing.assert_array_equal(s.get("dens"), o.get("dens"))
    np.testing.assert_array_equal(s.get("forces"), o.get("forces"))
    assert max_ulp(s.get("pres"), o.get("pres")) <= 1
    # and against the committed fixture
    g = np.load(os.path.join(GOLD, "sesph_default.npz"))
    np.testing.assert_array_equal(s.get("hash"), g["hash"])
    np.testing.assert_array_equal(s.get("index"), g["index"])
    assert rel_err(s.get("dens"), g["dens"]) <= TOL_STAGE
    assert rel_err(s.get("forces"), g["forces"]) <= TOL_STAGE


@pytest.mark.parametrize("ref", [False, True], ids=["tiled", "reforder"])
def test_sesph_stages_dam_break_with_boundaries(hip_lib, ref):
    p, sc = small_dam_break()
    o, s = make_pair(p, sc["pos"], sc["vel"], sc["bi"], sc["vbi"], ref=ref)
    # updateGrid rule + boundary tables
    np.testing.assert_array_equal(s.params.view(np.uint8), o.params.view(np.uint8))
    np.testing.assert_array_equal(s.get("bhash"), o.get("bhash"))
    np.testing.assert_array_equal(s.get("bindex"), o.get("bindex"))
    check_cell_tables(s.get("bCellStart"), s.get("bCellEnd"), o.get("bCellStart"), o.get("bCellEnd"))
    bs = s.get("bSorted")
    np.testing.assert_array_equal(bs[:, :3], o.get("sbi")[:, :3])
    np.testing.assert_array_equal(bs[:, 3], o.get("svbi"))
    o.step(1, stop=STOP_FORCES); s.step_partial(capi.STAGE_FORCES)
    np.testing.assert_array_equal(s.get("hash"), o.get("hash"))
    np.testing.assert_array_equal(s.get("index"), o.get("index"))
    check_cell_tables(s.get("cellStart"), s.get("cellEnd"), o.get("cellStart"), o.get("cellEnd"))
    assert rel_err(s.get("dens"), o.get("dens")) <= TOL_STAGE
    assert rel_err(s.get("pres"), o.get("pres")) <= TOL_STAGE
    assert rel_err(s.get("forces"), o.get("forces")) <= TOL_STAGE
    np.testing.assert_array_equal(s.get("dens"), o.get("dens"))   # bit-exact, boundary terms included
    assert max_ulp(s.get("pres"), o.get("pres")) <= 1
    # forces: bit-exact wherever the pressure is (a 1-ulp pressure difference propagates to its neighbours)
    same_p = s.get("pres") == o.get("pres")
    if same_p.all():
        np.testing.assert_array_equal(s.get("forces"), o.get("forces"))


@pytest.mark.parametrize("ref", [False, True], ids=["tiled", "reforder"])
@pytest.mark.parametrize("scene_name", ["default", "dambreak"])
def test_sesph_n_steps(hip_lib, ref, scene_name):
    if scene_name == "default":
        p, pos, vel = default_scene(SESPH)
        bi = vbi = None
        gold, steps = np.load(os.path.join(GOLD, "sesph_default.npz")), 10
    else:
        p, sc = small_dam_break()
        pos, vel, bi, vbi = sc["pos"], sc["vel"], sc["bi"], sc["vbi"]
        gold, steps = np.load(os.path.join(GOLD, "sesph_dambreak.npz")), 10
    o, s = make_pair(p, pos, vel, bi, vbi, ref=ref)
    o.step(steps); s.step(steps)
    gp, gv = s.download()
    assert rel_err(gp[:, :3], o.get("pos")[:, :3]) <= TOL_STEPS
    assert rel_err(gv[:, :3], o.get("vel")[:, :3]) <= TOL_STEPS
    np.testing.assert_array_equal(gp[:, 3], o.get("pos")[:, 3])
    assert rel_err(gp[:, :3], gold["pos%d" % steps][:, :3]) <= TOL_STEPS
    assert rel_err(gv[:, :3], gold["vel%d" % steps][:, :3]) <= TOL_STEPS
    # the sorted hash / index of the last step are bit-exact too (order of the output arrays)
    np.testing.assert_array_equal(s.get("hash"), o.get("hash"))
    np.testing.assert_array_equal(s.get("index"), o.get("index"))
    # 40 more steps (50 in total)
    o.step(40); s.step(40)
    gp, gv = s.download()
    assert rel_err(gp[:, :3], o.get("pos")[:, :3]) <= TOL_STEPS
    assert rel_err(gv[:, :3], o.get("vel")[:, :3]) <= 5 * TOL_STEPS


def test_tiled_equals_reference_order_bitwise(hip_lib):
    """Two different kernels (LDS hit-list vs plain 27-cell walk) must agree bit for bit."""
    p, sc = small_dam_break((20, 16, 14))
    outs = []
    for ref in (False, True):
        s = capi.Solver(p, len(sc["pos"]), reference_order=ref)
        s.set_particles(sc["pos"], sc["vel"])
        s.set_boundaries(sc["bi"], sc["vbi"], update_grid=True)
        s.step_partial(capi.STAGE_FORCES)
        outs.append((s.get("dens"), s.get("pres"), s.get("forces")))
        s.set_particles(sc["pos"], sc["vel"])
        s.step(5)
        outs[-1] += s.download()
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("ref", [False, True], ids=["tiled", "reforder"])
@pytest.mark.parametrize("solver", [SESPH, IISPH], ids=["sesph", "iisph"])
def test_surface_tension_off(hip_lib, ref, solver):
    """USE_SURFACE_TENSION=0 (CMakeLists.txt:28; sph_kernel_impl.cuh:535-548 drops the cohesion term): the library's
    SURF=false kernel instances, both kernel families, against the oracle with the same switch — the force evaluation
    (SESPH) / displacement factors (IISPH, whose predictAdvection calls the same cell loop), then 10 steps; and the
    result differs from the surface-tension build's (the switch is live)."""
    if solver == SESPH:
        p, sc = small_dam_break((14, 12, 10))
        pos, vel, bi, vbi, pres = sc["pos"], sc["vel"], sc["bi"], sc["vbi"], None
    else:
        p, pos, vel = compressed_block((11, 10, 9))
        bi = vbi = pres = None
    o, s = make_pair(p, pos, vel, bi, vbi, solver=solver, ref=ref, surf=False)
    if solver == SESPH:
        o.step(1, stop=STOP_FORCES); s.step_partial(capi.STAGE_FORCES)
        np.testing.assert_array_equal(s.get("dens"), o.get("dens"))
        assert rel_err(s.get("forces"), o.get("forces")) <= TOL_STAGE
        if (s.get("pres") == o.get("pres")).all():
            np.testing.assert_array_equal(s.get("forces"), o.get("forces"))
        f_off = s.get("forces")
        o1, s1 = make_pair(p, pos, vel, bi, vbi, solver=solver, ref=ref, surf=True)
        s1.step_partial(capi.STAGE_FORCES)
        assert rel_err(s1.get("forces"), f_off) > 1e-4        # the cohesion term is really gone
        s1.close()
    else:
        o.step(1, stop=STOP_I_DISPLACEMENT); s.step_partial(capi.STAGE_I_DISPLACEMENT)
        assert rel_err(s.get("forcesAdv"), o.get("forcesAdv")) <= TOL_STAGE
        assert rel_err(s.get("diiFluid"), o.get("diiFluid")) <= TOL_STAGE
    o.set_particles(pos, vel, pres); s.set_particles(pos, vel, pres)
    o.step(10); s.step(10)
    gp, gv = s.download()
    np.testing.assert_array_equal(s.get("hash"), o.get("hash"))
    np.testing.assert_array_equal(s.get("index"), o.get("index"))
    assert rel_err(gp[:, :3], o.get("pos")[:, :3]) <= TOL_STEPS
    assert rel_err(gv[:, :3], o.get("vel")[:, :3]) <= TOL_STEPS
    if solver == IISPH:
        assert s.last_iterations == o.last_iters


@pytest.mark.parametrize("flag", ["staged", "nowall"])
def test_alternative_density_launches_small(hip_lib, flag):
    """The two selectable forms of the density launch that are not the default (include/nereus_hip.h):
    NRS_FLAG_STAGED_SCAN — the LDS-staged wave-per-64-slots scan of nrs_kernels_staged.h (the north-star's literal shape:
    row hulls by ballot + readlane, a z-plane of candidates staged in LDS) — and NRS_FLAG_NO_WALL_WORKGROUPS (one kind of
    workgroup).  Small dam-break with walls: stage arrays against the ORACLE, then 10 steps; and bit for bit what the default
    launch produces."""
    fl = capi.FLAG_STAGED_SCAN if flag == "staged" else capi.FLAG_NO_WALL_WORKGROUPS
    p, sc = small_dam_break((20, 16, 14))
    o, s = make_pair(p, sc["pos"], sc["vel"], sc["bi"], sc["vbi"], flags=fl)
    o.step(1, stop=STOP_FORCES); s.step_partial(capi.STAGE_FORCES)
    np.testing.assert_array_equal(s.get("hash"), o.get("hash"))
    np.testing.assert_array_equal(s.get("index"), o.get("index"))
    np.testing.assert_array_equal(s.get("dens"), o.get("dens"))
    assert max_ulp(s.get("pres"), o.get("pres")) <= 1
    assert rel_err(s.get("forces"), o.get("forces")) <= TOL_STAGE
    d = capi.Solver(p, len(sc["pos"]))
    d.set_particles(sc["pos"], sc["vel"])
    d.set_boundaries(sc["bi"], sc["vbi"], update_grid=True)
    d.step_partial(capi.STAGE_FORCES)
    for k in ("dens", "pres", "forces"):
        np.testing.assert_array_equal(s.get(k), d.get(k), err_msg=k)
    for x in (o, s, d):
        x.set_particles(sc["pos"], sc["vel"])
        x.step(10)
    gp, gv = s.download()
    assert rel_err(gp[:, :3], o.get("pos")[:, :3]) <= TOL_STEPS
    assert rel_err(gv[:, :3], o.get("vel")[:, :3]) <= TOL_STEPS
    dp, dv = d.download()
    np.testing.assert_array_equal(gp, dp)
    np.testing.assert_array_equal(gv, dv)
    if flag == "staged":   # the staged kernel really ran: its diagnostics counter exists only on that path
        assert s.get_stat(capi.STAT_UNSTAGED) >= 0


def test_hash_bit_exact_on_adversarial_positions(hip_lib):
    """calcGridPos/calcGridHash (sph_kernel_impl.cuh:105-125): positions exactly on, and one ulp either side of, cell
    faces (where a reciprocal-multiply or a contracted FMA would flip floor()), far outside the grid (wrap, negative
    cells), for several grids and origins — hash must match the oracle bit for bit."""
    rng = np.random.default_rng(2024)
    base = Oracle.default_params(SESPH)
    for grid, origin, h in (((64, 64, 64), (-1.1, -1.1, -1.1), 0.0457), ((128, 32, 256), (0.013, -7.5, 2.25), 0.0537),
                            ((1024, 512, 256), (-0.1, -0.1, -0.1), 0.0457)):
        p = base.copy()
        p["gridSize"][0] = grid
        p["numCells"][0] = int(np.prod(grid))
        p["worldOrigin"][0] = origin
        p["interactionRadius"][0] = h
        p["cellSize"][0] = (h, h, h)
        p = Oracle.recompute_constants(p)
        o32 = np.array(p["worldOrigin"][0], np.float32)
        c32 = np.float32(p["cellSize"][0][0])
        k = rng.integers(-3, max(grid) + 3, size=(20000, 3)).astype(np.float32)
        face = (o32 + k * c32).astype(np.float32)            # nominally on a cell face
        pts = [face, np.nextafter(face, np.float32(np.inf)), np.nextafter(face, np.float32(-np.inf)),
               rng.uniform(-50, 50, (20000, 3)).astype(np.float32),
               (o32 + rng.uniform(0, 1, (20000, 3)).astype(np.float32) * (np.array(grid, np.float32) * c32)).astype(np.float32)]
        pos = np.ones((sum(len(a) for a in pts), 4), np.float32)
        pos[:, :3] = np.concatenate(pts)
        o, s = make_pair(p, pos, np.zeros_like(pos))
        o.step(1, stop=STOP_HASH); s.step_partial(capi.STAGE_HASH)
        np.testing.assert_array_equal(s.get("hash"), o.get("hash"))
        s.set_particles(pos, None)
        o.step(1, stop=STOP_REORDER); s.step_partial(capi.STAGE_REORDER)
        np.testing.assert_array_equal(s.get("hash"), o.get("hash"))
        np.testing.assert_array_equal(s.get("index"), o.get("index"))
        check_cell_tables(s.get("cellStart"), s.get("cellEnd"), o.get("cellStart"), o.get("cellEnd"))


def test_28_bit_grid_sort_paths(hip_lib):
    """A 2^28-cell grid (as the global grid of a 2-4 rank weak-scaling run): hashes of 28 key bits take the 10-bit
    radix configurations of both the full sort and the mover sort of the coherent re-sort.  Hash / index / cell tables
    against the oracle after the first sort, then default path == full-sort path bit for bit over several steps."""
    p, sc = small_dam_break((36, 34, 32))
    n = len(sc["pos"])
    assert n >= 32768
    p = p.copy()
    o0 = Oracle(p, solver=SESPH)
    o0.set_particles(sc["pos"], sc["vel"])
    o0.set_boundaries(sc["bi"], sc["vbi"], update_grid=True)   # the tank's own grid ...
    p = o0.params.copy()
    p["gridSize"][0] = (2048, 512, 256)                         # ... blown up to 2^28 cells around the same origin
    p["numCells"][0] = 2048 * 512 * 256
    o = Oracle(p, solver=SESPH, threads=min(16, os.cpu_count() or 1))
    o.set_particles(sc["pos"], sc["vel"])
    o.set_boundaries(sc["bi"], sc["vbi"], update_grid=False)
    solvers = []
    for flags in (0, capi.FLAG_FULL_SORT):
        s = capi.Solver(p, n, flags=flags)
        s.set_particles(sc["pos"], sc["vel"])
        s.set_boundaries(sc["bi"], sc["vbi"], update_grid=False)
        solvers.append(s)
    assert int(solvers[0].params["numCells"][0]) == 1 << 28
    o.step(3)
    for s in solvers:
        s.step(3)
        np.testing.assert_array_equal(s.get("hash"), o.get("hash"))
        np.testing.assert_array_equal(s.get("index"), o.get("index"))
    # (the cell table of a finished step has been reset; compare it on a partial step)
    o1, s1 = make_pair(p, sc["pos"], sc["vel"])
    o1.step(1, stop=STOP_REORDER); s1.step_partial(capi.STAGE_REORDER)
    check_cell_tables(s1.get("cellStart"), s1.get("cellEnd"), o1.get("cellStart"), o1.get("cellEnd"))
    s1.close()
    for s in solvers:
        s.step(9)
    a, b = [s.download() + (s.get("hash"), s.get("index"), s.get("dens")) for s in solvers]
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    assert solvers[0].resort_stats() == (11, 0)
    for s in solvers:
        s.close()


def test_kernel_path_flags_are_bitwise_equivalent(hip_lib):
    """Fused vs separate force/integrate/hash launches, shared hit lists vs a second scan: same bits."""
    p, sc = small_dam_break((20, 16, 14))
    outs = []
    for flags in (0, capi.FLAG_NO_FUSION, capi.FLAG_NO_SHARED_LISTS, capi.FLAG_NO_FUSION | capi.FLAG_NO_SHARED_LISTS):
        s = capi.Solver(p, len(sc["pos"]), flags=flags)
        s.set_particles(sc["pos"], sc["vel"])
        s.set_boundaries(sc["bi"], sc["vbi"], update_grid=True)
        s.step(7)
        outs.append(s.download() + (s.get("hash"), s.get("index"), s.get("dens")))
        s.close()
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("double", [False, True], ids=["f32", "f64"])
def test_coherent_resort_equals_full_sort(hip_lib, double):
    """Default path (sort only the particles that changed cell, merge into the rest) vs NRS_FLAG_FULL_SORT (the
    reference's sort-everything): identical hash / index / cell tables / state after every checked step, the oracle's
    hash and index bit for bit, and the fall-back when most particles change cell."""
    p, sc = small_dam_break((40, 36, 32), double=double)
    n = len(sc["pos"])
    assert n >= 32768  # below that the context always sorts from scratch
    names = ("hash", "index", "cellStart", "cellEnd", "dens")
    solvers = []
    for flags in (0, capi.FLAG_FULL_SORT):
        s = capi.Solver(p, n, flags=flags, double=double)
        s.set_particles(sc["pos"], sc["vel"])
        s.set_boundaries(sc["bi"], sc["vbi"], update_grid=True)
        solvers.append(s)
    o = Oracle(p, double=double, solver=SESPH, threads=min(16, os.cpu_count() or 1))
    o.set_particles(sc["pos"], sc["vel"])
    o.set_boundaries(sc["bi"], sc["vbi"], update_grid=True)
    done = 0
    for k in (1, 2, 5, 12, 20):
        for s in solvers:
            s.step(k)
        done += k
        a, b = [s.download() + tuple(s.get(x) for x in names) for s in solvers]
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)
        if done <= 8:
            o.step(k)
            np.testing.assert_array_equal(a[2], o.get("hash"))
            np.testing.assert_array_equal(a[3], o.get("index"))
    steps, fallbacks = solvers[0].resort_stats()
    assert steps == done - 1 and fallbacks == 0   # the step after an upload sorts from scratch
    assert solvers[1].resort_stats() == (0, 0)
    # most particles change cell in one step: the count exceeds N/8 and the step sorts from scratch
    h = float(p["interactionRadius"][0])
    dt = float(p["timestep"][0])
    rng = np.random.default_rng(5)
    pos = solvers[0].download()[0]
    vel = np.zeros_like(pos)
    vel[:, :3] = rng.uniform(-1.0, 1.0, (n, 3)).astype(pos.dtype) * (0.9 * h / dt)
    for s in solvers:
        s.set_particles(pos, vel)
        s.step(3)
    a, b = [s.download() + tuple(s.get(x) for x in names) for s in solvers]
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    assert solvers[0].resort_stats()[1] >= 1
    for s in solvers:
        s.close()


def test_iisph_coherent_resort_equals_full_sort(hip_lib):
    """IISPH: the integrate kernel leaves the next step's keys and mover counts; default path vs hash + full sort."""
    p, sc = small_dam_break((36, 34, 32), solver=IISPH)
    n = len(sc["pos"])
    assert n >= 32768
    names = ("hash", "index", "cellStart", "cellEnd", "dens", "P_l", "sumDij")
    outs = []
    for flags in (0, capi.FLAG_FULL_SORT | capi.FLAG_NO_FUSION):
        s = capi.Solver(p, n, solver=IISPH, flags=flags)
        s.set_particles(sc["pos"], sc["vel"])
        s.set_boundaries(sc["bi"], sc["vbi"], update_grid=True)
        s.set_max_iterations(6)
        s.step(3)
        s.step(4)
        outs.append(s.download(pressure=True) + tuple(s.get(x) for x in names) + (s.last_iterations,))
        if flags == 0:
            assert s.resort_stats() == (6, 0)
        s.close()
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)


def test_edge_cases(hip_lib):
    p = Oracle.default_params(SESPH)
    # empty: stepping an empty solver is a no-op
    s = capi.Solver(p, 16)
    s.step(2)
    assert s.n == 0
    # one particle: density is the self term only, it free-falls
    one = np.array([[0.1, 0.2, 0.3, 1.0]], np.float32)
    o, s = make_pair(p, one, np.zeros_like(one), capacity=16)
    o.step(3); s.step(3)
    gp, gv = s.download()
    np.testing.assert_array_equal(gp, o.get("pos"))
    np.testing.assert_array_equal(gv, o.get("vel"))
    # ragged sizes around the 256-thread block, and appending particles between steps (main.cpp:499-513)
    rng = np.random.default_rng(7)
    for n in (255, 256, 257, 1000):
        pos = np.ones((n, 4), np.float32)
        pos[:, :3] = rng.uniform(-0.3, 0.3, (n, 3)).astype(np.float32)
        vel = np.zeros_like(pos)
        vel[:, :3] = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
        o, s = make_pair(p, pos, vel, capacity=2048)
        o.step(2); s.step(2)
        gp, gv = s.download()
        assert rel_err(gp[:, :3], o.get("pos")[:, :3]) <= TOL_STEPS
        extra = np.ones((10, 4), np.float32)
        extra[:, :3] = rng.uniform(-0.3, 0.3, (10, 3)).astype(np.float32)
        s.set_particles(extra, None, first=s.n)
        o.set_particles(np.concatenate([o.get("pos"), extra]), np.concatenate([o.get("vel"), np.zeros_like(extra)]))
        o.step(1); s.step(1)
        assert s.n == n + 10
        gp, gv = s.download()
        assert rel_err(gp[:, :3], o.get("pos")[:, :3]) <= TOL_STEPS
        np.testing.assert_array_equal(s.get("hash"), o.get("hash"))


def test_crowded_cell_overflow_path_and_wrap(hip_lib):
    """>HIT_CAP neighbours per particle (overflow → reference-order path inside the tiled kernels), particles
    outside the grid (hash wraps by & (gridSize-1), negative cells included) and a cell at the x edge."""
    p = Oracle.default_params(SESPH)
    rng = np.random.default_rng(11)
    h = float(p["interactionRadius"][0])
    blob = np.ones((150, 4), np.float32)
    blob[:, :3] = (np.array([0.2, 0.1, -0.3]) + rng.uniform(-0.4 * h, 0.4 * h, (150, 3))).astype(np.float32)
    far = np.ones((40, 4), np.float32)
    far[:, :3] = rng.uniform(-3.0, 3.0, (40, 3)).astype(np.float32)          # outside [-1.1, 1.82)
    edge = np.ones((40, 4), np.float32)
    edge[:, :3] = rng.uniform(-1.1, -1.1 + 2 * h, (40, 3)).astype(np.float32)  # cells 0..1 in x,y,z
    pos = np.concatenate([blob, far, edge])
    vel = np.zeros_like(pos)
    for ref in (False, True):
        o, s = make_pair(p, pos, vel, ref=ref)
        o.step(1, stop=STOP_FORCES); s.step_partial(capi.STAGE_FORCES)
        np.testing.assert_array_equal(s.get("hash"), o.get("hash"))
        np.testing.assert_array_equal(s.get("index"), o.get("index"))
        check_cell_tables(s.get("cellStart"), s.get("cellEnd"), o.get("cellStart"), o.get("cellEnd"))
        assert rel_err(s.get("dens"), o.get("dens")) <= TOL_STAGE
        assert rel_err(s.get("forces"), o.get("forces")) <= 10 * TOL_STAGE


def test_compact_scan_guards(hip_lib):
    """The quantised neighbour scan (nrs_math.h quantize_pos, Sweep::scan_compact) and its ways out: owners further from the
    grid origin than the quanta's error budget covers (2^20 quanta = 4096 cells: reference-order walk for those lanes), NaN
    positions, a cell size the modulo-4-cells trick cannot serve (h > 1.99 cellSize: the context builds no lists at all), and a
    grid narrower than four cells in x (tags by position would alias)."""
    base = Oracle.default_params(SESPH)
    h = float(base["interactionRadius"][0])
    rng = np.random.default_rng(23)

    def cloud(centre, n, spread):
        a = np.ones((n, 4), np.float32)
        a[:, :3] = (np.asarray(centre) + rng.uniform(-spread * h, spread * h, (n, 3))).astype(np.float32)
        return a

    near = cloud([0.2, 0.1, -0.3], 400, 2.5)
    far_out = cloud([-1.1 + 5000.3 * h, 0.3, 0.2], 60, 1.2)     # ~5000 cells from the grid origin in x: beyond QP_FAR
    edge_far = cloud([-1.1 + 4095.6 * h, -0.2, 0.1], 80, 1.0)   # straddles the 4096-cell limit
    cases = []
    pos = np.concatenate([near, far_out, edge_far])
    cases.append(("far owners", base, pos))
    nanpos = np.concatenate([near, cloud([0.5, 0.5, 0.5], 30, 1.0)])
    nanpos[5, 0] = np.nan
    nanpos[410, 2] = np.nan   # (not inf: float -> int conversion of inf saturates differently on the host and the device)
    cases.append(("nan positions", base, nanpos))
    small = base.copy()
    small["cellSize"][0] = np.float32(0.45 * h)                  # h / cellSize = 2.2: pairs inside h may be > 2 cells apart
    cases.append(("cell size 0.45 h", small, near))
    narrow = base.copy()
    narrow["gridSize"][0] = (2, 64, 64)
    narrow["numCells"][0] = 2 * 64 * 64
    cases.append(("grid two cells wide", narrow, near))
    # with boundary particles the far owners are handed to the wall workgroups (exact-position scan) by the reorder kernel
    wall = np.ones((300, 4), np.float32)
    wall[:, :3] = (np.array([0.2, 0.1, -0.3]) + rng.uniform(-2.5 * h, 2.5 * h, (300, 3))).astype(np.float32)
    wall[:, 2] = np.float32(-0.3 - 2.6 * h)
    cases.append(("far owners, with walls", base, pos, wall, np.full(300, 2e-5, np.float32)))
    for case in cases:
        name, p, pos = case[:3]
        bi, vbi = (case[3], case[4]) if len(case) > 3 else (None, None)
        vel = np.zeros_like(pos)
        o, s = make_pair(p, pos, vel, bi, vbi)
        o.step(1, stop=STOP_FORCES); s.step_partial(capi.STAGE_FORCES)
        np.testing.assert_array_equal(s.get("hash"), o.get("hash"), err_msg=name)
        np.testing.assert_array_equal(s.get("index"), o.get("index"), err_msg=name)
        gd, od = s.get("dens"), o.get("dens")
        fin = np.isfinite(od)
        np.testing.assert_array_equal(np.isfinite(gd), fin, err_msg=name)
        assert rel_err(gd[fin], od[fin]) <= TOL_STAGE, name
        gf, of = s.get("forces")[:, :3], o.get("forces")[:, :3]
        finf = np.isfinite(of).all(1)
        np.testing.assert_array_equal(np.isfinite(gf).all(1), finf, err_msg=name)
        assert rel_err(gf[finf], of[finf]) <= 10 * TOL_STAGE, name
        s.close()


def test_in_range_division_guards(hip_lib):
    """The force walk and the density walk run most divisions and the square root of a hit as the bare arithmetic steps of the compiler's own
    expansions (nrs_math.h "operands in range") behind guards; this drives every guard into its way out and asks for the reference-order
    kernels' bits: owners with a coordinate of exactly +0 / -0 (the wave's vote fails: compiler divisions for the whole wave), owners and
    neighbours with coordinates of 1e-30 and in the denormal range (differences below 2^-100), pairs closer than h / 256 and exactly
    coincident pairs (the lane repeats its walk; rij / |rij| is then NaN as in the reference), and a cloud within four cells of the grid
    origin in x (the density walk's cell-tag division)."""
    base = Oracle.default_params(SESPH)
    h = float(base["interactionRadius"][0])
    rng = np.random.default_rng(77)

    def cloud(centre, n, spread):
        a = np.ones((n, 4), np.float32)
        a[:, :3] = (np.asarray(centre) + rng.uniform(-spread * h, spread * h, (n, 3))).astype(np.float32)
        return a

    cases = []
    z = cloud([0.0, 0.0, 0.0], 3000, 4.0)
    z[0:40, 0] = 0.0; z[40:80, 1] = -0.0; z[80:120, 2] = 0.0; z[120:140, :3] = 0.0   # (the last 20 are coincident at the origin)
    cases.append(("zero coordinates", z))
    t = cloud([0.0, 0.0, 0.0], 3000, 4.0)
    t[0:30, 0] = 1e-30; t[30:60, 0] = 2e-30; t[60:90, 0] = -1e-38; t[90:120, 0] = 1e-42; t[120:150, 1] = 3e-33
    t[150:180, 0] = 1.4e-20; t[180:210, 0] = 2.0e-20; t[210:240, 0] = 1.4e-20 * (1 + 2.0 ** -22)   # (just above the owner guard's 2^-66)
    t[0:240, 2] = np.repeat(rng.uniform(-h, h, 24).astype(np.float32), 10)                 # (so that they are neighbours of each other)
    t[0:240, 1] = np.where((np.arange(240) < 120) | (np.arange(240) >= 150), np.tile(rng.uniform(-0.3 * h, 0.3 * h, 10).astype(np.float32), 24), t[0:240, 1])
    cases.append(("tiny coordinates", t))
    c = cloud([0.31, -0.22, 0.4], 3000, 4.0)
    c[1000:1100, :3] = c[0:100, :3] + np.float32(1e-4 * h)
    c[1100:1200, :3] = c[100:200, :3] * np.float32(1 + 2e-7)
    c[1200:1300, :3] = c[200:300, :3]
    cases.append(("close and coincident pairs", c))
    cases.append(("next to the grid origin", cloud([-1.1 + 3.0 * h, 0.1, -0.2], 3000, 2.9)))
    zero_origin = base.copy()
    zero_origin["worldOrigin"][0] = (0.0, -1.1, -1.1)      # (then x - origin can be tiny: the density walk decides by a vote of the owners)
    cases.append(("grid origin at zero", cloud([3.0 * h, 0.1, -0.2], 3000, 2.9), zero_origin))
    tiny_origin = cloud([3.0 * h, 0.1, -0.2], 3000, 2.9)
    tiny_origin[0:60, 0] = np.repeat(np.float32([1e-30, 3e-29, -2e-31, 1e-40, 5e-28, 0.0]), 10)
    cases.append(("grid origin at zero, tiny x", tiny_origin, zero_origin))
    for case in cases:
        name, pos = case[:2]
        base_case = case[2] if len(case) > 2 else base
        vel = np.zeros_like(pos)
        vel[:, :3] = rng.uniform(-0.5, 0.5, (len(pos), 3)).astype(np.float32)
        outs = []
        for ref in (False, True):
            s = capi.Solver(base_case, len(pos), reference_order=ref)
            s.set_particles(pos, vel)
            s.set_boundaries(None, None, update_grid=True)
            s.step_partial(capi.STAGE_FORCES)
            outs.append([s.get("hash"), s.get("index"), s.get("dens"), s.get("pres"), s.get("forces")])
            s.set_particles(pos, vel)
            s.step(3)
            outs[-1] += list(s.download())
            s.close()
        for a, b in zip(*outs):
            np.testing.assert_array_equal(a, b, err_msg=name)    # (NaN == NaN here)
        o = Oracle(base_case, False, 1, SESPH)
        o.set_particles(pos, vel)
        o.set_boundaries(None, None, update_grid=True)
        o.step(1, stop=STOP_FORCES)
        np.testing.assert_array_equal(outs[0][0], o.get("hash"), err_msg=name)
        np.testing.assert_array_equal(outs[0][1], o.get("index"), err_msg=name)
        od, of = o.get("dens"), o.get("forces")[:, :3]
        np.testing.assert_array_equal(outs[0][2], od, err_msg=name)
        fin = np.isfinite(of).all(1)
        np.testing.assert_array_equal(np.isfinite(outs[0][4][:, :3]).all(1), fin, err_msg=name)
        assert rel_err(outs[0][4][:, :3][fin], of[fin]) <= 10 * TOL_STAGE, name


def test_iisph_stages_and_steps(hip_lib):
    p, pos, vel = compressed_block()
    o, s = make_pair(p, pos, vel, solver=IISPH)
    stages = [(STOP_I_DISPLACEMENT, capi.STAGE_I_DISPLACEMENT, ["dens", "velAdv", "forcesAdv", "diiFluid", "diiBoundary"]),
              (STOP_I_ADVECTION, capi.STAGE_I_ADVECTION, ["densAdv", "aii", "P_l"]),
              (STOP_I_SOLVE, capi.STAGE_I_SOLVE, ["sumDij", "densCorr", "P_l", "pres"]),
              (STOP_I_PFORCE, capi.STAGE_I_PFORCE, ["forcesP"])]
    for ostop, gstop, names in stages:
        o.set_particles(pos, vel); s.set_particles(pos, vel)
        o.step(1, stop=ostop); s.step_partial(gstop)
        for nm in names:
            assert rel_err(s.get(nm), o.get(nm)) <= 5 * TOL_STAGE, nm
    assert s.last_iterations == o.last_iters
    g = np.load(os.path.join(GOLD, "iisph_compressed.npz"))
    assert rel_err(s.get("forcesP"), g["forcesP"]) <= 5 * TOL_STAGE
    o.set_particles(pos, vel); s.set_particles(pos, vel)
    o.step(5); s.step(5)
    gp, gv, gpr = s.download(pressure=True)
    assert s.last_iterations == o.last_iters == int(g["iters5"][0])
    assert rel_err(gp[:, :3], o.get("pos")[:, :3]) <= TOL_STEPS
    assert rel_err(gv[:, :3], o.get("vel")[:, :3]) <= TOL_STEPS
    assert rel_err(gpr, o.get("pressure")) <= 10 * TOL_STEPS
    assert rel_err(gp[:, :3], g["pos5"][:, :3]) <= TOL_STEPS


def test_iisph_list_kernels_equal_reference_order_bitwise(hip_lib):
    """The list-driven IISPH chain (one scan per step) against the plain reference-order kernels, bit for bit:
    no boundaries, with boundaries, and a crowded blob whose hit lists overflow (per-particle fallback)."""
    scenes = []
    p, pos, vel = compressed_block()
    scenes.append((p, pos, vel, None, None))
    p2, sc = small_dam_break(solver=IISPH)
    scenes.append((p2, sc["pos"], sc["vel"], sc["bi"], sc["vbi"]))
    rng = np.random.default_rng(5)
    h = float(p["interactionRadius"][0])
    blob = np.ones((120, 4), np.float32)
    blob[:, :3] = (np.array([0.2, 0.1, -0.3]) + rng.uniform(-0.45 * h, 0.45 * h, (120, 3))).astype(np.float32)
    loose = np.ones((200, 4), np.float32)
    loose[:, :3] = (np.array([0.2, 0.1, -0.3]) + rng.uniform(-3 * h, 3 * h, (200, 3))).astype(np.float32)
    crowd = np.concatenate([blob, loose])
    scenes.append((p, crowd, np.zeros_like(crowd), None, None))
    names = ["dens", "velAdv", "forcesAdv", "diiFluid", "diiBoundary", "densAdv", "aii", "sumDij", "densCorr", "P_l", "pres",
             "forcesP"]
    for (pp, pos, vel, bi, vbi) in scenes:
        outs = []
        for ref in (False, True):
            s = capi.Solver(pp, len(pos), solver=capi.IISPH, reference_order=ref)
            s.set_particles(pos, vel)
            s.set_boundaries(bi, vbi, update_grid=True)
            s.step_partial(capi.STAGE_I_PFORCE)
            got = [s.get(nm) for nm in names] + [np.array([s.last_iterations])]
            s.set_particles(pos, vel)
            s.step(3)
            got += list(s.download(pressure=True))
            outs.append(got)
            s.close()
        for nm, a, b in zip(names + ["iters", "pos", "vel", "pressure"], *outs):
            np.testing.assert_array_equal(a, b, err_msg=nm)


def test_iisph_self_by_slot_flag(hip_lib):
    """NRS_FLAG_IISPH_SELF_BY_SLOT (SURVEY Q5 off): against the oracle in the same mode, both kernel paths; and the point of the
    flag — the result no longer depends on the order of the input arrays (with the default flags it does, by centimetres)."""
    from scipy.spatial import cKDTree

    p, pos, vel = compressed_block()
    vel = vel.copy()
    vel[:, 0] = np.where(np.arange(len(pos)) % 2 == 0, 1.0, -1.0).astype(np.float32)   # movers, so that Q5 bites
    perm = np.random.default_rng(3).permutation(len(pos))
    o = Oracle(p, solver=IISPH, self_by_slot=True)
    o.set_particles(pos, vel); o.set_boundaries(None, None)
    o.step(4)
    spread = {}
    for flags in (capi.FLAG_IISPH_SELF_BY_SLOT, 0):
        outs = []
        for ref in (False, True):
            for order in (None, perm):
                s = capi.Solver(p, len(pos), solver=capi.IISPH, reference_order=ref, flags=flags)
                s.set_particles(pos if order is None else pos[order], vel if order is None else vel[order])
                s.set_boundaries(None, None, update_grid=True)
                s.step(4)
                gp, gv = s.download()
                if flags and order is None:
                    assert s.last_iterations == o.last_iters
                    assert rel_err(gp[:, :3], o.get("pos")[:, :3]) <= TOL_STEPS
                    assert rel_err(gv[:, :3], o.get("vel")[:, :3]) <= TOL_STEPS
                outs.append(gp)
                s.close()
        np.testing.assert_array_equal(outs[0], outs[2])   # list-driven chain == reference-order kernels, bit for bit
        d, idx = cKDTree(outs[0][:, :3]).query(outs[1][:, :3])
        spread[flags] = float(d.max())
    assert spread[capi.FLAG_IISPH_SELF_BY_SLOT] < 1e-5 and spread[0] > 1e-3, spread


def test_iisph_with_boundaries(hip_lib):
    p, sc = small_dam_break(solver=IISPH)
    o, s = make_pair(p, sc["pos"], sc["vel"], sc["bi"], sc["vbi"], solver=IISPH)
    o.step(1, stop=STOP_I_PFORCE); s.step_partial(capi.STAGE_I_PFORCE)
    for nm in ("dens", "diiBoundary", "aii", "densAdv", "sumDij", "densCorr", "forcesP"):
        assert rel_err(s.get(nm), o.get(nm)) <= 5 * TOL_STAGE, nm
    o.set_particles(sc["pos"], sc["vel"]); s.set_particles(sc["pos"], sc["vel"])
    o.step(5); s.step(5)
    gp, gv = s.download()
    assert rel_err(gp[:, :3], o.get("pos")[:, :3]) <= TOL_STEPS
    assert rel_err(gv[:, :3], o.get("vel")[:, :3]) <= TOL_STEPS


@pytest.mark.parametrize("double,kset", [(False, 0), (True, 1), (True, 0)], ids=["f32-monaghan", "f64-muller", "f64-monaghan"])
def test_other_precision_and_kernel_sets(hip_lib, double, kset):
    """DOUBLE_PRECISION / KERNEL_SET variants (config 5 = fp64 + Monaghan), incl. the float-scalar helper
    semantics of SURVEY Q11."""
    p, sc = small_dam_break(double=double, kernel_set=kset)
    for ref in (False, True):
        o, s = make_pair(p, sc["pos"], sc["vel"], sc["bi"], sc["vbi"], double=double, kset=kset, ref=ref)
        o.step(1, stop=STOP_FORCES); s.step_partial(capi.STAGE_FORCES)
        np.testing.assert_array_equal(s.get("hash"), o.get("hash"))
        np.testing.assert_array_equal(s.get("index"), o.get("index"))
        assert rel_err(s.get("dens"), o.get("dens")) <= TOL_STAGE
        assert rel_err(s.get("forces"), o.get("forces")) <= TOL_STAGE
        o.set_particles(sc["pos"], sc["vel"]); s.set_particles(sc["pos"], sc["vel"])
        o.step(10); s.step(10)
        gp, gv = s.download()
        assert rel_err(gp[:, :3], o.get("pos")[:, :3]) <= TOL_STEPS
        assert rel_err(gv[:, :3], o.get("vel")[:, :3]) <= TOL_STEPS


def test_full_size_c2_properties(hip_lib):
    """BASELINE config C2 (1,000,000 particles + tank): size-independent properties, tiled == reference-order
    bit for bit, and the oracle on the same inputs for one full force evaluation."""
    p = Oracle.default_params(SESPH)
    sc = scene.dam_break("C2", h=float(p["interactionRadius"][0]), kpoly=float(p["kpoly"][0]))
    n = len(sc["pos"])
    assert n == 1_000_000
    res = []
    for ref in (False, True):
        s = capi.Solver(p, n, reference_order=ref)
        s.set_particles(sc["pos"], sc["vel"])
        s.set_boundaries(sc["bi"], sc["vbi"], update_grid=True)
        s.step_partial(capi.STAGE_FORCES)
        res.append(dict(hash=s.get("hash"), index=s.get("index"), cs=s.get("cellStart"), ce=s.get("cellEnd"),
                        dens=s.get("dens"), forces=s.get("forces"), params=s.params))
        s.close()
    a, b = res
    assert np.all(np.diff(a["hash"].astype(np.int64)) >= 0)                 # sortedness
    assert np.array_equal(np.sort(a["index"]), np.arange(n, dtype=np.uint32))  # a permutation
    m = a["cs"] != 0xFFFFFFFF
    assert int((a["ce"][m].astype(np.int64) - a["cs"][m]).sum()) == n        # cells tile the sorted array
    assert np.array_equal(np.unique(a["hash"]), np.nonzero(m)[0])
    for k in ("hash", "index", "dens", "forces"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    o = Oracle(p, solver=SESPH, threads=min(16, os.cpu_count() or 1))
    o.set_particles(sc["pos"], sc["vel"])
    o.set_boundaries(sc["bi"], sc["vbi"], update_grid=True)
    np.testing.assert_array_equal(a["params"].view(np.uint8), o.params.view(np.uint8))
    o.step(1, stop=STOP_FORCES)
    np.testing.assert_array_equal(a["hash"], o.get("hash"))
    np.testing.assert_array_equal(a["index"], o.get("index"))
    assert rel_err(a["dens"], o.get("dens")) <= TOL_STAGE
    assert rel_err(a["forces"], o.get("forces")) <= TOL_STAGE


def test_full_size_c2_staged_scan_vs_oracle(hip_lib):
    """BASELINE config C2 with NRS_FLAG_STAGED_SCAN: one full force evaluation of the LDS-staged density launch (+ the list-driven
    force kernel fed by its lists) against the oracle on the same inputs, and against the default launch bit for bit."""
    p = Oracle.default_params(SESPH)
    sc = scene.dam_break("C2", h=float(p["interactionRadius"][0]), kpoly=float(p["kpoly"][0]))
    n = len(sc["pos"])
    res = []
    for fl in (capi.FLAG_STAGED_SCAN, 0):
        s = capi.Solver(p, n, flags=fl)
        s.set_particles(sc["pos"], sc["vel"])
        s.set_boundaries(sc["bi"], sc["vbi"], update_grid=True)
        s.step_partial(capi.STAGE_FORCES)
        res.append({k: s.get(k) for k in ("hash", "index", "dens", "pres", "forces")})
        s.close()
    for k in res[0]:
        np.testing.assert_array_equal(res[0][k], res[1][k], err_msg=k)
    o = Oracle(p, solver=SESPH, threads=min(16, os.cpu_count() or 1))
    o.set_particles(sc["pos"], sc["vel"])
    o.set_boundaries(sc["bi"], sc["vbi"], update_grid=True)
    o.step(1, stop=STOP_FORCES)
    np.testing.assert_array_equal(res[0]["hash"], o.get("hash"))
    np.testing.assert_array_equal(res[0]["index"], o.get("index"))
    np.testing.assert_array_equal(res[0]["dens"], o.get("dens"))
    assert rel_err(res[0]["forces"], o.get("forces")) <= TOL_STAGE


def test_full_size_c2_velocity_bar_tait_double7(hip_lib):
    """The north_star float bar — positions AND velocities within 1e-5 after N steps — at BASELINE config C2 (1 M particles)
    after 50 and 150 steps, particle by particle, asserted in the oracle's `tait = double7` mode: x^7 of the Tait equation formed
    in double and rounded once, which is what the device evaluates (nrs_math.h pow7f).  In the oracle's default mode (glibc powf,
    1 ulp away in ~0.1 % of the pressures) the weakly compressible system amplifies that last bit to 2.6e-5 / 1.3e-4 on the
    velocities (tests/test_fast_arith_gpu.py records it): there the bar measures two pow conventions, here it
    measures the kernels.  Every other operation of the step is the same IEEE sequence on both sides, so the states are expected
    to agree to the last bit; the assertion is the bar."""
    p = Oracle.default_params(SESPH)
    sc = scene.dam_break("C2", h=float(p["interactionRadius"][0]), kpoly=float(p["kpoly"][0]))
    n = len(sc["pos"])
    vel = sc["vel"].copy()
    vel[:, 3] = np.arange(n, dtype=np.float32)   # particle id rides in vel.w (the SESPH step preserves w)
    o, s = make_pair(p, sc["pos"], vel, sc["bi"], sc["vbi"], tait="double7", threads=min(16, os.cpu_count() or 1))
    done = 0
    for k in (50, 150):
        o.step(k - done); s.step(k - done)
        done = k
        gp, gv = s.download()
        op, ov = o.get("pos"), o.get("vel")
        ig, io = np.argsort(gv[:, 3], kind="stable"), np.argsort(ov[:, 3], kind="stable")
        ep, ev = rel_err(gp[ig, :3], op[io, :3]), rel_err(gv[ig, :3], ov[io, :3])
        print("C2 exact vs oracle(tait=double7) after %d steps: pos %.2e vel %.2e, bitwise %s" % (
            k, ep, ev, np.array_equal(gp[ig], op[io]) and np.array_equal(gv[ig], ov[io])))
        assert ep <= TOL_STEPS and ev <= TOL_STEPS, (k, ep, ev)
        np.testing.assert_array_equal(s.get("hash"), o.get("hash"))
        np.testing.assert_array_equal(s.get("index"), o.get("index"))
    assert s.resort_stats() == (149, 0)


def test_full_size_north_star_production_equals_reference_order(hip_lib):
    """The bench workload itself (216^3 = 10,077,696 particles + tank, the north-star size): one force evaluation and three full
    steps on the production path (quantised scan, wall workgroups, pair gathers, fused force launch, coherent re-sort) against the
    reference-order kernels with a full sort — every array bit for bit; plus the size-independent properties; plus the CPU oracle on
    the same 10 M particles (VERDICT r2 missing 5)."""
    p = Oracle.default_params(SESPH)
    sc = scene.dam_break("NS", h=float(p["interactionRadius"][0]), kpoly=float(p["kpoly"][0]))
    n = len(sc["pos"])
    assert n == 216 ** 3
    res = []
    for ref in (False, True):
        s = capi.Solver(p, n, reference_order=ref, flags=capi.FLAG_FULL_SORT if ref else 0)
        s.set_particles(sc["pos"], sc["vel"])
        s.set_boundaries(sc["bi"], sc["vbi"], update_grid=True)
        s.step_partial(capi.STAGE_FORCES)
        r = dict(hash=s.get("hash"), index=s.get("index"), dens=s.get("dens"), pres=s.get("pres"), forces=s.get("forces"))
        s.set_particles(sc["pos"], sc["vel"])
        s.step(3)
        r["pos"], r["vel"] = s.download()
        if not ref:
            assert s.resort_stats()[0] >= 2     # the merge path ran
        res.append(r)
        s.close()
    a, b = res
    assert np.all(np.diff(a["hash"].astype(np.int64)) >= 0)
    assert np.array_equal(np.sort(a["index"]), np.arange(n, dtype=np.uint32))
    assert np.isfinite(a["pos"]).all() and np.isfinite(a["vel"]).all() and float(a["dens"].min()) > 0
    for k in ("hash", "index", "dens", "pres", "forces", "pos", "vel"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    # ... and the ORACLE at the size the bench quotes: one force evaluation (hash / index bit-exact, density / forces within the
    # stage tolerance), then the same three steps (positions / velocities within the north_star bar)
    del b
    o = Oracle(p, solver=SESPH, threads=min(16, os.cpu_count() or 1))
    o.set_particles(sc["pos"], sc["vel"])
    o.set_boundaries(sc["bi"], sc["vbi"], update_grid=True)
    o.step(1, stop=STOP_FORCES)
    np.testing.assert_array_equal(a["hash"], o.get("hash"))
    np.testing.assert_array_equal(a["index"], o.get("index"))
    np.testing.assert_array_equal(a["dens"], o.get("dens"))
    assert max_ulp(a["pres"], o.get("pres")) <= 1
    assert rel_err(a["forces"], o.get("forces")) <= TOL_STAGE
    o.set_particles(sc["pos"], sc["vel"])
    o.step(3)
    assert rel_err(a["pos"][:, :3], o.get("pos")[:, :3]) <= TOL_STEPS
    assert rel_err(a["vel"][:, :3], o.get("vel")[:, :3]) <= TOL_STEPS
    np.testing.assert_array_equal(a["pos"][:, 3], o.get("pos")[:, 3])


def test_full_size_north_star_developed_state_bitwise(hip_lib):
    """The same scene once the dam has broken (state after 760 steps: ~11 neighbours within h, 11 % movers per step, lists close to
    their capacity): density, pressure, forces and one full step of the production path against the reference-order kernels, bit
    for bit, from that state."""
    p = Oracle.default_params(SESPH)
    sc = scene.dam_break("NS", h=float(p["interactionRadius"][0]), kpoly=float(p["kpoly"][0]))
    n = len(sc["pos"])
    s = capi.Solver(p, n)
    s.set_particles(sc["pos"], sc["vel"])
    s.set_boundaries(sc["bi"], sc["vbi"], update_grid=True)
    s.step(760)
    pos, vel = s.download()
    assert np.isfinite(pos).all() and s.get_stat(capi.STAT_HIT_MEAN) > 8.0
    s.close()
    res = []
    for ref in (False, True):
        s = capi.Solver(p, n, reference_order=ref)
        s.set_particles(pos, vel)
        s.set_boundaries(sc["bi"], sc["vbi"], update_grid=True)
        s.step_partial(capi.STAGE_FORCES)
        r = dict(index=s.get("index"), dens=s.get("dens"), pres=s.get("pres"), forces=s.get("forces"))
        s.set_particles(pos, vel)
        s.step(1)
        r["pos"], r["vel"] = s.download()
        res.append(r)
        s.close()
    for k in res[0]:
        np.testing.assert_array_equal(res[0][k], res[1][k], err_msg=k)


def test_full_size_c2_resort_long_run_bitwise(hip_lib):
    """BASELINE config C2 for 150 steps: the default path (coherent re-sort, fused launches, shared hit lists) and the
    reference-shaped path (full sort every step, separate launches) end in the same bits; the mover fraction grows
    along the run, every step after the first took the merge path."""
    p = Oracle.default_params(SESPH)
    sc = scene.dam_break("C2", h=float(p["interactionRadius"][0]), kpoly=float(p["kpoly"][0]))
    n = len(sc["pos"])
    outs = []
    for flags in (0, capi.FLAG_FULL_SORT | capi.FLAG_NO_FUSION):
        s = capi.Solver(p, n, flags=flags)
        s.set_particles(sc["pos"], sc["vel"])
        s.set_boundaries(sc["bi"], sc["vbi"], update_grid=True)
        s.step(150)
        outs.append(s.download() + (s.get("hash"), s.get("index"), s.get("dens")))
        if flags == 0:
            assert s.resort_stats() == (149, 0)
        s.close()
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)
    h = outs[0][2].astype(np.int64)
    assert np.all(np.diff(h) >= 0) and np.array_equal(np.sort(outs[0][3]), np.arange(n, dtype=np.uint32))


def test_full_size_c3_iisph_properties(hip_lib):
    """BASELINE config C3 (160^3 = 4,096,000 particles, IISPH, fp32): one full step against the oracle on the same
    inputs (threads on the host cores), plus solver-loop properties."""
    p = Oracle.default_params(IISPH)
    sc = scene.dam_break("C3", h=float(p["interactionRadius"][0]), kpoly=float(p["kpoly"][0]))
    n = len(sc["pos"])
    assert n == 4_096_000
    s = capi.Solver(p, n, solver=capi.IISPH)
    s.set_particles(sc["pos"], sc["vel"])
    s.set_boundaries(sc["bi"], sc["vbi"], update_grid=True)
    s.step(1)
    gp, gv, gpr = s.download(pressure=True)
    assert s.last_iterations >= 2                      # the reference's loop runs at least twice (sph_cuda.cu:741)
    assert np.isfinite(gp).all() and np.isfinite(gv).all() and (gpr >= 0).all()   # pressures are clamped at 0
    assert np.all(gp[:, 3] == 1.0) and np.all(gv[:, 3] == 0.0)   # iisph_integrate sets w (sph_kernel_impl.cuh:1653)
    o = Oracle(p, solver=IISPH, threads=min(16, os.cpu_count() or 1))
    o.set_particles(sc["pos"], sc["vel"])
    o.set_boundaries(sc["bi"], sc["vbi"], update_grid=True)
    o.step(1)
    assert s.last_iterations == o.last_iters
    np.testing.assert_array_equal(s.get("hash"), o.get("hash"))
    np.testing.assert_array_equal(s.get("index"), o.get("index"))
    assert rel_err(gp[:, :3], o.get("pos")[:, :3]) <= TOL_STEPS
    assert rel_err(gv[:, :3], o.get("vel")[:, :3]) <= TOL_STEPS
    assert rel_err(gpr, o.get("pressure")) <= 10 * TOL_STEPS


def test_full_size_c5_fp64_monaghan(hip_lib):
    """BASELINE config C5 (1,000,000 particles, DOUBLE_PRECISION=1, KERNEL_SET=0): tiled == reference-order bit
    for bit, and the oracle's density/forces on the same inputs."""
    p = Oracle.default_params(SESPH, double=True, kernel_set=0)
    sc = scene.dam_break("C5", h=float(p["interactionRadius"][0]), kpoly=float(p["kpoly"][0]), real=np.float64)
    n = len(sc["pos"])
    res = []
    for ref in (False, True):
        s = capi.Solver(p, n, double=True, kernel_set=0, reference_order=ref)
        s.set_particles(sc["pos"], sc["vel"])
        s.set_boundaries(sc["bi"], sc["vbi"], update_grid=True)
        s.step_partial(capi.STAGE_FORCES)
        res.append((s.get("hash"), s.get("index"), s.get("dens"), s.get("pres"), s.get("forces")))
        s.close()
    for a, b in zip(*res):
        np.testing.assert_array_equal(a, b)
    o = Oracle(p, True, 0, SESPH, threads=min(16, os.cpu_count() or 1))
    o.set_particles(sc["pos"], sc["vel"])
    o.set_boundaries(sc["bi"], sc["vbi"], update_grid=True)
    o.step(1, stop=STOP_FORCES)
    np.testing.assert_array_equal(res[0][0], o.get("hash"))
    np.testing.assert_array_equal(res[0][1], o.get("index"))
    assert rel_err(res[0][2], o.get("dens")) <= TOL_STAGE
    assert rel_err(res[0][4], o.get("forces")) <= TOL_STAGE


def test_params_change_and_diagnostics(hip_lib):
    p, pos, vel = default_scene(SESPH)
    o, s = make_pair(p, pos, vel)
    q = p.copy()
    q["gravity"][0][1] = 0.0        # setGravity(0.0), main.cpp:538
    q["gridSize"][0] = (128, 64, 32)  # a re-grid: cell tables are re-allocated
    q["numCells"][0] = 128 * 64 * 32
    o.set_params(q); s.set_params(q)
    o.step(3); s.step(3)
    gp, gv = s.download()
    assert rel_err(gp[:, :3], o.get("pos")[:, :3]) <= TOL_STEPS
    np.testing.assert_array_equal(s.get("hash"), o.get("hash"))
    assert abs(s.max_density() - float(o.get("dens").max())) <= 1e-3
    assert abs(s.max_velocity() - float(np.linalg.norm(o.get("vel")[:, :3].astype(np.float64), axis=1).max())) <= 1e-5


def test_regrid_between_steps_invalidates_prepared_keys(hip_lib):
    """SPH::updateGpuBoundaries / nrs_set_boundaries(update_grid=1) after fused steps: the keys the force kernel prepared
    for the next step belong to the OLD grid and must not be used (stale hashNext -> wrong cells, or a cell
    index beyond a smaller table).  Shifted tank (new origin, same cell count) and a smaller tank (fewer cells)."""
    p, sc = small_dam_break((14, 12, 10))
    o, s = make_pair(p, sc["pos"], sc["vel"], sc["bi"], sc["vbi"])
    o.step(3); s.step(3)
    np.testing.assert_array_equal(s.get("hash"), o.get("hash"))
    for variant in ("shifted", "smaller"):
        bi = sc["bi"].copy()
        if variant == "shifted":
            bi[:, :3] += np.array([0.013, 0.0, 0.071], dtype=bi.dtype)
            vbi = sc["vbi"]
        else:  # drop the far third of the tank in x: the AABB (and the pow2 grid) shrinks
            keep = bi[:, 0] <= 0.62 * bi[:, 0].max()
            bi, vbi = bi[keep], sc["vbi"][keep]
        o.set_boundaries(bi, vbi, update_grid=True)
        s.set_boundaries(bi, vbi, update_grid=True)
        np.testing.assert_array_equal(s.params.view(np.uint8), o.params.view(np.uint8))
        o.step(3); s.step(3)
        np.testing.assert_array_equal(s.get("hash"), o.get("hash"))
        np.testing.assert_array_equal(s.get("index"), o.get("index"))
        gp, gv = s.download()
        assert rel_err(gp[:, :3], o.get("pos")[:, :3]) <= TOL_STEPS
        assert rel_err(gv[:, :3], o.get("vel")[:, :3]) <= TOL_STEPS
    # set_params with a moved origin but the same cell count must also rebuild the boundary tables
    q = s.params.copy()
    q["worldOrigin"][0][0] -= q["cellSize"][0][0] * 0.5
    o.set_params(q); s.set_params(q)
    o.set_boundaries(bi, vbi, update_grid=False)  # the oracle re-hashes its boundaries here; the library did in set_params
    o.step(2); s.step(2)
    np.testing.assert_array_equal(s.get("bhash"), o.get("bhash"))
    np.testing.assert_array_equal(s.get("hash"), o.get("hash"))
    gp, _ = s.download()
    assert rel_err(gp[:, :3], o.get("pos")[:, :3]) <= TOL_STEPS


def test_step_statistics(hip_lib):
    """nrs_get_stat: mover count of the coherent re-sort and the hit-list diagnostics bench.py's `developed` record reports"""
    p, sc = small_dam_break((40, 32, 28))
    n = len(sc["pos"])
    s = capi.Solver(p, n)
    s.set_particles(sc["pos"], sc["vel"])
    s.set_boundaries(sc["bi"], sc["vbi"], update_grid=True)
    s.step(6)
    movers = s.get_stat(capi.STAT_MOVERS)
    assert 0 <= movers <= n // 2
    # reference: hit counts recomputed on the host from the sorted positions of the last density pass are not available,
    # so check the invariants: a resting lattice at spacing h-0.005 has at most 6 fluid neighbours within h
    assert s.get_stat(capi.STAT_HIT_OVERFLOW) == 0
    mean, mx = s.get_stat(capi.STAT_HIT_MEAN), s.get_stat(capi.STAT_HIT_MAX)
    assert 3.0 < mean < 20.0 and mean <= mx <= 20
    r = capi.Solver(p, n, reference_order=True)
    r.set_particles(sc["pos"], sc["vel"])
    r.step(1)
    with pytest.raises(capi.NereusError):
        r.get_stat(capi.STAT_HIT_MEAN)


def test_host_driven_iisph_phase_guards(hip_lib):
    """nrs_iisph_predict .. nrs_iisph_finish: while the phase is open, calls that would change the grid, the cuts or the
    arrays under the predicted hit lists are refused with NRS_E_STATE; uploading particles abandons the step and the context is
    as good as new — the following run equals a fresh context bit for bit."""
    p, pos, vel = compressed_block()
    s = capi.Solver(p, 2 * len(pos), solver=capi.IISPH)
    s.set_particles(pos, vel)
    s.iisph_predict()
    s.iisph_iterate()
    for call in (lambda: s.set_params(p), lambda: s.set_boundaries(pos[:10], np.full(10, 1e-5, np.float32)), lambda: s.step(1),
                 lambda: s.iisph_predict(), lambda: s.slab_configure(-1000, 1000, 8)):
        with pytest.raises(capi.NereusError):
            call()
    s.set_particles(pos, vel)            # abandons the open step
    with pytest.raises(capi.NereusError):
        s.iisph_iterate()                # (nrs_iisph_predict first)
    s.step(3)
    f = capi.Solver(p, 2 * len(pos), solver=capi.IISPH)
    f.set_particles(pos, vel)
    f.step(3)
    for a, b in zip(s.download(pressure=True), f.download(pressure=True)):
        np.testing.assert_array_equal(a, b)
    # a shrinking particle count mid-phase abandons it too
    s.iisph_predict()
    s.lib.nrs_set_num_particles(s.h, len(pos) - 7)
    s.step(1)
    assert np.isfinite(s.download()[0]).all()


def test_nrs_step_returns_before_the_device_finishes(hip_lib):
    """include/nereus_hip.h: nrs_step(ctx, k >= 2) hands the steps to the context's own thread and returns at once (VERDICT r2 weak 11:
    the mover-count read-back used to keep the CALLER inside nrs_step for the whole run).  The call returns long before the work is
    done, every later call sees the finished state, and the result is bit for bit what one synchronous step at a time gives."""
    import time

    p, sc = small_dam_break((40, 36, 32))
    n = len(sc["pos"])
    steps = 600

    def make():
        s = capi.Solver(p, n)
        s.set_particles(sc["pos"], sc["vel"])
        s.set_boundaries(sc["bi"], sc["vbi"], update_grid=True)
        s.step(2)
        s.synchronize()
        return s

    a = make()
    t0 = time.perf_counter()
    a.step(steps)
    t_call = time.perf_counter() - t0
    a.synchronize()
    t_all = time.perf_counter() - t0
    print("nrs_step(%d) returned after %.2f ms; the steps took %.1f ms" % (steps, 1e3 * t_call, 1e3 * t_all))
    assert t_call < 0.1 * t_all and a.resort_stats()[0] >= steps
    b = make()
    for _ in range(steps):
        b.step(1)           # (single steps run on the calling thread)
    for x, y in zip(a.download() + (a.get("hash"), a.get("index")), b.download() + (b.get("hash"), b.get("index"))):
        np.testing.assert_array_equal(x, y)
    # a call the state does not allow is still refused at once, also with steps in flight behind it
    a.step(50)
    a.step_partial(capi.STAGE_DENSITY)      # (waits for the 50 queued steps, then leaves the state mid-update)
    with pytest.raises(capi.NereusError):
        a.step(5)
    a.set_particles(sc["pos"], sc["vel"])
    a.step(3)
    a.step(4)                                # queued behind the three
    c = make()
    c.set_particles(sc["pos"], sc["vel"])
    c.step(7)
    for x, y in zip(a.download(), c.download()):
        np.testing.assert_array_equal(x, y)
