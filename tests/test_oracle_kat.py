"""Oracle vs the only reference-derived numbers that exist, and vs the committed fixtures.

The reference ships no tests/golden vectors; SURVEY §8c records what its kernels produced for the shipped
scenes: SESPH N=2197 with mean density 821.7374 (Muller fp32) / 458.9021 (Monaghan fp32) / 821.7375 and
458.9021 (fp64); IISPH N=1331, 2 solver iterations per step, mean density ~807.26.
"""
import os

import numpy as np
import pytest

from tests.common import compressed_block, default_scene, small_dam_break
from tests.oracle_lib import IISPH, SESPH, STOP_DENSITY, STOP_FORCES, STOP_I_PFORCE, Oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("double,kset,expect", [(0, 1, 821.7374), (0, 0, 458.9021), (1, 1, 821.7375), (1, 0, 458.9021)])
def test_sesph_default_scene_known_answers(double, kset, expect):
    p, pos, vel = default_scene(SESPH, double, kset)
    assert len(pos) == 2197
    o = Oracle(p, double, kset, SESPH)
    o.set_particles(pos, vel)
    o.set_boundaries(None, None)
    o.step(1, stop=STOP_DENSITY)
    mean = float(o.get("dens").astype(np.float64).mean())
    assert abs(mean - expect) < 6e-5, mean  # the survey printed 4 decimals


def test_iisph_default_scene_known_answers():
    p, pos, vel = default_scene(IISPH)
    assert len(pos) == 1331
    p["gravity"][0][1] = 0.0  # main.cpp:538 setGravity(0.0)
    o = Oracle(p, solver=IISPH)
    o.set_particles(pos, vel)
    o.set_boundaries(None, None)
    for _ in range(3):
        o.step(1)
        assert o.last_iters == 2
        assert abs(float(o.get("dens").astype(np.float64).mean()) - 807.26) < 0.01


def test_params_layout_and_defaults():
    p = Oracle.default_params(SESPH)
    assert p.dtype.itemsize == 132
    assert tuple(p["gridSize"][0]) == (64, 64, 64) and p["numCells"][0] == 64 ** 3
    assert p["gasStiffness"][0] == 800 and p["interactionRadius"][0] == np.float32(0.0457)
    assert np.isclose(p["particleMass"][0], 0.5 * 0.0457 ** 3 * 1000, rtol=1e-6)
    q = Oracle.default_params(IISPH)
    assert tuple(q["gridSize"][0]) == (128, 128, 128) and q["interactionRadius"][0] == np.float32(0.0537)
    assert q["beta"][0] == 1050 and q["viscosity"][0] == np.float32(0.01)
    assert Oracle.default_params(SESPH, double=True).dtype.itemsize == 240


def test_oracle_matches_committed_sesph_fixture():
    g = np.load(os.path.join(GOLD, "sesph_default.npz"))
    p, pos, vel = default_scene(SESPH)
    np.testing.assert_array_equal(pos, g["pos0"])
    o = Oracle(p, solver=SESPH)
    o.set_particles(pos, vel)
    o.set_boundaries(None, None)
    o.step(1, stop=STOP_FORCES)
    for k in ("hash", "index", "dens", "pres", "forces"):
        np.testing.assert_array_equal(o.get(k), g[k], err_msg=k)
    cs = o.get("cellStart")
    np.testing.assert_array_equal(np.nonzero(cs != 0xFFFFFFFF)[0], g["cell_ids"])
    np.testing.assert_array_equal(cs[g["cell_ids"]], g["cell_start"])
    np.testing.assert_array_equal(o.get("cellEnd")[g["cell_ids"]], g["cell_end"])
    o2 = Oracle(p, solver=SESPH)
    o2.set_particles(pos, vel)
    o2.set_boundaries(None, None)
    o2.step(10)
    np.testing.assert_array_equal(o2.get("pos"), g["pos10"])
    np.testing.assert_array_equal(o2.get("vel"), g["vel10"])


def test_oracle_matches_committed_dambreak_fixture():
    g = np.load(os.path.join(GOLD, "sesph_dambreak.npz"))
    p, sc = small_dam_break()
    np.testing.assert_array_equal(sc["pos"], g["pos0"])
    np.testing.assert_array_equal(sc["bi"], g["bi"])
    np.testing.assert_array_equal(sc["vbi"], g["vbi"])
    o = Oracle(p, solver=SESPH)
    o.set_particles(sc["pos"], sc["vel"])
    o.set_boundaries(sc["bi"], sc["vbi"], update_grid=True)
    np.testing.assert_array_equal(o.params.view(np.uint8), g["params"])
    o.step(10)
    np.testing.assert_array_equal(o.get("pos"), g["pos10"])
    np.testing.assert_array_equal(o.get("vel"), g["vel10"])


def test_oracle_matches_committed_iisph_fixture():
    g = np.load(os.path.join(GOLD, "iisph_compressed.npz"))
    p, pos, vel = compressed_block()
    o = Oracle(p, solver=IISPH)
    o.set_particles(pos, vel)
    o.set_boundaries(None, None)
    o.step(1, stop=STOP_I_PFORCE)
    assert o.last_iters == int(g["iters"][0])
    for k in ("dens", "aii", "densAdv", "P_l", "forcesP", "sumDij"):
        np.testing.assert_array_equal(o.get(k), g[k], err_msg=k)


def test_oracle_threads_do_not_change_results():
    """The OpenMP legs used by bench.py's cpu_baseline give the same bits as the sequential oracle."""
    p, sc = small_dam_break()
    outs = []
    for threads in (1, 4):
        o = Oracle(p, solver=SESPH, threads=threads)
        o.set_particles(sc["pos"], sc["vel"])
        o.set_boundaries(sc["bi"], sc["vbi"], update_grid=True)
        o.step(3)
        outs.append((o.get("pos"), o.get("vel")))
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])


def test_empty_and_single_particle():
    p = Oracle.default_params(SESPH)
    o = Oracle(p, solver=SESPH)
    o.set_particles(np.zeros((0, 4), np.float32))
    o.step(1)  # no-op, must not crash
    one = np.array([[0.1, 0.2, 0.3, 1.0]], np.float32)
    o.set_particles(one)
    o.set_boundaries(None, None)
    o.step(1, stop=STOP_DENSITY)
    h, kp, m = float(p["interactionRadius"][0]), float(p["kpoly"][0]), float(p["particleMass"][0])
    assert np.isclose(o.get("dens")[0], m * kp * h ** 6, rtol=1e-6)  # only the self term m*W(0)


@pytest.mark.parametrize("double", [False, True])
@pytest.mark.parametrize("solver", [SESPH, IISPH])
def test_python_default_params_match_oracle_bitwise(solver, double):
    """nereus_amd.params.default_params (product-side plumbing) restates the same constructors."""
    from nereus_amd.params import default_params, kernel_constants

    a = default_params(solver, double)
    b = Oracle.default_params(solver, double)
    for name in a.dtype.names:
        np.testing.assert_array_equal(a[name], b[name], err_msg=name)
    c = kernel_constants(a.copy(), 2)
    d = Oracle.recompute_constants(b, double)
    for name in a.dtype.names:
        np.testing.assert_array_equal(c[name], d[name], err_msg=name)
