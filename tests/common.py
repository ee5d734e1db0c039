"""Shared scene builders and comparison helpers for the parity tests."""
import numpy as np

from nereus_amd import scene
from tests.oracle_lib import IISPH, SESPH, Oracle


def default_scene(solver=SESPH, double=False, kernel_set=1):
    """The reference's shipped scene: constructor defaults + generateParticleCube(center(-0.4,0.04,0.5), 0.5^3)
    (main.cpp:533-538).  SESPH: N=2197, IISPH: N=1331.  No boundaries."""
    p = Oracle.default_params(solver, double, kernel_set)
    pos = Oracle.generate_cube(p, [-0.4, 0.04, 0.5, 1.0], [0.5, 0.5, 0.5, 1.0], double, kernel_set)
    vel = np.zeros_like(pos)
    return p, pos, vel


def small_dam_break(lattice=(12, 10, 9), solver=SESPH, double=False, kernel_set=1, jitter=0.01):
    """A small dam-break with the 5-face boundary box; the grid comes from the updateGrid rule."""
    p = Oracle.default_params(solver, double, kernel_set)
    real = np.float64 if double else np.float32
    sc = scene.dam_break(lattice, h=float(p["interactionRadius"][0]), kpoly=float(p["kpoly"][0]), real=real,
                         jitter=jitter)
    return p, sc


def compressed_block(lattice=(10, 10, 10), solver=IISPH, double=False, kernel_set=1, ratio=0.72):
    """A jittered lattice at spacing ratio*h (< 0.794 h, i.e. denser than rest density) so the IISPH pressure
    solve has positive pressures to work on (the shipped scene is under-dense: all pressures clamp to 0)."""
    p = Oracle.default_params(solver, double, kernel_set)
    real = np.float64 if double else np.float32
    h = float(p["interactionRadius"][0])
    pos = scene.fluid_block(*lattice, h, real=real, jitter=0.02, spacing=ratio * h)
    return p, pos, np.zeros_like(pos)


def rel_err(a, b):
    """max |a-b| / max|b| (array-level relative error, robust near zero entries)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    scale = np.max(np.abs(b))
    if scale == 0:
        return float(np.max(np.abs(a - b)))
    return float(np.max(np.abs(a - b)) / scale)


def check_cell_tables(cs_a, ce_a, cs_b, ce_b):
    """cellStart bit-exact everywhere; cellEnd only where the cell is non-empty (stale elsewhere)."""
    np.testing.assert_array_equal(cs_a, cs_b)
    m = cs_b != 0xFFFFFFFF
    np.testing.assert_array_equal(ce_a[m], ce_b[m])
