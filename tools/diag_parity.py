"""Diagnostic (GPU box): bit-level mismatch statistics HIP vs oracle per stage."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nereus_amd import capi  # noqa: E402
from tests.common import default_scene, small_dam_break  # noqa: E402
from tests.oracle_lib import SESPH, STOP_FORCES, Oracle  # noqa: E402


def ulps(a, b):
    a = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return np.abs(a - b)


def report(tag, x, y):
    u = ulps(x, y)
    print("  %-8s mismatching %d / %d  max ulp %d  rel(max) %.3e" % (
        tag, int((u > 0).sum()), u.size, int(u.max()), float(np.abs(x.astype(np.float64) - y).max() / np.abs(y).max())))


for name in ("default", "dambreak"):
    if name == "default":
        p, pos, vel = default_scene(SESPH)
        bi = vbi = None
    else:
        p, sc = small_dam_break()
        pos, vel, bi, vbi = sc["pos"], sc["vel"], sc["bi"], sc["vbi"]
    for ref in (True, False):
        o = Oracle(p, solver=SESPH)
        o.set_particles(pos, vel)
        o.set_boundaries(bi, vbi, True)
        s = capi.Solver(p, len(pos), reference_order=ref)
        s.set_particles(pos, vel)
        s.set_boundaries(bi, vbi, True)
        o.step(1, stop=STOP_FORCES)
        s.step_partial(capi.STAGE_FORCES)
        print(name, "ref-order" if ref else "tiled")
        report("dens", s.get("dens"), o.get("dens"))
        report("pres", s.get("pres"), o.get("pres"))
        report("forces", s.get("forces")[:, :3], o.get("forces")[:, :3])
