import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nereus_amd import capi
from tests.oracle_lib import SESPH, STOP_FORCES, Oracle
p = Oracle.default_params(SESPH)
h = float(p["interactionRadius"][0])
rng = np.random.default_rng(3)
for K in (1, 2):
    shown = 0
    for t in range(200):
        pos = np.array([[0.1, 0.2, 0.3, 1.0]], np.float32)
        vel = np.array([[0.3, -0.2, 0.1, 0.0]], np.float32)
        bi = np.ones((K, 4), np.float32)
        bi[:, :3] = pos[0, :3] + rng.uniform(-0.9 * h, 0.9 * h, (K, 3)).astype(np.float32)
        vbi = rng.uniform(1e-5, 2e-5, K).astype(np.float32)
        if K != 2:
            continue
        o = Oracle(p, solver=SESPH); o.set_particles(pos, vel); o.set_boundaries(bi, vbi, False)
        s = capi.Solver(p, 4, reference_order=True); s.set_particles(pos, vel); s.set_boundaries(bi, vbi, False)
        o.step(1, stop=STOP_FORCES); s.step_partial(capi.STAGE_FORCES)
        d1, d2 = s.get("dens")[0], o.get("dens")[0]
        if d1 != d2 and shown < 4:
            shown += 1
            print("t", t, "gpu %r cpu %r" % (d1, d2))
            print("  bhash gpu", s.get("bhash"), "cpu", o.get("bhash"), " bindex gpu", s.get("bindex"), "cpu", o.get("bindex"))
            bs = s.get("bSorted"); print("  bSorted eq:", np.array_equal(bs[:, :3], o.get("sbi")[:, :3]), np.array_equal(bs[:, 3], o.get("svbi")))
            cs, ce = s.get("bCellStart"), s.get("bCellEnd"); ocs, oce = o.get("bCellStart"), o.get("bCellEnd")
            nz = np.nonzero(cs != 0xFFFFFFFF)[0]; onz = np.nonzero(ocs != 0xFFFFFFFF)[0]
            print("  gpu cells", nz, cs[nz], ce[nz], " cpu cells", onz, ocs[onz], oce[onz])
            # same scene with a second fluid particle far away to see if result depends on anything else
            s2 = capi.Solver(p, 4, reference_order=False); s2.set_particles(pos, vel); s2.set_boundaries(bi, vbi, False)
            s2.step_partial(capi.STAGE_FORCES); print("  tiled gpu %r" % s2.get("dens")[0])
        s.close()
