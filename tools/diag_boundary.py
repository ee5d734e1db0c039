"""Diagnostic: single fluid particle + K boundary particles, bitwise density/forces comparison."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nereus_amd import capi  # noqa: E402
from tests.oracle_lib import SESPH, STOP_FORCES, Oracle  # noqa: E402

p = Oracle.default_params(SESPH)
h = float(p["interactionRadius"][0])
rng = np.random.default_rng(3)
for K in (1, 2, 5, 20):
    bad_d = bad_f = 0
    trials = 200
    for t in range(trials):
        pos = np.array([[0.1, 0.2, 0.3, 1.0]], np.float32)
        vel = np.array([[0.3, -0.2, 0.1, 0.0]], np.float32)
        bi = np.ones((K, 4), np.float32)
        bi[:, :3] = pos[0, :3] + rng.uniform(-0.9 * h, 0.9 * h, (K, 3)).astype(np.float32)
        vbi = rng.uniform(1e-5, 2e-5, K).astype(np.float32)
        o = Oracle(p, solver=SESPH)
        o.set_particles(pos, vel)
        o.set_boundaries(bi, vbi, False)
        s = capi.Solver(p, 4, reference_order=True)
        s.set_particles(pos, vel)
        s.set_boundaries(bi, vbi, False)
        o.step(1, stop=STOP_FORCES)
        s.step_partial(capi.STAGE_FORCES)
        d1, d2 = s.get("dens"), o.get("dens")
        f1, f2 = s.get("forces"), o.get("forces")
        if d1[0] != d2[0]:
            bad_d += 1
            if bad_d <= 3:
                print("  K=%d dens gpu %.9g cpu %.9g  nb within h: %d" % (K, d1[0], d2[0], int((np.linalg.norm(bi[:, :3] - pos[0, :3], axis=1) < h).sum())))
        if not np.array_equal(f1, f2):
            bad_f += 1
            if bad_f <= 3:
                print("  K=%d forces gpu %s cpu %s" % (K, f1[0], f2[0]))
        s.close()
    print("K=%d: dens mismatches %d/%d, forces mismatches %d/%d" % (K, bad_d, trials, bad_f, trials))
