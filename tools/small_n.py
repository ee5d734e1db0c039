"""Wall time per step at small particle counts (launch-bound regime), GPU box."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nereus_amd import capi, scene
from nereus_amd.params import default_params
for lat in ((11, 11, 11), (32, 32, 32), (53, 53, 53), (100, 100, 100)):
    for solver in (0, 1):
        p = default_params(solver)
        sc = scene.dam_break(lat, h=float(p["interactionRadius"][0]), kpoly=float(p["kpoly"][0]))
        s = capi.Solver(p, len(sc["pos"]), solver=solver)
        s.set_particles(sc["pos"], sc["vel"]); s.set_boundaries(sc["bi"], sc["vbi"], True)
        s.step(20); s.synchronize()
        t0 = time.perf_counter(); s.step(200); s.synchronize(); dt = time.perf_counter() - t0
        print("n=%8d %s  %.1f us/step" % (len(sc["pos"]), "IISPH" if solver else "SESPH", 1e6 * dt / 200), flush=True)
        s.close()
