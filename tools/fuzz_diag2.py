"""per-step comparison of full IISPH steps: production context vs reference-order context (GPU box)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from nereus_amd import capi
from fuzz_parity import make_scene
sc = make_scene(int(sys.argv[1]))
ss = []
for ref in (False, True):
    s = capi.Solver(sc["p"], sc["n"], solver=sc["solver"], double=sc["double"], kernel_set=sc["kset"], reference_order=ref)
    s.set_particles(sc["pos"], sc["vel"]); s.set_boundaries(sc["bi"], sc["vbi"], update_grid=False)
    ss.append(s)
for step in range(1, 4):
    outs = []
    for s in ss:
        s.step(1)
        p, v, pr = s.download(pressure=True)
        outs.append((p, v, pr, s.get("index"), s.last_iterations))
    a, b = outs
    def neq(x, y): return int((~((x == y) | (np.isnan(x) & np.isnan(y)))).sum())
    print("step", step, "iters", a[4], b[4], "pos!=", neq(a[0], b[0]), "vel!=", neq(a[1], b[1]), "pres!=", neq(a[2], b[2]), "index!=", neq(a[3], b[3]),
          "finite pos", np.isfinite(a[0]).all(), np.isfinite(b[0]).all())
