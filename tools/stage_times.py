"""Print per-stage HIP-event times for a scene (GPU box)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nereus_amd import capi, scene
from nereus_amd.params import default_params
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
SOLVER = 1 if os.environ.get("IISPH") else 0
p = default_params(SOLVER)
lat = scene.CONFIGS[cfg] if cfg in scene.CONFIGS else tuple(int(v) for v in cfg.split(","))
sc = scene.dam_break(lat, h=float(p["interactionRadius"][0]), kpoly=float(p["kpoly"][0]))
s = capi.Solver(p, len(sc["pos"]), solver=SOLVER, reference_order=bool(os.environ.get("REF")))
s.set_particles(sc["pos"], sc["vel"]); s.set_boundaries(sc["bi"], sc["vbi"], True)
if os.environ.get("GRIDX"):
    q = s.params
    q["gridSize"][0][0] = int(os.environ["GRIDX"]); q["numCells"][0] = int(np.prod(q["gridSize"][0].astype(np.int64)))
    s.set_params(q); s.set_boundaries(sc["bi"], sc["vbi"], False)
    print("grid", q["gridSize"][0], "cells 2^%d" % int(np.log2(q["numCells"][0])))
s.step(5); s.set_profiling(True); s.step(steps); s.synchronize()
t = s.stage_ms()
print(cfg, "dbg", os.environ.get("NEREUS_DBG_STOP"), {k: round(v[0] / v[1], 4) for k, v in t.items()}, "total/step", round(sum(v[0] for v in t.values()) / steps, 4))
if not os.environ.get("NEREUS_DBG_STOP"):
    d = s.get("dens"); print("dens mean", d.mean())
else:
    s.set_particles(sc["pos"], sc["vel"]); s.step_partial(capi.STAGE_DENSITY); d = s.get("dens"); print("dbg dens mean", d.mean(), "min", d.min(), "max", d.max())
