"""What does publishing the hit lists cost the density kernel?  Same scene, with and without NRS_FLAG_NO_SHARED_LISTS.
usage (GPU box): python tools/density_publish_cost.py [config]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nereus_amd import capi, scene  # noqa: E402
from nereus_amd.params import default_params  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "NS"
p = default_params(0)
sc = scene.dam_break(scene.CONFIGS[cfg], h=float(p["interactionRadius"][0]), kpoly=float(p["kpoly"][0]))
for name, flags in (("shared lists", 0), ("no shared lists", capi.FLAG_NO_SHARED_LISTS)):
    s = capi.Solver(p, len(sc["pos"]), flags=flags)
    s.set_particles(sc["pos"], sc["vel"])
    s.set_boundaries(sc["bi"], sc["vbi"], update_grid=True)
    s.step(20)
    s.set_profiling(True)
    s.step(60)
    t = s.stage_ms()
    print(name, {k: round(v[0] / v[1], 4) for k, v in t.items()}, flush=True)
    s.close()
