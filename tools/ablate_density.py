"""Ablation (GPU box): time the density stage alone (step_partial) under NEREUS_DBG_STOP modes, one process."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nereus_amd import capi, scene
from nereus_amd.params import default_params
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
p = default_params(0)
lat = scene.CONFIGS[cfg] if cfg in scene.CONFIGS else tuple(int(v) for v in cfg.split(","))
sc = scene.dam_break(lat, h=float(p["interactionRadius"][0]), kpoly=float(p["kpoly"][0]))
n = len(sc["pos"])
fast = int(os.environ.get("NEREUS_ABLATE_FAST", "0"))
for ref in (False,) if os.environ.get("NEREUS_ABLATE_NOREF") else (False, True):
    s = capi.Solver(p, n, reference_order=ref, flags=capi.FLAG_FAST_ARITH if fast else 0)
    if os.environ.get("NEREUS_ABLATE_NOBOUND"):
        from nereus_amd.params import update_grid
        bi = sc["bi"]
        s.set_params(update_grid(p.copy(), bi[:, :3].min(0), bi[:, :3].max(0)))
    else:
        s.set_boundaries(sc["bi"], sc["vbi"], True)
    for mode in [0]:
        os.environ["NEREUS_DBG_STOP"] = str(mode)
        ts = []
        for it in range(4):
            s.set_particles(sc["pos"], sc["vel"])
            s.set_profiling([capi.STAGE_DENSITY, capi.STAGE_FORCES])
            s.step_partial(capi.STAGE_FORCES if mode == 0 else capi.STAGE_DENSITY)
            t = s.stage_ms()
            ts.append((t["density"][0], t.get("forces", (0, 0))[0]))
        d = s.get("dens")
        print("%s n=%d ref=%s fast=%d density %.3f ms forces %.3f ms (dens stat mean %.3f) unstaged %s" % (cfg, n, ref, fast, min(a for a, b in ts), min(b for a, b in ts), float(d.mean()), "-"))
    s.close()
os.environ.pop("NEREUS_DBG_STOP")
