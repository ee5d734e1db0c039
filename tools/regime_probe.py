"""Regime of the north-star dam-break over time (GPU box): every CHUNK steps print ms/step of the chunk, neighbours per particle,
mover fraction, full-sort fall-backs, max |v| and max density.  Documents what `developed` means at which step.
usage: python tools/regime_probe.py [config] [total steps] [chunk] [dt]   (dt: fixed time step in seconds, default = the reference's 1e-3)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nereus_amd import capi, scene
from nereus_amd.params import default_params

cfg = sys.argv[1] if len(sys.argv) > 1 else "NS"
total = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 250
p = default_params(0)
if len(sys.argv) > 4:
    p["timestep"][0] = float(sys.argv[4])
lat = scene.CONFIGS[cfg]
sc = scene.dam_break(lat, h=float(p["interactionRadius"][0]), kpoly=float(p["kpoly"][0]))
n = len(sc["pos"])
s = capi.Solver(p, n)
s.set_particles(sc["pos"], sc["vel"])
s.set_boundaries(sc["bi"], sc["vbi"], True)
h = float(p["interactionRadius"][0]); dt = float(p["timestep"][0]); cs = float(p["soundSpeed"][0])
done, fb0 = 0, 0
rows = []
while done < total:
    s.synchronize(); t0 = time.perf_counter()
    s.step(chunk)
    s.synchronize(); ms = 1e3 * (time.perf_counter() - t0) / chunk
    done += chunk
    st, fb = s.resort_stats()
    vmax = s.max_velocity()
    row = dict(step=done, t=round(done * dt, 4), ms_per_step=round(ms, 3), neighbours_mean=round(s.get_stat(capi.STAT_HIT_MEAN), 2),
               neighbours_max=s.get_stat(capi.STAT_HIT_MAX), overflow=s.get_stat(capi.STAT_HIT_OVERFLOW) / n,
               movers=round(s.get_stat(capi.STAT_MOVERS) / n, 4), full_sort_fallbacks_in_chunk=fb - fb0,
               vmax=round(vmax, 2), cells_per_step=round(vmax * dt / h, 3), cfl_dt_limit=round(0.4 * h / (cs + vmax), 6))
    fb0 = fb
    rows.append(row)
    print(json.dumps(row), flush=True)
