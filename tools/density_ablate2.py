"""Timing of the density and force stages alone on a FIXED state (GPU box), for A/B and ablation builds of the library
(NEREUS_HIP_LIB): `density_ablate2.py save N path` runs N steps of the NS scene and saves the state; `density_ablate2.py time path|rest`
uploads it and times step_partial(FORCES) (production kernels with shared hit lists) a few times."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nereus_amd import capi, scene
from nereus_amd.params import default_params

mode, arg = sys.argv[1], sys.argv[-1]
p = default_params(0)
if os.environ.get("NEREUS_ABL_DT"):   # time step of the run that produces the saved state (round 3: 2.5e-4, the bench's)
    p["timestep"][0] = float(os.environ["NEREUS_ABL_DT"])
sc = scene.dam_break(scene.CONFIGS["NS"], h=float(p["interactionRadius"][0]), kpoly=float(p["kpoly"][0]))
n = len(sc["pos"])
s = capi.Solver(p, n)
s.set_particles(sc["pos"], sc["vel"])
s.set_boundaries(sc["bi"], sc["vbi"], True)
if mode == "save":
    s.step(int(sys.argv[2]))
    pos, vel = s.download()
    np.savez(arg, pos=pos, vel=vel)
    print("saved", arg, "after", sys.argv[2], "steps")
else:
    pos, vel = (sc["pos"], sc["vel"]) if arg == "rest" else (lambda z: (z["pos"], z["vel"]))(np.load(arg))
    ts = []
    for it in range(5):
        s.set_particles(pos, vel)
        s.set_profiling([capi.STAGE_DENSITY, capi.STAGE_FORCES])
        s.step_partial(capi.STAGE_FORCES)
        t = s.stage_ms()
        ts.append((t["density"][0], t["forces"][0]))
    print("%-28s %-6s density %.3f ms   forces %.3f ms" % (os.path.basename(os.environ.get("NEREUS_HIP_LIB", "main")), os.path.basename(arg)[:6],
                                                           min(a for a, b in ts), min(b for a, b in ts)))
    if os.environ.get("NEREUS_ABL_ALL"):
        print("   per evaluation (density, forces) ms:", " ".join("(%.3f, %.3f)" % t for t in ts))
