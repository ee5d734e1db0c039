"""How many particles change grid cell in one step?  (sizing of the coherent re-sort: stayers keep their order)
usage (GPU box): python tools/diag_movers.py [config] [steps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nereus_amd import capi, scene  # noqa: E402
from nereus_amd.params import default_params  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    lattice = scene.CONFIGS[cfg]
    p = default_params(0)
    sc = scene.dam_break(lattice, h=float(p["interactionRadius"][0]), kpoly=float(p["kpoly"][0]))
    s = capi.Solver(p, len(sc["pos"]), solver=capi.SESPH)
    s.set_particles(sc["pos"], sc["vel"])
    s.set_boundaries(sc["bi"], sc["vbi"], update_grid=True)
    s.step(1)
    prev = s.get("hash").copy()
    out = []
    for k in range(steps):
        s.step(1)
        h = s.get("hash")
        idx = s.get("index")
        # sorted slot j of this step held slot idx[j] of the previous one; its previous hash was prev[idx[j]]
        moved = int(np.count_nonzero(prev[idx] != h))
        out.append(moved / len(h))
        prev = h.copy()
        if k % 20 == 0:
            print("step %d movers %.4f" % (k, out[-1]), flush=True)
    print("mean %.4f max %.4f" % (np.mean(out), np.max(out)))


if __name__ == "__main__":
    main()
