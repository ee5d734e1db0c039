"""Neighbour-count statistics along a run (how many particles exceed the 20-entry hit lists and take the overflow path?)
usage (GPU box): python tools/diag_neighbours.py [config] [steps ...]"""
import os
import sys

import numpy as np
from scipy.spatial import cKDTree

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nereus_amd import capi, scene  # noqa: E402
from nereus_amd.params import default_params  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
    marks = [int(v) for v in sys.argv[2:]] or [20, 120]
    p = default_params(0)
    h = float(p["interactionRadius"][0])
    sc = scene.dam_break(scene.CONFIGS[cfg], h=h, kpoly=float(p["kpoly"][0]))
    s = capi.Solver(p, len(sc["pos"]), solver=capi.SESPH)
    s.set_particles(sc["pos"], sc["vel"])
    s.set_boundaries(sc["bi"], sc["vbi"], update_grid=True)
    btree = cKDTree(sc["bi"][:, :3].astype(np.float64))
    done = 0
    for m in marks:
        s.step(m - done)
        done = m
        pos = s.download()[0][:, :3].astype(np.float64)
        t = cKDTree(pos)
        nf = t.query_ball_point(pos, h * (1 - 1e-7), return_length=True, workers=-1) - 1
        nb = btree.query_ball_point(pos, h * (1 - 1e-7), return_length=True, workers=-1)
        tot = nf + nb
        print("step %d: fluid neighbours mean %.2f max %d | +boundary mean %.2f max %d | > 20: %.4f %%  (waves with one: ~%.2f %%)"
              % (m, nf.mean(), nf.max(), tot.mean(), tot.max(), 100.0 * np.mean(tot > 20),
                 100.0 * (1 - (1 - np.mean(tot > 20)) ** 64)), flush=True)


if __name__ == "__main__":
    main()
