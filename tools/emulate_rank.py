"""One rank of an N-rank weak-scaling run on a single GPU, without neighbours: the global grid (2^30 cells at N = 8), both
cuts active, message and ghost streams populated (the messages are packed but never sent, arrivals never come, so the
physics at the cuts is wrong — this is a memory / speed rehearsal of the per-rank work, not a correctness test).
usage (GPU box): python tools/emulate_rank.py [world] [rank] [steps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nereus_amd import scene, slab  # noqa: E402
from nereus_amd.params import default_params  # noqa: E402


def main():
    import torch

    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    rank = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    lattice = scene.CONFIGS["NS"]
    t0 = time.perf_counter()
    p, cuts, pos, vel, bi, vbi, info = slab.rank_scene(lattice, rank, world, default_params(0))
    print("scene %.1f s: %d particles, %d boundary, grid %s = 2^%.0f cells, cuts %s" % (
        time.perf_counter() - t0, len(pos), len(bi), tuple(int(v) for v in p["gridSize"][0]), np.log2(float(p["numCells"][0])),
        (cuts[rank], cuts[rank + 1])), flush=True)
    msg_cap, cap = slab.capacities(lattice, float(p["interactionRadius"][0]), len(pos))
    eng = slab.HipSlabEngine(p, cap, msg_cap, cuts[rank], cuts[rank + 1], 0)
    eng.load(pos, vel, bi, vbi)
    print("loaded; device memory in use %.1f GB" % ((torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 1e9), flush=True)
    send_l, send_r = eng.make_buffer(), eng.make_buffer()  # packed, never sent
    for phase, k in (("warm-up", 10), ("timed", steps)):
        eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            eng.pack(send_l, send_r)
            eng.unpack(None, None)
            counts = eng.last_counts()
            eng.step(1)
        eng.synchronize()
        dt = time.perf_counter() - t0
        print("%s: %.3f ms/step, local %d owned %d, stream counts %s, resort %s" % (
            phase, 1e3 * dt / k, eng.n_local, eng.n_owned, list(counts), eng.solver.resort_stats()), flush=True)


if __name__ == "__main__":
    main()
