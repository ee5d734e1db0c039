"""Long-run robustness soak (GPU box): random scenes of tools/fuzz_parity.py stepped 120 times on the production path — through
whatever the physics does (clumps explode, particles leave the grid, NaN appears) — checking after every 20 steps that the
device-side consistency guard stayed silent (nrs_synchronize), that the sorted keys are sorted and the index array is a permutation.
usage: python tools/fuzz_long.py [seeds=40] [first=0]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from nereus_amd import capi
from fuzz_parity import make_scene

def one(seed):
    sc = make_scene(seed)
    s = capi.Solver(sc["p"], sc["n"], solver=sc["solver"], double=sc["double"], kernel_set=sc["kset"])
    s.set_particles(sc["pos"], sc["vel"]); s.set_boundaries(sc["bi"], sc["vbi"], update_grid=False)
    try:
        for chunk in range(6):
            s.step(20)
            s.synchronize()                      # raises on a device-side guard
            h, idx = s.get("hash"), s.get("index")
            if not np.all(np.diff(h.astype(np.int64)) >= 0):
                return "seed %d: keys not sorted after %d steps" % (seed, 20 * (chunk + 1))
            if not np.array_equal(np.sort(idx), np.arange(sc["n"], dtype=np.uint32)):
                return "seed %d: index not a permutation after %d steps" % (seed, 20 * (chunk + 1))
    except Exception as e:
        return "seed %d: %s" % (seed, str(e)[:200])
    finally:
        s.close()
    return None

if __name__ == "__main__":
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    fails = 0
    for sd in range(first, first + seeds):
        r = one(sd)
        if r:
            print(r, flush=True); fails += 1
            break                                  # (stop at the first problem: do not pile faults up)
        if (sd - first) % 10 == 9: print("... %d seeds done" % (sd - first + 1), flush=True)
    print("long-run soak: %d failures" % fails)
    sys.exit(1 if fails else 0)
