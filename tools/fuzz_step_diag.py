"""One fuzz seed, production kernels against reference-order kernels STEP BY STEP (GPU box): hash / index / positions after every step,
to find where two runs that agree on a force evaluation part ways.  usage: python tools/fuzz_step_diag.py <seed> [steps]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from nereus_amd import capi
from fuzz_parity import make_scene

seed = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sc = make_scene(seed)
print("seed", seed, "n", sc["n"], "grid", sc["gs"], "solver", sc["solver"], "walls", sc["bi"] is not None, "nonfinite positions", int((~np.isfinite(sc["pos"])).any(axis=1).sum()))
S = []
for ref in (False, True):
    s = capi.Solver(sc["p"], sc["n"], solver=sc["solver"], double=sc["double"], kernel_set=sc["kset"], reference_order=ref)
    vel = sc["vel"].copy(); vel[:, 3] = np.arange(sc["n"])   # ids (SESPH keeps w)
    s.set_particles(sc["pos"], vel); s.set_boundaries(sc["bi"], sc["vbi"], update_grid=False)
    S.append(s)
for k in range(steps):
    outs = []
    for s in S:
        s.step(1)
        outs.append((s.get("hash"), s.get("index")) + s.download())
    (ha, ia, pa, va), (hb, ib, pb, vb) = outs
    eq = lambda x, y: np.array_equal(x, y, equal_nan=True)
    print("step", k + 1, "hash", eq(ha, hb), "index", eq(ia, ib), "pos", eq(pa, pb), "vel", eq(va, vb), "| ids equal", eq(va[:, 3], vb[:, 3]))
    if not eq(ha, hb):
        d = np.nonzero(ha != hb)[0]
        print("  first hash difference at slot", d[0], ha[d[0] - 2:d[0] + 3], hb[d[0] - 2:d[0] + 3], "count", len(d))
    if not eq(pa, pb):
        oa, ob = np.argsort(va[:, 3], kind="stable"), np.argsort(vb[:, 3], kind="stable")
        bad = np.nonzero(~((pa[oa] == pb[ob]) | (np.isnan(pa[oa]) & np.isnan(pb[ob]))).all(axis=1))[0]
        print("  by id: particles whose position differs:", len(bad), bad[:10], "nonfinite among them", int((~np.isfinite(pa[oa][bad])).any(axis=1).sum()))
        if len(bad):
            print("  e.g.", pa[oa][bad[0]], pb[ob][bad[0]])
