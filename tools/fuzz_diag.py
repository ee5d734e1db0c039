"""Diagnose one fuzz seed (GPU box): IISPH stage arrays of the list kernels vs the reference-order kernels vs the CPU oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from nereus_amd import capi
from fuzz_parity import make_scene
from tests.oracle_lib import Oracle, IISPH, STOP_I_PFORCE
seed = int(sys.argv[1])
sc = make_scene(seed)
names = ["hash", "index", "dens", "velAdv", "forcesAdv", "diiFluid", "diiBoundary", "densAdv", "aii", "sumDij", "densCorr", "P_l", "pres", "forcesP"]
res = {}
for ref in (False, True):
    s = capi.Solver(sc["p"], sc["n"], solver=sc["solver"], double=sc["double"], kernel_set=sc["kset"], reference_order=ref)
    s.set_particles(sc["pos"], sc["vel"]); s.set_boundaries(sc["bi"], sc["vbi"], update_grid=False)
    s.step_partial(capi.STAGE_I_PFORCE)
    res[ref] = {nm: s.get(nm) for nm in names}; res[ref]["iters"] = s.last_iterations
    if not ref:
        try: print("stats: overflow", s.get_stat(capi.STAT_HIT_OVERFLOW), "mean", s.get_stat(capi.STAT_HIT_MEAN), "max", s.get_stat(capi.STAT_HIT_MAX))
        except Exception as e: print("stats n/a", e)
    s.close()
o = Oracle(sc["p"], sc["double"], sc["kset"], IISPH)
o.set_particles(sc["pos"], sc["vel"]); o.set_boundaries(sc["bi"], sc["vbi"], update_grid=False)
o.step(1, stop=STOP_I_PFORCE)
print("iters tiled/ref/oracle", res[False]["iters"], res[True]["iters"], o.last_iters)
for nm in names:
    a, b, c = res[False][nm], res[True][nm], o.get(nm)
    d_ab = int((~((a == b) | (np.isnan(a) & np.isnan(b)))).sum())
    def rel(x, y):
        x = np.asarray(x, np.float64); y = np.asarray(y, np.float64); m = np.isfinite(x) & np.isfinite(y)
        sc_ = np.abs(y[m]).max() if m.any() else 1.0
        return float(np.abs(x[m] - y[m]).max() / (sc_ or 1.0)) if m.any() else 0.0
    print("%-12s tiled!=ref at %6d places | tiled vs oracle %.2e | ref vs oracle %.2e" % (nm, d_ab, rel(a, c), rel(b, c)))
    if d_ab and nm not in ("hash", "index") and "--all" not in sys.argv:
        bad = np.argwhere(~((a == b) | (np.isnan(a) & np.isnan(b))))[:, 0]
        i = int(bad[0])
        print("   first differing slot", i, "tiled", a[i], "ref", b[i], "oracle", c[i])
        break
