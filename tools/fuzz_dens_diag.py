"""One SESPH fuzz seed (GPU box): density of the production kernels, the reference-order kernels and the CPU oracle side by side; prints the
slots where they differ with the particle's position and cell.  usage: python tools/fuzz_dens_diag.py <seed>"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from nereus_amd import capi
from fuzz_parity import make_scene
from tests.oracle_lib import SESPH, STOP_FORCES, Oracle

seed = int(sys.argv[1])
sc = make_scene(seed)
o = Oracle(sc["p"], sc["double"], sc["kset"], SESPH)
o.set_particles(sc["pos"], sc["vel"]); o.set_boundaries(sc["bi"], sc["vbi"], update_grid=False)
o.step(1, stop=STOP_FORCES)
res = {}
for ref in (False, True):
    s = capi.Solver(sc["p"], sc["n"], solver=sc["solver"], double=sc["double"], kernel_set=sc["kset"], reference_order=ref)
    s.set_particles(sc["pos"], sc["vel"]); s.set_boundaries(sc["bi"], sc["vbi"], update_grid=False)
    s.step_partial(capi.STAGE_FORCES)
    res[ref] = {k: s.get(k) for k in ("hash", "index", "dens", "forces", "sortedPos")}
    s.close()
print("seed", seed, "n", sc["n"], "grid", sc["gs"], "walls", sc["bi"] is not None, "nb", 0 if sc["bi"] is None else len(sc["bi"]))
print("hash equal (tiled, ref, oracle):", np.array_equal(res[False]["hash"], o.get("hash")), np.array_equal(res[True]["hash"], o.get("hash")))
print("index equal:", np.array_equal(res[False]["index"], o.get("index")), np.array_equal(res[True]["index"], o.get("index")))
od, of = o.get("dens"), o.get("forces")
for nm, ref in (("tiled", False), ("reforder", True)):
    d = res[ref]["dens"]
    bad = np.nonzero(~((d == od) | (np.isnan(d) & np.isnan(od))))[0]
    print(nm, "density differs from the oracle at", len(bad), "slots")
    for i in bad[:6]:
        print("   slot", i, "hash", res[ref]["hash"][i], "pos", res[ref]["sortedPos"][i], "dens", d[i], "oracle", od[i], "oracle sortedPos", o.get("sortedPos")[i])
if sc["bi"] is not None:
    b = np.asarray(sc["bi"])
    print("boundary particles: nonfinite", int((~np.isfinite(b)).any(axis=1).sum()), "min", np.nanmin(b[:, :3], axis=0), "max", np.nanmax(b[:, :3], axis=0))
print("fluid nonfinite rows:", np.nonzero((~np.isfinite(sc["pos"])).any(axis=1))[0], sc["pos"][(~np.isfinite(sc["pos"])).any(axis=1)])
