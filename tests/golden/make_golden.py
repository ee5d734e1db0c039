"""Regenerates tests/golden/*.npz.

There is nothing to generate these from except our own oracle: the reference ships no tests, fixtures or
golden vectors, and its CUDA path cannot be built in this image (see oracle/nereus_oracle.cpp header).  The
fixtures therefore lock the ORACLE's behaviour (so a later edit to it is caught) and let the GPU tests
compare against committed numbers; the only reference-derived anchors are the known answers of
SURVEY.md §8c, asserted in tests/test_oracle_kat.py.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from tests.common import compressed_block, default_scene, small_dam_break  # noqa: E402
from tests.oracle_lib import (IISPH, SESPH, STOP_FORCES, STOP_I_PFORCE, Oracle)  # noqa: E402


def sesph_default():
    p, pos, vel = default_scene(SESPH)
    o = Oracle(p, solver=SESPH)
    o.set_particles(pos, vel)
    o.set_boundaries(None, None)
    o.step(1, stop=STOP_FORCES)
    out = dict(params=p.view(np.uint8), pos0=pos, vel0=vel, hash=o.get("hash"), index=o.get("index"),
               dens=o.get("dens"), pres=o.get("pres"), forces=o.get("forces"))
    cs = o.get("cellStart")
    nz = np.nonzero(cs != 0xFFFFFFFF)[0].astype(np.uint32)
    out.update(cell_ids=nz, cell_start=cs[nz], cell_end=o.get("cellEnd")[nz])
    o2 = Oracle(p, solver=SESPH)
    o2.set_particles(pos, vel)
    o2.set_boundaries(None, None)
    for steps in (1, 10):
        o2.step(steps - (0 if steps == 1 else 1))
        out["pos%d" % steps] = o2.get("pos")
        out["vel%d" % steps] = o2.get("vel")
    np.savez_compressed(os.path.join(HERE, "sesph_default.npz"), **out)


def sesph_dambreak():
    p, sc = small_dam_break()
    o = Oracle(p, solver=SESPH)
    o.set_particles(sc["pos"], sc["vel"])
    o.set_boundaries(sc["bi"], sc["vbi"], update_grid=True)
    o.step(1, stop=STOP_FORCES)
    out = dict(params=o.params.view(np.uint8), pos0=sc["pos"], bi=sc["bi"], vbi=sc["vbi"], hash=o.get("hash"),
               index=o.get("index"), dens=o.get("dens"), pres=o.get("pres"), forces=o.get("forces"),
               bhash=o.get("bhash"), bindex=o.get("bindex"))
    o2 = Oracle(p, solver=SESPH)
    o2.set_particles(sc["pos"], sc["vel"])
    o2.set_boundaries(sc["bi"], sc["vbi"], update_grid=True)
    o2.step(10)
    out["pos10"] = o2.get("pos")
    out["vel10"] = o2.get("vel")
    np.savez_compressed(os.path.join(HERE, "sesph_dambreak.npz"), **out)


def iisph_compressed():
    p, pos, vel = compressed_block()
    o = Oracle(p, solver=IISPH)
    o.set_particles(pos, vel)
    o.set_boundaries(None, None)
    o.step(1, stop=STOP_I_PFORCE)
    out = dict(params=p.view(np.uint8), pos0=pos, dens=o.get("dens"), aii=o.get("aii"), densAdv=o.get("densAdv"),
               diiFluid=o.get("diiFluid"), velAdv=o.get("velAdv"), sumDij=o.get("sumDij"), densCorr=o.get("densCorr"),
               P_l=o.get("P_l"), forcesP=o.get("forcesP"), iters=np.array([o.last_iters]))
    o2 = Oracle(p, solver=IISPH)
    o2.set_particles(pos, vel)
    o2.set_boundaries(None, None)
    o2.step(5)
    out["pos5"] = o2.get("pos")
    out["vel5"] = o2.get("vel")
    out["pressure5"] = o2.get("pressure")
    out["iters5"] = np.array([o2.last_iters])
    np.savez_compressed(os.path.join(HERE, "iisph_compressed.npz"), **out)


if __name__ == "__main__":
    sesph_default()
    sesph_dambreak()
    iisph_compressed()
    print("golden fixtures written to", HERE)
