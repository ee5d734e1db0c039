"""libnereus_refshim.so (include/nereus_refshim.h): the reference's own launcher names (sph/sph.cuh) on caller-owned HIP pointers.

CPU: the library loads and exports every function the header declares.  GPU: a caller written against the reference's launcher layer
— it owns every device array and sequences the stages itself, exactly as SPH::update (sph/sph.cpp:233-284) and IISPH::update
(sph/iisph/iisph.cpp:172-216) do — runs the SESPH and IISPH steps through these entry points; results against the oracle and against
nrs_step of a reference-order context."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "nereus_amd", "libnereus_refshim.so")
HEADER = os.path.join(ROOT, "include", "nereus_refshim.h")


def declared():
    txt = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    return sorted(set(re.findall(r"^\s*(?:void|nrs_sreal|SUint_t|nrs_vec3|nrs_vec4)\s+\**(\w+)\s*\(", txt, flags=re.M)))


def test_refshim_exports_every_declared_symbol():
    names = declared()
    assert len(names) >= 19 and "calcHash" in names and "pressureSolve" in names and "computeDensityPressure" in names
    for suffix in ("", "_f64", "_monaghan", "_f64_monaghan"):   # one library per DOUBLE_PRECISION x KERNEL_SET of the reference
        lib = C.CDLL(LIB.replace(".so", suffix + ".so"))
        for n in names:
            assert hasattr(lib, n), (suffix, n)


def vec_types(double):
    ct = C.c_double if double else C.c_float

    class V3(C.Structure):
        _fields_ = [("x", ct), ("y", ct), ("z", ct)]

    class V4(C.Structure):
        _fields_ = [("x", ct), ("y", ct), ("z", ct), ("w", ct)]

    return ct, V3, V4


class Dev:
    """a device array owned by the CALLER, allocated through the shim's allocateArray (hipMalloc)"""

    def __init__(self, lib, nbytes, host=None):
        self.lib, self.nbytes = lib, int(nbytes)
        self.p = C.c_void_p()
        lib.allocateArray(C.byref(self.p), C.c_size_t(max(16, self.nbytes)))
        if host is not None:
            self.put(host)

    def put(self, a):
        a = np.ascontiguousarray(a)
        assert a.nbytes <= self.nbytes
        self.lib.copyArrayToDevice(self.p, a.ctypes.data_as(C.c_void_p), 0, a.nbytes)

    def get(self, dtype, shape):
        a = np.empty(shape, dtype)
        self.lib.copyArrayFromDevice(a.ctypes.data_as(C.c_void_p), self.p, None, a.nbytes)
        return a

    def free(self):
        if self.p:
            self.lib.freeArray(self.p)
            self.p = None


def load(double=False, kset=1):
    lib = C.CDLL(LIB.replace(".so", ("_f64" if double else "") + ("_monaghan" if kset == 0 else "") + ".so"))
    ct, V3, V4 = vec_types(double)
    vp = C.c_void_p
    lib.allocateArray.argtypes = [C.POINTER(vp), C.c_size_t]
    lib.freeArray.argtypes = [vp]
    lib.copyArrayToDevice.argtypes = [vp, vp, C.c_int, C.c_int]
    lib.copyArrayFromDevice.argtypes = [vp, vp, vp, C.c_int]
    lib.setParameters.argtypes = [vp]
    lib.integrateSystem.argtypes = [vp, vp, vp, ct, C.c_uint]
    lib.calcHash.argtypes = [vp, vp, vp, C.c_int]
    lib.sortParticles.argtypes = [vp, vp, C.c_uint]
    lib.reorderDataAndFindCellStartDBoundary.argtypes = [vp] * 8 + [C.c_uint, C.c_uint]
    lib.reorderDataAndFindCellStart.argtypes = [vp] * 16 + [C.c_uint, C.c_uint]
    lib.computeDensityPressure.argtypes = [vp] * 14 + [C.c_uint, C.c_uint, C.c_uint]
    lib.predictAdvection.argtypes = [vp] * 26 + [C.c_uint, C.c_uint, C.c_uint]
    lib.pressureSolve.argtypes = [vp] * 26 + [C.c_uint, C.c_uint, C.c_uint]
    lib.BBMin.argtypes = [vp, C.c_uint]
    lib.BBMin.restype = V3
    lib.BBMax.argtypes = [vp, C.c_uint]
    lib.BBMax.restype = V3
    lib.maxDensity.argtypes = [vp, C.c_uint]
    lib.maxDensity.restype = ct
    lib.maxVelocity.argtypes = [vp, C.c_uint]
    lib.maxVelocity.restype = V4
    lib.nrs_refshim_last_iterations.restype = C.c_uint
    return lib


class RefCaller:
    """What the reference's sph.cpp / iisph.cpp do with the launcher layer, written against the same entry points."""

    def __init__(self, lib, params, pos, vel, bi, vbi, iisph=False, double=False):
        self.lib, self.iisph = lib, iisph
        self.real = np.float64 if double else np.float32
        self.p = np.array(params).copy()
        self.n, self.nb = len(pos), 0 if bi is None else len(bi)
        self.cells = int(self.p["numCells"][0])
        n, V, S, U = self.n, (32 if double else 16), (8 if double else 4), 4
        d = lambda nbytes, host=None: Dev(lib, nbytes, host)
        self.pos, self.vel = d(n * V, pos.astype(self.real)), d(n * V, vel.astype(self.real))
        self.pres = d(n * S, np.zeros(n, self.real))
        self.sPos, self.sVel, self.sDens, self.sPres, self.sForces = d(n * V), d(n * V), d(n * S), d(n * S), d(n * V)
        self.hash, self.index = d(n * U), d(n * U)
        self.cs, self.ce = d(self.cells * U), d(self.cells * U)
        self.bcs, self.bce = d(self.cells * U), d(self.cells * U)
        nb = max(1, self.nb)
        self.bPos, self.bVbi = d(nb * V), d(nb * S)
        self.sbPos, self.sbVbi, self.bHash, self.bIndex = d(nb * V), d(nb * S), d(nb * U), d(nb * U)
        lib.setParameters(self.p.ctypes.data_as(C.c_void_p))
        if self.nb:   # SPH::updateGpuBoundaries (sph.cpp:391-432)
            self.bPos.put(bi.astype(self.real)); self.bVbi.put(vbi.astype(self.real))
            lib.calcHash(self.bHash.p, self.bIndex.p, self.bPos.p, self.nb)
            lib.sortParticles(self.bHash.p, self.bIndex.p, self.nb)
            lib.reorderDataAndFindCellStartDBoundary(self.bcs.p, self.bce.p, self.sbPos.p, self.sbVbi.p, self.bHash.p, self.bIndex.p, self.bPos.p,
                                                     self.bVbi.p, self.nb, self.cells)
        else:
            self.bcs.put(np.full(self.cells, 0xFFFFFFFF, np.uint32))
        if iisph:
            self.extra = {k: d(n * (V if k in ("velAdv", "forcesAdv", "forcesP", "diiF", "diiB", "sumDij", "normal") else S),
                               np.zeros(n * (4 if k in ("velAdv", "forcesAdv", "forcesP", "diiF", "diiB", "sumDij", "normal") else 1), self.real))
                          for k in ("densAdv", "densCorr", "P_l", "prevP", "aii", "velAdv", "forcesAdv", "forcesP", "diiF", "diiB", "sumDij", "normal")}

    def update(self):
        lib, n = self.lib, self.n
        lib.setParameters(self.p.ctypes.data_as(C.c_void_p))                      # sph.cpp:236
        lib.calcHash(self.hash.p, self.index.p, self.pos.p, n)                    # :238
        lib.sortParticles(self.hash.p, self.index.p, n)                          # :240
        lib.reorderDataAndFindCellStart(self.cs.p, self.ce.p, self.sPos.p, self.sVel.p, None, self.sPres.p, None, None, self.hash.p, self.index.p,
                                        self.pos.p, self.vel.p, None, self.pres.p, None, None, n, self.cells)   # :242-260
        if not self.iisph:
            # SESPH kernels index the UNSORTED boundary arrays through gridBoundaryIndex (sph_kernel_impl.cuh:341-344)
            lib.computeDensityPressure(self.sPos.p, self.sVel.p, self.sDens.p, self.sPres.p, self.sForces.p, None, self.bPos.p, self.bVbi.p,
                                       self.index.p, self.cs.p, self.ce.p, self.bIndex.p, self.bcs.p, self.bce.p, n, self.cells, self.nb)   # :262-278
            lib.integrateSystem(self.sPos.p, self.sVel.p, self.sForces.p, float(self.p["timestep"][0]), n)                                 # :280
        else:
            e = self.extra
            args = [self.sPos.p, self.sVel.p, self.sDens.p, self.sPres.p, self.sForces.p, None, self.cs.p, self.ce.p, self.index.p, self.sbPos.p,
                    self.sbVbi.p, self.bcs.p, self.bce.p, self.bIndex.p, e["densAdv"].p, e["densCorr"].p, e["P_l"].p, e["prevP"].p, e["aii"].p,
                    e["velAdv"].p, e["forcesAdv"].p, e["forcesP"].p, e["diiF"].p, e["diiB"].p, e["sumDij"].p, e["normal"].p, n, self.nb, self.cells]
            lib.predictAdvection(*args)                                           # iisph.cpp:201-206
            lib.pressureSolve(*args)                                              # :207-212
            self.pres_swap()
        # the reference copies sorted -> host -> device (sph.cpp:283-284, 233-234); here device to device through the host arrays
        self.host_pos, self.host_vel = self.sPos.get(self.real, (n, 4)), self.sVel.get(self.real, (n, 4))
        self.pos.put(self.host_pos); self.vel.put(self.host_vel)

    def pres_swap(self):   # iisph.cpp:216: m_pressure <- sorted pressures, the next step's warm start
        self.pres.put(self.sPres.get(self.real, (self.n,)))


@pytest.mark.gpu
@pytest.mark.parametrize("double,kset", [(False, 1), (True, 0)], ids=["f32-muller", "f64-monaghan"])
def test_refshim_sesph_chain_equals_context_and_oracle(hip_lib, double, kset):
    from nereus_amd import capi
    from tests.common import rel_err, small_dam_break
    from tests.oracle_lib import SESPH, Oracle

    real = np.float64 if double else np.float32
    p, sc = small_dam_break((16, 14, 12), double=double, kernel_set=kset)
    o = Oracle(p, double, kset, SESPH)
    o.set_particles(sc["pos"], sc["vel"])
    o.set_boundaries(sc["bi"], sc["vbi"], update_grid=True)
    lib = load(double, kset)
    c = RefCaller(lib, o.params, sc["pos"], sc["vel"], sc["bi"], sc["vbi"], double=double)
    s = capi.Solver(p, len(sc["pos"]), double=double, kernel_set=kset, reference_order=True, flags=capi.FLAG_FULL_SORT | capi.FLAG_NO_FUSION)
    s.set_particles(sc["pos"], sc["vel"])
    s.set_boundaries(sc["bi"], sc["vbi"], update_grid=True)
    # the boundary tables the caller built through the shim == the context's
    np.testing.assert_array_equal(c.bHash.get(np.uint32, (c.nb,)), s.get("bhash"))
    np.testing.assert_array_equal(c.bIndex.get(np.uint32, (c.nb,)), s.get("bindex"))
    np.testing.assert_array_equal(c.sbPos.get(real, (c.nb, 4))[:, :3], s.get("bSorted")[:, :3])
    np.testing.assert_array_equal(c.sbVbi.get(real, (c.nb,)), s.get("bSorted")[:, 3])
    mn, mx = lib.BBMin(c.sbPos.p, c.nb), lib.BBMax(c.sbPos.p, c.nb)           # sph.cpp:313-337 computes the grid from these
    assert (mn.x, mn.y, mn.z) == tuple(sc["bi"][:, :3].min(0)) and (mx.x, mx.y, mx.z) == tuple(sc["bi"][:, :3].max(0))
    for _ in range(3):
        c.update(); s.step(1); o.step(1)
        np.testing.assert_array_equal(c.hash.get(np.uint32, (c.n,)), o.get("hash"))
        np.testing.assert_array_equal(c.index.get(np.uint32, (c.n,)), o.get("index"))
        np.testing.assert_array_equal(c.sDens.get(real, (c.n,)), s.get("dens"))
        gp, gv = s.download()
        np.testing.assert_array_equal(c.host_pos, gp)       # the same kernels behind both interfaces: bit for bit
        np.testing.assert_array_equal(c.host_vel, gv)
        assert rel_err(c.host_pos[:, :3], o.get("pos")[:, :3]) <= 1e-5 and rel_err(c.host_vel[:, :3], o.get("vel")[:, :3]) <= 1e-5
    assert abs(lib.maxDensity(c.sDens.p, c.n) - float(s.get("dens").max())) == 0.0
    v = lib.maxVelocity(c.sVel.p, c.n)
    assert abs(np.sqrt(v.x * v.x + v.y * v.y + v.z * v.z) - np.linalg.norm(c.host_vel[:, :3].astype(np.float64), axis=1).max()) <= 1e-6


@pytest.mark.gpu
def test_refshim_iisph_chain_equals_oracle(hip_lib):
    from tests.common import compressed_block, rel_err
    from tests.oracle_lib import IISPH, Oracle

    p, pos, vel = compressed_block()
    o = Oracle(p, solver=IISPH)
    o.set_particles(pos, vel)
    o.set_boundaries(None, None)
    lib = load()
    c = RefCaller(lib, o.params, pos, vel, None, None, iisph=True)
    for _ in range(3):
        c.update(); o.step(1)
        assert lib.nrs_refshim_last_iterations() == o.last_iters
        np.testing.assert_array_equal(c.hash.get(np.uint32, (c.n,)), o.get("hash"))
        np.testing.assert_array_equal(c.index.get(np.uint32, (c.n,)), o.get("index"))
        assert rel_err(c.host_pos[:, :3], o.get("pos")[:, :3]) <= 1e-5
        assert rel_err(c.host_vel[:, :3], o.get("vel")[:, :3]) <= 1e-5
        assert rel_err(c.sPres.get(np.float32, (c.n,)), o.get("pressure")) <= 1e-4   # (the IISPH test runs the fp32 Muller shim)
