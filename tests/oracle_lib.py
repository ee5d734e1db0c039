"""ctypes wrapper around oracle/libnereus_oracle_*.so — the CPU checker (test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

from nereus_amd.params import params_dtype

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

SESPH, IISPH = 0, 1
STOP_HASH, STOP_SORT, STOP_REORDER, STOP_DENSITY, STOP_FORCES = 1, 2, 3, 4, 5
STOP_I_DENSITY, STOP_I_DISPLACEMENT, STOP_I_ADVECTION, STOP_I_SOLVE, STOP_I_PFORCE = 10, 11, 12, 13, 14

_U32 = {"hash", "index", "cellStart", "cellEnd", "bhash", "bindex", "bCellStart", "bCellEnd"}
_VEC4 = {"pos", "vel", "sortedPos", "sortedVel", "forces", "sbi", "velAdv", "forcesAdv", "forcesP",
         "diiFluid", "diiBoundary", "sumDij"}


def _lib_path(double, kernel_set):
    return os.path.join(ORACLE_DIR, "libnereus_oracle_d%dk%d.so" % (int(double), int(kernel_set)))


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "all"])


_libs = {}


def _load(double, kernel_set):
    key = (bool(double), int(kernel_set))
    if key in _libs:
        return _libs[key]
    path = _lib_path(*key)
    src = os.path.join(ORACLE_DIR, "nereus_oracle.cpp")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        build_oracle()
    lib = C.CDLL(path)
    lib.orc_create.restype = C.c_void_p
    lib.orc_create.argtypes = [C.c_void_p]
    lib.orc_destroy.argtypes = [C.c_void_p]
    lib.orc_set_params.argtypes = [C.c_void_p, C.c_void_p]
    lib.orc_get_params.argtypes = [C.c_void_p, C.c_void_p]
    lib.orc_set_mode.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.orc_set_self_by_slot.argtypes = [C.c_void_p, C.c_int]
    lib.orc_set_surface_tension.argtypes = [C.c_void_p, C.c_int]
    lib.orc_set_tait_mode.argtypes = [C.c_void_p, C.c_int]
    lib.orc_set_particles.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint]
    lib.orc_set_boundaries.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_int]
    lib.orc_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.orc_num_particles.restype = C.c_uint
    lib.orc_num_particles.argtypes = [C.c_void_p]
    lib.orc_last_iters.restype = C.c_uint
    lib.orc_last_iters.argtypes = [C.c_void_p]
    lib.orc_get.restype = C.c_long
    lib.orc_get.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
    lib.orc_default_params.argtypes = [C.c_int, C.c_void_p]
    lib.orc_recompute_constants.argtypes = [C.c_void_p]
    lib.orc_generate_cube.restype = C.c_uint
    lib.orc_generate_cube.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint]
    assert lib.orc_sizeof_real() == (8 if key[0] else 4)
    assert lib.orc_kernel_set() == key[1]
    assert lib.orc_sizeof_params() == params_dtype(key[0]).itemsize
    _libs[key] = lib
    return lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    """One simulated solver (mirrors the nrs_ctx API of the HIP library so tests read symmetrically)."""

    def __init__(self, params=None, double=False, kernel_set=1, solver=SESPH, threads=1, jacobi=True, self_by_slot=False,
                 surface_tension=True, tait="powf"):
        self.double, self.kernel_set, self.solver = bool(double), int(kernel_set), solver
        self.real = np.float64 if double else np.float32
        self.lib = _load(double, kernel_set)
        if params is None:
            params = self.default_params(solver, double, kernel_set)
        self._p = np.array(params, dtype=params_dtype(double)).reshape(1).copy()
        self.h = self.lib.orc_create(_ptr(self._p))
        self.lib.orc_set_mode(self.h, int(jacobi), int(threads))
        if self_by_slot:  # SURVEY Q5 switched off: order-independent IISPH (see nereus_oracle.cpp, Sim::selfBySlot)
            self.lib.orc_set_self_by_slot(self.h, 1)
        if not surface_tension:  # USE_SURFACE_TENSION=0 (CMakeLists.txt:28, sph_kernel_impl.cuh:535-548)
            self.lib.orc_set_surface_tension(self.h, 0)
        # Tait x^7: "powf" = glibc powf, what g++ gives the reference on the host (default);
        # "double7" = formed in double, rounded once to float, what the device evaluates (nrs_math.h pow7f)
        assert tait in ("powf", "double7")
        if tait == "double7":
            self.lib.orc_set_tait_mode(self.h, 1)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.orc_destroy(self.h)
                self.h = None
        except Exception:
            pass

    @staticmethod
    def default_params(solver=SESPH, double=False, kernel_set=1):
        lib = _load(double, kernel_set)
        p = np.zeros(1, dtype=params_dtype(double))
        lib.orc_default_params(int(solver), _ptr(p))
        return p

    @staticmethod
    def recompute_constants(p, double=False, kernel_set=1):
        lib = _load(double, kernel_set)
        p = np.array(p, dtype=params_dtype(double)).reshape(1).copy()
        lib.orc_recompute_constants(_ptr(p))
        return p

    @staticmethod
    def generate_cube(params, center, size, double=False, kernel_set=1):
        lib = _load(double, kernel_set)
        real = np.float64 if double else np.float32
        p = np.array(params, dtype=params_dtype(double)).reshape(1).copy()
        c = np.asarray(center, dtype=real)
        s = np.asarray(size, dtype=real)
        n = lib.orc_generate_cube(_ptr(p), _ptr(c), _ptr(s), None, 0)
        out = np.zeros((n, 4), dtype=real)
        lib.orc_generate_cube(_ptr(p), _ptr(c), _ptr(s), _ptr(out), n)
        return out

    @property
    def params(self):
        p = np.zeros(1, dtype=params_dtype(self.double))
        self.lib.orc_get_params(self.h, _ptr(p))
        return p

    def set_params(self, p):
        p = np.array(p, dtype=params_dtype(self.double)).reshape(1).copy()
        self.lib.orc_set_params(self.h, _ptr(p))

    def set_particles(self, pos4, vel4=None, pres=None):
        pos4 = np.ascontiguousarray(pos4, dtype=self.real).reshape(-1, 4)
        n = pos4.shape[0]
        vel4 = np.zeros((n, 4), self.real) if vel4 is None else np.ascontiguousarray(vel4, dtype=self.real)
        pres = None if pres is None else np.ascontiguousarray(pres, dtype=self.real)
        self.lib.orc_set_particles(self.h, _ptr(pos4), _ptr(vel4), _ptr(pres), n)

    def set_boundaries(self, bi4, vbi, update_grid=True):
        if bi4 is None or len(bi4) == 0:
            self.lib.orc_set_boundaries(self.h, None, None, 0, 0)
            return
        bi4 = np.ascontiguousarray(bi4, dtype=self.real).reshape(-1, 4)
        vbi = np.ascontiguousarray(vbi, dtype=self.real).reshape(-1)
        assert vbi.shape[0] == bi4.shape[0]
        self.lib.orc_set_boundaries(self.h, _ptr(bi4), _ptr(vbi), bi4.shape[0], int(update_grid))

    def step(self, nsteps=1, stop=0, max_iters=0):
        for _ in range(nsteps):
            self.lib.orc_step(self.h, self.solver, stop, max_iters)

    @property
    def n(self):
        return self.lib.orc_num_particles(self.h)

    @property
    def last_iters(self):
        return self.lib.orc_last_iters(self.h)

    def get(self, name):
        nbytes = self.lib.orc_get(self.h, name.encode(), None)
        if nbytes < 0:
            raise KeyError(name)
        dt = np.uint32 if name in _U32 else self.real
        a = np.zeros(nbytes // np.dtype(dt).itemsize, dtype=dt)
        self.lib.orc_get(self.h, name.encode(), _ptr(a))
        return a.reshape(-1, 4) if name in _VEC4 else a
