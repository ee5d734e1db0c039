"""ctypes binding of libnereus_hip.so (include/nereus_hip.h).

This is plumbing for tests/bench (Python side); the product host layer is the C++ mirror of the
reference's classes in nereus_amd/host/.  There is NO CPU fallback: if the shared library is missing
or no HIP device is usable, calls raise.
"""
import ctypes as C
import os

import numpy as np

from .params import params_dtype

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NEREUS_HIP_LIB") or os.path.join(_HERE, "libnereus_hip.so")  # (override: kernel A/B builds in tools/)

SESPH, IISPH = 0, 1
MONAGHAN, MULLER = 0, 1
FLAG_REFERENCE_ORDER = 1
FLAG_NO_FUSION = 4
FLAG_NO_SHARED_LISTS = 8
FLAG_FULL_SORT = 16
FLAG_FAST_ARITH = 32
FLAG_IISPH_SELF_BY_SLOT = 64
FLAG_NO_WALL_WORKGROUPS = 128
FLAG_STAGED_SCAN = 256
E_NOTREADY = -6
STAT_MOVERS, STAT_HIT_OVERFLOW, STAT_HIT_MEAN, STAT_HIT_MAX, STAT_UNSTAGED = 0, 1, 2, 3, 4

# NRS_STAGE_*
STAGE_HASH, STAGE_SORT, STAGE_REORDER, STAGE_DENSITY, STAGE_FORCES, STAGE_INTEGRATE = 1, 2, 3, 4, 5, 6
STAGE_I_DENSITY, STAGE_I_DISPLACEMENT, STAGE_I_ADVECTION, STAGE_I_SOLVE, STAGE_I_PFORCE, STAGE_I_INTEGRATE = (
    10, 11, 12, 13, 14, 15)
STAGE_NAMES = {1: "hash", 2: "sort", 3: "reorder", 4: "density", 5: "forces", 6: "integrate", 10: "i_density",
               11: "i_displacement", 12: "i_advection", 13: "i_solve", 14: "i_pforce", 15: "i_integrate"}

# NRS_ARR_*: name -> (id, kind) with kind in {"v4", "s", "u"}
ARRAYS = {
    "pos": (0, "v4"), "vel": (1, "v4"), "pressure": (2, "s"), "hash": (3, "u"), "index": (4, "u"),
    "cellStart": (5, "u"), "cellEnd": (6, "u"), "sortedPos": (7, "v4"), "sortedVel": (8, "v4"),
    "dens": (9, "s"), "pres": (10, "s"), "forces": (11, "v4"), "bhash": (12, "u"), "bindex": (13, "u"),
    "bCellStart": (14, "u"), "bCellEnd": (15, "u"), "bSorted": (16, "v4"),
    "densAdv": (20, "s"), "densCorr": (21, "s"), "P_l": (22, "s"), "aii": (23, "s"), "velAdv": (24, "v4"),
    "forcesAdv": (25, "v4"), "forcesP": (26, "v4"), "diiFluid": (27, "v4"), "diiBoundary": (28, "v4"),
    "sumDij": (29, "v4"),
}

# every symbol include/nereus_hip.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "nrs_last_error", "nrs_version", "nrs_device_count", "nrs_create", "nrs_destroy", "nrs_set_params",
    "nrs_get_params", "nrs_upload_particles", "nrs_set_num_particles", "nrs_num_particles", "nrs_set_boundaries",
    "nrs_step", "nrs_step_partial", "nrs_synchronize", "nrs_download", "nrs_get_array", "nrs_device_ptr",
    "nrs_last_iterations", "nrs_set_max_iterations", "nrs_set_profiling", "nrs_stage_ms", "nrs_max_density",
    "nrs_max_velocity", "nrs_slab_configure", "nrs_slab_pack", "nrs_slab_unpack", "nrs_num_owned",
    "nrs_slab_message_bytes", "nrs_slab_histogram", "nrs_resort_stats", "nrs_snapshot_begin", "nrs_snapshot_wait",
    "nrs_get_stat", "nrs_boundary_volumes", "nrs_eval_smoothing", "nrs_iisph_predict", "nrs_iisph_iterate", "nrs_iisph_finish",
    "nrs_slab_last_counts",
]


class NrsConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("device", C.c_int32), ("solver", C.c_int32), ("precision", C.c_int32),
        ("kernel_set", C.c_int32), ("surface_tension", C.c_int32), ("flags", C.c_uint32), ("reserved", C.c_uint32),
        ("capacity", C.c_uint64), ("stream", C.c_void_p),
    ]


class NereusError(RuntimeError):
    pass


_lib = None


def load_library(path=None):
    """Load libnereus_hip.so (once).  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise NereusError(
            "libnereus_hip.so not found at %s — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C nereus_amd/csrc`; there is no CPU fallback" % path)
    lib = C.CDLL(path)
    vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int
    lib.nrs_last_error.restype = C.c_char_p
    lib.nrs_version.restype = C.c_uint32
    lib.nrs_device_count.restype = i32
    lib.nrs_create.argtypes = [C.POINTER(NrsConfig), vp, C.POINTER(vp)]
    lib.nrs_destroy.argtypes = [vp]
    lib.nrs_set_params.argtypes = [vp, vp]
    lib.nrs_get_params.argtypes = [vp, vp]
    lib.nrs_upload_particles.argtypes = [vp, vp, vp, vp, u64, u64]
    lib.nrs_set_num_particles.argtypes = [vp, u64]
    lib.nrs_num_particles.argtypes = [vp]
    lib.nrs_num_particles.restype = u64
    lib.nrs_set_boundaries.argtypes = [vp, vp, vp, u64, i32]
    lib.nrs_step.argtypes = [vp, i32]
    lib.nrs_step_partial.argtypes = [vp, i32]
    lib.nrs_synchronize.argtypes = [vp]
    lib.nrs_download.argtypes = [vp, vp, vp, vp]
    lib.nrs_get_array.argtypes = [vp, i32, vp, u64, C.POINTER(u64)]
    lib.nrs_device_ptr.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(u64)]
    lib.nrs_last_iterations.argtypes = [vp, C.POINTER(C.c_uint32)]
    lib.nrs_set_max_iterations.argtypes = [vp, C.c_uint32]
    lib.nrs_set_profiling.argtypes = [vp, C.c_uint32]
    lib.nrs_stage_ms.argtypes = [vp, i32, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
    lib.nrs_max_density.argtypes = [vp, C.POINTER(C.c_double)]
    lib.nrs_max_velocity.argtypes = [vp, C.POINTER(C.c_double)]
    lib.nrs_slab_configure.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32]
    lib.nrs_slab_pack.argtypes = [vp, vp, vp, u64, C.POINTER(C.c_uint32)]
    lib.nrs_slab_unpack.argtypes = [vp, vp, vp, u64]
    lib.nrs_slab_last_counts.argtypes = [vp, C.POINTER(C.c_uint32)]
    lib.nrs_num_owned.argtypes = [vp]
    lib.nrs_num_owned.restype = u64
    lib.nrs_slab_message_bytes.argtypes = [u64, i32]
    lib.nrs_slab_message_bytes.restype = u64
    lib.nrs_slab_histogram.argtypes = [vp, C.c_int32, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.nrs_resort_stats.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    lib.nrs_get_stat.argtypes = [vp, i32, C.POINTER(C.c_double)]
    lib.nrs_iisph_predict.argtypes = [vp]
    lib.nrs_iisph_iterate.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(u64)]
    lib.nrs_iisph_finish.argtypes = [vp]
    lib.nrs_eval_smoothing.argtypes = [i32, i32, u64, vp, vp, C.c_double, C.c_double, C.c_double, vp]
    lib.nrs_boundary_volumes.argtypes = [i32, i32, vp, u64, C.c_double, vp]
    lib.nrs_snapshot_begin.argtypes = [vp, i32]
    lib.nrs_snapshot_wait.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(u64), C.POINTER(u64)]
    _lib = lib
    return lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def boundary_volumes(bi4, h, double=False, device=-1):
    """Akinci volumes of the boundary particles bi4 (n,4) on the device (nrs_boundary_volumes); returns (n,) SReal."""
    lib = load_library()
    real = np.float64 if double else np.float32
    bi4 = np.ascontiguousarray(bi4, dtype=real).reshape(-1, 4)
    out = np.empty(bi4.shape[0], dtype=real)
    rc = lib.nrs_boundary_volumes(int(device), 64 if double else 32, _ptr(bi4), bi4.shape[0], float(h), _ptr(out))
    if rc != 0:
        raise NereusError("libnereus_hip error %d: %s" % (rc, lib.nrs_last_error().decode()))
    return out


def eval_smoothing(which, r, s, h, c0, c1, double=False):
    """device smoothing kernel / vector helper number `which` on separations r (n,3) [and s]; returns (n,3) (nrs_eval_smoothing)"""
    lib = load_library()
    real = np.float64 if double else np.float32
    r = np.ascontiguousarray(r, dtype=real)
    s = None if s is None else np.ascontiguousarray(s, dtype=real)
    out = np.zeros_like(r)
    rc = lib.nrs_eval_smoothing(64 if double else 32, int(which), r.shape[0], _ptr(r), _ptr(s), float(h), float(c0), float(c1), _ptr(out))
    if rc != 0:
        raise NereusError("libnereus_hip error %d: %s" % (rc, lib.nrs_last_error().decode()))
    return out


class Solver:
    """Thin object wrapper over an nrs_ctx (device-resident SESPH / IISPH solver)."""

    def __init__(self, params, capacity, solver=SESPH, double=False, kernel_set=MULLER, surface_tension=True,
                 reference_order=False, device=-1, stream=None, flags=0):
        self.lib = load_library()
        self.double = bool(double)
        self.real = np.float64 if double else np.float32
        self.solver = solver
        p = np.array(params, dtype=params_dtype(double)).reshape(1).copy()
        cfg = NrsConfig()
        cfg.struct_size = C.sizeof(NrsConfig)
        cfg.device = device
        cfg.solver = solver
        cfg.precision = 64 if double else 32
        cfg.kernel_set = kernel_set
        cfg.surface_tension = int(bool(surface_tension))
        cfg.flags = (FLAG_REFERENCE_ORDER if reference_order else 0) | int(flags)
        cfg.capacity = int(capacity)
        cfg.stream = stream
        h = C.c_void_p()
        self.h = None
        self._chk(self.lib.nrs_create(C.byref(cfg), _ptr(p), C.byref(h)))
        self.h = h

    def _chk(self, rc):
        if rc != 0:
            raise NereusError("libnereus_hip error %d: %s" % (rc, self.lib.nrs_last_error().decode()))

    def close(self):
        if self.h:
            self.lib.nrs_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def params(self):
        p = np.zeros(1, dtype=params_dtype(self.double))
        self._chk(self.lib.nrs_get_params(self.h, _ptr(p)))
        return p

    def set_params(self, params):
        p = np.array(params, dtype=params_dtype(self.double)).reshape(1).copy()
        self._chk(self.lib.nrs_set_params(self.h, _ptr(p)))

    def set_particles(self, pos4, vel4=None, pres=None, first=0):
        pos4 = np.ascontiguousarray(pos4, dtype=self.real).reshape(-1, 4)
        n = pos4.shape[0]
        vel4 = None if vel4 is None else np.ascontiguousarray(vel4, dtype=self.real)
        pres = None if pres is None else np.ascontiguousarray(pres, dtype=self.real)
        if first == 0:
            self._chk(self.lib.nrs_set_num_particles(self.h, 0))
        self._chk(self.lib.nrs_upload_particles(self.h, _ptr(pos4), _ptr(vel4), _ptr(pres), first, n))

    def set_boundaries(self, bi4, vbi, update_grid=True):
        if bi4 is None or len(bi4) == 0:
            self._chk(self.lib.nrs_set_boundaries(self.h, None, None, 0, 0))
            return
        bi4 = np.ascontiguousarray(bi4, dtype=self.real).reshape(-1, 4)
        vbi = np.ascontiguousarray(vbi, dtype=self.real).reshape(-1)
        assert vbi.shape[0] == bi4.shape[0]
        self._chk(self.lib.nrs_set_boundaries(self.h, _ptr(bi4), _ptr(vbi), bi4.shape[0], int(update_grid)))

    @property
    def n(self):
        return int(self.lib.nrs_num_particles(self.h))

    def step(self, nsteps=1):
        self._chk(self.lib.nrs_step(self.h, int(nsteps)))

    def step_partial(self, stage):
        self._chk(self.lib.nrs_step_partial(self.h, int(stage)))

    def synchronize(self):
        self._chk(self.lib.nrs_synchronize(self.h))

    def download(self, pressure=False):
        n = self.n
        pos = np.empty((n, 4), self.real)
        vel = np.empty((n, 4), self.real)
        pres = np.empty(n, self.real) if pressure else None
        self._chk(self.lib.nrs_download(self.h, _ptr(pos), _ptr(vel), _ptr(pres)))
        return (pos, vel, pres) if pressure else (pos, vel)

    def get(self, name):
        aid, kind = ARRAYS[name]
        nbytes = C.c_uint64(0)
        self._chk(self.lib.nrs_get_array(self.h, aid, None, 0, C.byref(nbytes)))
        dt = np.uint32 if kind == "u" else self.real
        a = np.empty(nbytes.value // np.dtype(dt).itemsize, dtype=dt)
        if nbytes.value:
            self._chk(self.lib.nrs_get_array(self.h, aid, _ptr(a), nbytes.value, None))
        return a.reshape(-1, 4) if kind == "v4" else a

    def device_ptr(self, name):
        aid, _ = ARRAYS[name]
        p, b = C.c_void_p(), C.c_uint64()
        self._chk(self.lib.nrs_device_ptr(self.h, aid, C.byref(p), C.byref(b)))
        return p.value, b.value

    @property
    def last_iterations(self):
        it = C.c_uint32(0)
        self._chk(self.lib.nrs_last_iterations(self.h, C.byref(it)))
        return it.value

    def set_max_iterations(self, m):
        self._chk(self.lib.nrs_set_max_iterations(self.h, int(m)))

    def set_profiling(self, stages=True):
        """stages: True = all, False = off, or an iterable of stage ids."""
        if stages is True:
            mask = 0xFFFFFFFF
        elif not stages:
            mask = 0
        else:
            mask = 0
            for s in stages:
                mask |= 1 << int(s)
        self._chk(self.lib.nrs_set_profiling(self.h, mask))

    def stage_ms(self):
        """{stage name: (ms summed over the steps since the last set_profiling call, launches)}; synchronizes"""
        out = {}
        for sid, name in STAGE_NAMES.items():
            ms, cnt = C.c_float(0), C.c_uint32(0)
            self._chk(self.lib.nrs_stage_ms(self.h, sid, C.byref(ms), C.byref(cnt)))
            if cnt.value:
                out[name] = (ms.value, cnt.value)
        return out

    # ---- slab decomposition ------------------------------------------------------------------------
    def slab_configure(self, cell_lo, cell_hi, halo_cells=2):
        self._chk(self.lib.nrs_slab_configure(self.h, int(cell_lo), int(cell_hi), int(halo_cells)))

    def slab_pack(self, send_left_ptr, send_right_ptr, capacity, want_counts=True):
        """want_counts=False: do not wait for the device (the counts are read back inside slab_unpack; slab_last_counts())"""
        if not want_counts:
            self._chk(self.lib.nrs_slab_pack(self.h, send_left_ptr, send_right_ptr, int(capacity), None))
            return None
        counts = (C.c_uint32 * 6)()
        self._chk(self.lib.nrs_slab_pack(self.h, send_left_ptr, send_right_ptr, int(capacity), counts))
        return list(counts)

    def slab_last_counts(self):
        counts = (C.c_uint32 * 6)()
        self._chk(self.lib.nrs_slab_last_counts(self.h, counts))
        return list(counts)

    def slab_unpack(self, recv_left_ptr, recv_right_ptr, capacity):
        self._chk(self.lib.nrs_slab_unpack(self.h, recv_left_ptr, recv_right_ptr, int(capacity)))

    def slab_histogram(self, first_cell, ncells):
        out = (C.c_uint32 * int(ncells))()
        self._chk(self.lib.nrs_slab_histogram(self.h, int(first_cell), int(ncells), out))
        return np.frombuffer(out, dtype=np.uint32).copy()

    def snapshot_begin(self, with_vel=False):
        """start an asynchronous copy of the current positions (and velocities) to pinned host memory"""
        self._chk(self.lib.nrs_snapshot_begin(self.h, 1 if with_vel else 0))

    def snapshot_wait(self, block=True):
        """oldest pending snapshot as (pos, vel or None, step) — numpy views of the library's pinned buffers, valid until
        two more snapshot_begin calls — or None if block is False and the transfer is still running"""
        pp, pv, n, step = C.c_void_p(), C.c_void_p(), C.c_uint64(), C.c_uint64()
        rc = self.lib.nrs_snapshot_wait(self.h, 1 if block else 0, C.byref(pp), C.byref(pv), C.byref(n), C.byref(step))
        if rc == E_NOTREADY:
            return None
        self._chk(rc)
        ct = C.c_double if self.real == np.float64 else C.c_float

        def view(ptr):
            if not ptr.value or not n.value:
                return np.empty((0, 4), self.real) if ptr.value or not n.value else None
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(int(n.value), 4))

        return view(pp), (view(pv) if pv.value else None), int(step.value)

    def resort_stats(self):
        """(steps that used the coherent re-sort path, how many of them fell back to the full radix sort)"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._chk(self.lib.nrs_resort_stats(self.h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def iisph_predict(self):
        self._chk(self.lib.nrs_iisph_predict(self.h))

    def iisph_iterate(self):
        """one solver iteration; returns (sum of corrected densities over the owned particles, their number)"""
        sm, cnt = C.c_double(0), C.c_uint64(0)
        self._chk(self.lib.nrs_iisph_iterate(self.h, C.byref(sm), C.byref(cnt)))
        return sm.value, int(cnt.value)

    def iisph_finish(self):
        self._chk(self.lib.nrs_iisph_finish(self.h))

    def get_stat(self, which):
        v = C.c_double()
        self._chk(self.lib.nrs_get_stat(self.h, int(which), C.byref(v)))
        return v.value

    @property
    def n_owned(self):
        return int(self.lib.nrs_num_owned(self.h))

    def message_bytes(self, capacity):
        return int(self.lib.nrs_slab_message_bytes(int(capacity), 64 if self.double else 32))

    def max_density(self):
        v = C.c_double()
        self._chk(self.lib.nrs_max_density(self.h, C.byref(v)))
        return v.value

    def max_velocity(self):
        v = C.c_double()
        self._chk(self.lib.nrs_max_velocity(self.h, C.byref(v)))
        return v.value
