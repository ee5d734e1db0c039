"""Shared pieces of the oracle <-> reference pin (tests/test_oracle_ref_pin.py, tests/golden/make_ref_pin.py).

oracle/_ref/libnereus_refkernels_d{d}k{k}.so is the REFERENCE's own common/kernels_impl.cuh + helper_math.h compiled by
path, unmodified, with g++ (oracle/Makefile target `ref`, oracle/ref_kernels_driver.cpp).  The oracle exports the same
batch evaluator (orc_eval); both are fed identical inputs and compared bit for bit.
"""
import ctypes as C
import os

import numpy as np

from tests.oracle_lib import ORACLE_DIR, Oracle, _load, _ptr

REF_DIR = os.path.join(ORACLE_DIR, "_ref")
# (which, name, uses second vector)
FUNCTIONS = [
    (0, "Wdefault", False), (1, "Wdefault_grad", False), (2, "Wpressure_grad", False), (3, "Wviscosity_grad", False),
    (4, "Wmonaghan", False), (5, "Wmonaghan_grad", False), (6, "Cakinci", False), (7, "Aboundary", False),
    (8, "dot", True), (9, "length", False), (10, "vec*scalar", False), (11, "scalar*vec", False),
    (12, "vec/scalar", False), (13, "make_SVec3(SVec4)", False), (14, "vec+vec", True), (15, "vec-vec", True),
]
VARIANTS = [(0, 1), (0, 0), (1, 1), (1, 0)]  # (DOUBLE_PRECISION, KERNEL_SET)


def ref_lib_path(double, kset):
    return os.path.join(REF_DIR, "libnereus_refkernels_d%dk%d.so" % (int(double), int(kset)))


def _sig(fn, real):
    fn.restype = C.c_int
    fn.argtypes = [C.c_int, C.c_uint, C.c_void_p, C.c_void_p, real, real, real, C.c_void_p]


def load_ref(double, kset):
    lib = C.CDLL(ref_lib_path(double, kset))
    assert lib.ref_sizeof_real() == (8 if double else 4)
    _sig(lib.ref_eval, C.c_double if double else C.c_float)
    return lib.ref_eval


def load_orc(double, kset):
    lib = _load(double, kset)
    _sig(lib.orc_eval, C.c_double if double else C.c_float)
    return lib.orc_eval


def constants(which, h, double, solver=0):
    """(c0, c1) the reference passes to function `which`: the precomputed kernel constants of SphSimParams for this h."""
    p = Oracle.default_params(solver, double, 1)
    p["interactionRadius"][0] = h
    p = Oracle.recompute_constants(p, double, 1)
    g = lambda k: float(p[k][0])
    return {0: (g("kpoly"), 0.0), 1: (g("kpoly_grad"), 0.0), 2: (g("kpress_grad"), 0.0),
            3: (g("kvisc_grad"), g("kvisc_denum")), 6: (g("ksurf1"), g("ksurf2")), 7: (g("bpol"), 0.0),
            10: (0.7310585786300049, 0.0), 11: (1e-3, 0.0), 12: (0.0457, 0.0)}.get(which, (0.0, 0.0))


def inputs(n_random, h, double, seed):
    """r (and a second vector s): random separations up to 2.5 h plus the edge cases of the kernels' branches."""
    real = np.float64 if double else np.float32
    rng = np.random.default_rng(seed)
    r = rng.uniform(-1.0, 1.0, (n_random, 3))
    r *= (rng.uniform(0.0, 2.5, (n_random, 1)) * h / np.maximum(np.linalg.norm(r, axis=1, keepdims=True), 1e-30))
    hh = real(h)
    f32 = np.float32
    edge = [
        (0, 0, 0), (hh, 0, 0), (0, -hh, 0), (0, 0, hh),
        (np.nextafter(f32(h), f32(0)), 0, 0), (np.nextafter(f32(h), f32(1)), 0, 0),
        (np.nextafter(hh, real(0)), 0, 0), (np.nextafter(hh, real(1)), 0, 0),
        (2 * hh, 0, 0), (np.nextafter(f32(2 * h), f32(0)), 0, 0), (np.nextafter(f32(2 * h), f32(1)), 0, 0),
        (hh / 2, 0, 0), (np.nextafter(f32(h / 2), f32(0)), 0, 0), (np.nextafter(f32(h / 2), f32(1)), 0, 0),
        (0.04, 0, 0), (0.02, 0, 0), (1e-20, 0, 0), (1e-30, 1e-30, 0), (3.0, -4.0, 12.0), (1e18, 0, 0),
        (hh / np.sqrt(real(3)),) * 3, (-hh / np.sqrt(real(2)), hh / np.sqrt(real(2)), 0),
    ]
    # points scattered within a few ulp of |r| = h, h/2 and 2h along random directions
    d = rng.normal(size=(600, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    shell = np.concatenate([d[:200] * h, d[200:400] * (h / 2), d[400:] * (2 * h)]) * (1 + rng.integers(-4, 5, (600, 1)) * 1e-7)
    r = np.concatenate([np.array(edge, dtype=np.float64), shell, r]).astype(real)
    s = rng.uniform(-0.1, 0.1, r.shape).astype(real)
    return np.ascontiguousarray(r), np.ascontiguousarray(s)


def evaluate(fn, which, r, s, h, c0, c1, use_s):
    out = np.zeros_like(r)
    rc = fn(which, r.shape[0], _ptr(r), _ptr(s) if use_s else None, h, c0, c1, _ptr(out))
    assert rc == 0
    return out


def bits(a):
    return a.view(np.uint64 if a.dtype == np.float64 else np.uint32)
