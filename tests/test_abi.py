"""The C-ABI shared library loads and exports every symbol include/nereus_hip.h declares (no GPU needed)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from nereus_amd import capi
from nereus_amd.params import params_dtype

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "nereus_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nrs_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(capi.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = capi.load_library()
    for name in _declared_symbols():
        assert hasattr(lib, name), name


def test_struct_sizes_match_header():
    assert params_dtype(False).itemsize == 132 and params_dtype(True).itemsize == 240
    assert C.sizeof(capi.NrsConfig) == 48


def test_no_silent_cpu_fallback():
    """Without a GPU nrs_create must fail loudly (NRS_E_NODEVICE), never run on the CPU."""
    lib = capi.load_library()
    if lib.nrs_device_count() > 0:
        pytest.skip("a GPU is present; the failure path is exercised on the CPU box")
    p = np.zeros(1, dtype=params_dtype(False))
    with pytest.raises(capi.NereusError) as e:
        capi.Solver(p, capacity=16)
    assert "-5" in str(e.value) or "no HIP device" in str(e.value)


def test_makefiles_can_build_every_library_from_nothing():
    """`make -n -B` (dry run, everything out of date) must print a link line for each library the package loads: a rule that gets lost
    shows up here and not only on a fresh checkout (the .so files are not in the history)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    want = {"nereus_amd/csrc": ["../libnereus_hip.so", "../libnereus_refshim.so", "../libnereus_refshim_f64.so",
                                "../libnereus_refshim_monaghan.so", "../libnereus_refshim_f64_monaghan.so"],
            "nereus_amd/host": ["libnereus_host.so"], "oracle": ["libnereus_oracle"]}
    for d, outs in want.items():
        r = subprocess.run(["make", "-n", "-B", "-C", os.path.join(root, d)], capture_output=True, text=True)
        assert r.returncode == 0, (d, r.stderr[-400:])
        for o in outs:
            assert any(("-o " + o) in line or ("-o ../" + o) in line or (o in line and " -o " in line) for line in r.stdout.splitlines()), (d, o)
