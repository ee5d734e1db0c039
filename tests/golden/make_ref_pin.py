#!/usr/bin/env python3
"""Generates tests/golden/ref_kernels_pin.npz: inputs and the REFERENCE's outputs for every smoothing kernel and vector
helper of /root/reference/common/kernels_impl.cuh + cuda_helpers/helper_math.h.

The outputs come from oracle/_ref/libnereus_refkernels_*.so, i.e. from the reference's own source files compiled
unmodified with g++ in this container (`make -C oracle ref`; needs /root/reference).  The fixture is data (inputs and
expected outputs), so the pin also holds where /root/reference and oracle/_ref are absent.

  python tests/golden/make_ref_pin.py
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from tests import ref_pin  # noqa: E402

N_RANDOM = 400
RADII = (0.0457, 0.0537)  # SPH::SPH() and IISPH::IISPH() defaults (sph.cpp:60, iisph.cpp:45)


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all", "ref"])
    out = {}
    for double, kset in ((0, 1), (1, 1)):  # KERNEL_SET does not occur in either reference file: one per precision
        ref = ref_pin.load_ref(double, kset)
        for hi, h in enumerate(RADII):
            r, s = ref_pin.inputs(N_RANDOM, h, double, seed=1234 + hi)
            tag = "d%d_h%d" % (double, hi)
            out[tag + "_r"] = r
            out[tag + "_s"] = s
            for which, name, use_s in ref_pin.FUNCTIONS:
                c0, c1 = ref_pin.constants(which, h, double)
                out["%s_f%d" % (tag, which)] = ref_pin.evaluate(ref, which, r, s, h, c0, c1, use_s)
    path = os.path.join(ROOT, "tests", "golden", "ref_kernels_pin.npz")
    np.savez_compressed(path, radii=np.array(RADII), **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
