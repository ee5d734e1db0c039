"""A slice of the randomised parity soak (tools/fuzz_parity.py): production kernels == reference-order kernels bit for bit on random
particle clouds with random grid geometry, wall sheets, precision, kernel set and solver."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_random_scenes_production_equals_reference_order(hip_lib):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from fuzz_parity import one

    failures, diverged = [], 0
    for seed in range(9000, 9150):
        r = one(seed)
        if r == "diverged":
            diverged += 1
        elif r:
            failures.append(r)
    assert not failures, failures[:3]
    assert diverged < 40


@pytest.mark.gpu
def test_random_slab_runs_equal_single_domain(hip_lib):
    """A slice of tools/fuzz_slab.py: 2-4 ranks as contexts of this process, random cuts / re-cuts / migrations."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from fuzz_slab import STATS, one

    failures = [r for r in (one(seed) for seed in range(7000, 7012)) if r and r != "skip"]
    assert not failures, failures[:3]
    assert STATS["migrants"] > 0
