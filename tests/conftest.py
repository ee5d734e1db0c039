import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """libnereus_hip.so, required: GPU tests must exercise the HIP path, never a fallback."""
    from nereus_amd import capi

    lib = capi.load_library()
    if lib.nrs_device_count() <= 0:
        pytest.fail("no HIP device visible: -m gpu tests need a GPU (there is no CPU fallback)")
    return lib
