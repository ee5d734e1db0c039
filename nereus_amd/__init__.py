"""nereus_amd — MI355X-native SPH fluid step (SESPH / IISPH) behind the Nereus::SPH host API.

Layout:
  csrc/   hand-written gfx950 HIP kernels + the C ABI (libnereus_hip.so, include/nereus_hip.h)
  host/   C++ mirror of the reference's host classes (Nereus::SPH, Nereus::IISPH) over that ABI
  capi.py ctypes plumbing used by tests/ and bench.py
  scene.py deterministic synthetic dam-break generator
  slab.py  multi-GPU slab decomposition driver over torch.distributed (RCCL)
"""
from . import params  # noqa: F401

__all__ = ["params"]
