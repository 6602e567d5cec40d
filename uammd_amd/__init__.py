"""uammd_amd — MI355X-native implementation of UAMMD's data-parallel hot path.

The product is libuammd_hip.so (hand-written HIP for gfx950, `uammd_amd/csrc`) behind the C ABI of
`include/uammd_hip.h`; UAMMD-shaped C++14 headers live in `include/uammd/`.  This Python package is
the thin host-side harness used by the tests and `bench.py`: it mirrors the reference's class names
(ParticleData, Box, CellList, Potential.LJ, PairForces, VerletNVT.GronbechJensen, BD.EulerMaruyama,
BDHI.FCM ...) and forwards every call through ctypes.  PyTorch is used only for device memory,
streams and torch.distributed.  There is no CPU fallback anywhere in this package.
"""
from ._lib import UammdHipError, load  # noqa: F401
from .md import (BD, Box, CellList, Integrator, Interactor, PairForces, ParticleData, ParticleGroup, Potential,  # noqa: F401
                 VerletList, VerletNVT, current_stream)
from .electrostatics import Poisson  # noqa: F401
from .hydro import Hydro  # noqa: F401
from .checkpoint import restoreParticleData, saveParticleData  # noqa: F401
from .bdhi import BDHI, IBM, FCMKernels, Kernels, nextFFTWiseSize3D  # noqa: F401

__all__ = ["UammdHipError", "load", "Box", "ParticleData", "ParticleGroup", "CellList", "VerletList", "Potential", "PairForces", "Interactor",
           "Integrator", "VerletNVT", "BD", "BDHI", "IBM", "Poisson", "Kernels", "FCMKernels", "current_stream"]
