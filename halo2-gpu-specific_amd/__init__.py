"""halo2-gpu-specific_amd -- MI355X-native halo2 prover hot path (MSM / NTT / coset FFT / batched
polynomial arithmetic) behind the C ABI of include/halo2_hip.h.

This Python layer is a thin ctypes binding used by tests and bench.py; the product is
libhalo2_hip.so (csrc/).  There is no CPU fallback: importing `lib()` without the built
extension raises.
"""
from ._lib import H2Error, build, lib, lib_path  # noqa: F401
from . import arithmetic  # noqa: F401
