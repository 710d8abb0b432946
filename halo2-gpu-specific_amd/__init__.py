"""halo2-gpu-specific_amd -- MI355X-native halo2 prover hot path (MSM / NTT / coset FFT / batched
polynomial arithmetic) behind the C ABI of include/halo2_hip.h.

This Python layer is a thin ctypes binding used by tests and bench.py; the product is
libhalo2_hip.so (csrc/).  There is no CPU fallback: importing `lib()` without the built
extension raises.
"""
import os as _os

# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share one serialise.
# The prover runs a compute stream, a copy stream, a side stream and the library's own second MSM stream next to torch's:
# with four queues the two-stream MSM pipeline of a wide witness waited for the copy stream's backlog (k = 22, 64 columns:
# 515 -> 452 ms with eight queues, 412 -> 390 from a compact witness, when measured; since the 16-bit columns fuse at that
# size too: 450 either way, 382 -> 373 compact; nothing else moves -- profiles/r5_hw_queues_ab.txt).  Read when the HIP
# runtime initialises, i.e. before the first HIP call of the process (importing torch does not make one); a caller's own
# setting wins.  libhalo2_hip.so sets the same default when it is loaded (csrc/context.hip) for hosts without Python.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from ._lib import H2Error, build, lib, lib_path  # noqa: F401,E402
from . import arithmetic  # noqa: F401,E402
