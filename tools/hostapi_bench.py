"""Host-buffer entry points (what a minimal Rust drop-in calls): PCIe-inclusive timings.  usage: python tools/hostapi_bench.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before HIP initialises (halo2-gpu-specific_amd/__init__.py says why)
import torch  # noqa: E402

torch.cuda.init()
import numpy as np  # noqa: E402

import halo2_gpu_specific_amd as h2  # noqa: E402
from halo2_gpu_specific_amd import arithmetic as ar  # noqa: E402

R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
ROOT = 0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C


def mont(v):
    m = (v << 256) % R
    return np.array([(m >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)


def t(name, f, reps=3):
    f()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    dt = (time.perf_counter() - t0) / reps
    print("%-46s %8.2f ms" % (name, dt * 1e3))
    return dt


L = h2.lib()
rng = np.random.default_rng(1)
for log_n in (20, 22, 24):
    n = 1 << log_n
    a = rng.integers(0, 2**62, size=(n, 4), dtype=np.uint64)
    w = mont(pow(ROOT, 1 << (28 - log_n), R))
    dt = t("h2_ntt 2^%d (host buffer in/out, %d MiB each way)" % (log_n, n * 32 >> 20), lambda: ar.best_fft(a, w, log_n))
    print("    -> %.1f GB/s of PCIe traffic" % (2 * n * 32 / dt / 1e9))
n = 1 << 20
d_pts = torch.empty((n, 8), dtype=torch.int64, device="cuda")
L.h2_dev_random_points(7, n, d_pts.data_ptr(), None)
L.h2_synchronize()
pts = d_pts.cpu().numpy().view(np.uint64)
sc = rng.integers(0, 2**62, size=(n, 4), dtype=np.uint64)
t("h2_msm 2^20 (scalars + bases uploaded)", lambda: ar.gpu_multiexp_single_gpu_with_bound(sc, pts, 254))
L.h2_bases_register(pts.ctypes.data, n)
t("h2_msm 2^20 (bases registered / resident)", lambda: ar.gpu_multiexp_single_gpu_with_bound(sc, pts, 254))


# ---- concurrent callers (the reference's entry points are invoked from rayon workers): T threads, each its own vectors; with
# two host-API slots per device (H2_HOST_SLOTS, default 2) one call's transfers overlap another's kernels
import threading  # noqa: E402


def concurrent(name, make_call, threads, reps=6):
    calls = [make_call(i) for i in range(threads)]
    for c in calls:
        c()
    t0 = time.perf_counter()
    ts = [threading.Thread(target=lambda c=c: [c() for _ in range(reps)]) for c in calls]
    for th in ts:
        th.start()
    for th in ts:
        th.join()
    dt = time.perf_counter() - t0
    print("%-58s %8.2f ms per call  (%d threads)" % (name, dt / (threads * reps) * 1e3, threads))


log_n = 22
w22 = mont(pow(ROOT, 1 << (28 - log_n), R))
arrs = [rng.integers(0, 2**62, size=(1 << log_n, 4), dtype=np.uint64) for _ in range(4)]
scs = [rng.integers(0, 2**62, size=(n, 4), dtype=np.uint64) for _ in range(4)]
print("H2_HOST_SLOTS =", os.environ.get("H2_HOST_SLOTS", "2 (default)"))
for T in (1, 2, 4):
    concurrent("h2_ntt 2^22, concurrent callers", lambda i: (lambda: ar.best_fft(arrs[i], w22, log_n)), T)
for T in (1, 2, 4):
    concurrent("h2_msm 2^20 (registered bases), concurrent callers", lambda i: (lambda: ar.gpu_multiexp_single_gpu_with_bound(scs[i], pts, 254)), T)
