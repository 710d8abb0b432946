"""h2_intt on 32 MiB host vectors (k = 20) in ordinary memory: the same vector again and again, a new numpy / torch vector per call,
one thread and four -- what a 64-column witness does to the host-slice entry points.  python tools/experiments/pageable_intt_probe.py"""
import concurrent.futures
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# (as bench.py: OpenMP teams -- torch's CPU operators here -- sleep instead of spinning; a spinning team of 256 on a box whose cgroup
# grants 16 CPUs gets the whole process throttled, and every number below with it)
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
os.environ.setdefault("GOMP_SPINCOUNT", "0")
import numpy as np  # noqa: E402
import torch  # noqa: E402

import halo2_gpu_specific_amd as h2  # noqa: E402
from halo2_gpu_specific_amd import prover  # noqa: E402

L = h2.lib()
k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << k
dom = prover.Domain(k, 5)
wi, dv = prover._fr(dom.omega_inv), prover._fr(dom.ifft_divisor)
rng = np.random.default_rng(1)
src = rng.integers(0, 2**61, size=(n, 4), dtype=np.uint64)


def intt(addr):
    t0 = time.perf_counter()
    assert L.h2_intt(addr, wi, dv, k) == 0
    return (time.perf_counter() - t0) * 1e3


a = src.copy()
intt(a.ctypes.data)
print("same numpy vector, 5 calls:      ", ["%.2f" % intt(a.ctypes.data) for _ in range(5)])
news = [src.copy() for _ in range(5)]
print("a new numpy copy per call:       ", ["%.2f" % intt(x.ctypes.data) for x in news])
tn = [torch.from_numpy(src.view(np.int64)).clone() for _ in range(5)]
print("a new torch clone per call:      ", ["%.2f" % intt(x.data_ptr()) for x in tn])
print("... those again:                 ", ["%.2f" % intt(x.data_ptr()) for x in tn])
tp = [torch.from_numpy(src.view(np.int64)).clone().pin_memory() for _ in range(5)]
print("page-locked clones:              ", ["%.2f" % intt(x.data_ptr()) for x in tp])
pool = concurrent.futures.ThreadPoolExecutor(max_workers=4)
for what, make in (("new torch clones", lambda: torch.from_numpy(src.view(np.int64)).clone()),
                   ("page-locked clones", lambda: torch.from_numpy(src.view(np.int64)).clone().pin_memory())):
    vs = [make() for _ in range(16)]
    t0 = time.perf_counter()
    each = list(pool.map(lambda x: intt(x.data_ptr()), vs))
    print("4 threads, 16 %s: wall %.2f ms, per call mean %.2f max %.2f" % (what, (time.perf_counter() - t0) * 1e3, sum(each) / 16, max(each)))
    t0 = time.perf_counter()
    each = list(pool.map(lambda x: intt(x.data_ptr()), vs))
    print("   ... again:                    wall %.2f ms, per call mean %.2f max %.2f" % ((time.perf_counter() - t0) * 1e3, sum(each) / 16, max(each)))
