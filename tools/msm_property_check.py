"""Size-independent MSM properties at sizes the oracle does not reach in seconds (2^22, 2^24): linearity
MSM(a + b) = MSM(a) + MSM(b) over repeated bases, and a 28-bit column split into its 16-bit and 12-bit halves
(MSM(lo) + 2^16 MSM(hi) = MSM(full): bounded / narrow-column shapes with row ranges).   usage: python tools/msm_property_check.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before HIP initialises (halo2-gpu-specific_amd/__init__.py says why)
import torch
torch.cuda.init()
import numpy as np
from h2util import Oracle
from halo2_gpu_specific_amd import arithmetic as ar
oracle = Oracle.get()
def aff(j):
    return tuple(oracle.to_affine(j).tolist())
for log_n in (22, 24):
    n = 1 << log_n
    t0 = time.time()
    pts = oracle.random_g1(5, 1 << 16)
    pts = np.tile(pts, (n >> 16, 1))          # repeated points: also exercises P + P inside buckets
    a, b = oracle.random_fr(6, n), oracle.random_fr(7, n)
    ab = oracle.eval_op(ar.OP_SUM, a, b, 0, 0, None)
    ra, rb, rab = ar.best_multiexp(a, pts), ar.best_multiexp(b, pts), ar.best_multiexp(ab, pts)
    s = np.zeros(12, dtype=np.uint64)
    oracle.lib.oracle_g1_add(ra.ctypes.data, rb.ctypes.data, s.ctypes.data)
    print(log_n, "linearity", aff(s) == aff(rab), "%.1f s" % (time.time() - t0))
    # bounded + narrow columns at this size: MSM(16-bit) + 2^16 * MSM(16-bit hi) == MSM(32-bit)
    lo = np.zeros((n, 4), dtype=np.uint64); hi = np.zeros((n, 4), dtype=np.uint64); full = np.zeros((n, 4), dtype=np.uint64)
    rows = np.arange(n, dtype=np.uint64)
    lo[:, 0] = (rows * 2654435761) % 65536
    hi[:, 0] = (rows * 40503 + 7) % 4096
    full[:, 0] = lo[:, 0] + (hi[:, 0] << np.uint64(16))
    import halo2_gpu_specific_amd as h2
    L = h2.lib()
    def dev_mont(a):
        t = torch.from_numpy(a.view(np.int64)).cuda()
        assert L.h2_dev_batch_mont(t.data_ptr(), n, None) == 0
        return t.cpu().numpy().view(np.uint64)
    lo_m, hi_m, full_m = dev_mont(lo), dev_mont(hi), dev_mont(full)
    r_lo = ar.gpu_multiexp_single_gpu_with_bound(lo_m, pts, 16)
    r_hi = ar.gpu_multiexp_single_gpu_with_bound(hi_m, pts, 12)
    r_full = ar.gpu_multiexp_single_gpu_with_bound(full_m, pts, 28)
    # [2^16] r_hi by 16 doublings through the oracle
    acc = r_hi.copy()
    for _ in range(16):
        t = np.zeros(12, dtype=np.uint64)
        oracle.lib.oracle_g1_add(acc.ctypes.data, acc.ctypes.data, t.ctypes.data)
        acc = t
    t = np.zeros(12, dtype=np.uint64)
    oracle.lib.oracle_g1_add(acc.ctypes.data, r_lo.ctypes.data, t.ctypes.data)
    print(log_n, "narrow-column split", aff(t) == aff(r_full))
