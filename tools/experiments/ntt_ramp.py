# per-step duration of the bench's NTT step over a long run: how long the device takes to reach its steady clock
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import os  # noqa: E402

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before HIP initialises (halo2-gpu-specific_amd/__init__.py says why)
import torch
import halo2_gpu_specific_amd as h2
from halo2_gpu_specific_amd._lib import check
import numpy as np
R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
ROOT_OF_UNITY = 0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C
def fr_limbs(v):
    v = v * (1 << 256) % R_MOD
    return np.array([(v >> (64 * i)) & ((1 << 64) - 1) for i in range(4)], dtype=np.uint64)
L = h2.lib()
dev = torch.device("cuda", 0)
ts = torch.cuda.Stream()
torch.cuda.set_stream(ts)
stream = ctypes.c_void_p(ts.cuda_stream)
vp = lambda x: x.ctypes.data_as(ctypes.c_void_p)
log_n = 24; n = 1 << log_n
omega = pow(ROOT_OF_UNITY, 1 << (28 - log_n), R_MOD)
w_f, w_i, n_inv = fr_limbs(omega), fr_limbs(pow(omega, -1, R_MOD)), fr_limbs(pow(n, -1, R_MOD))
a = torch.randint(-(2**63), 2**63 - 1, (n, 4), dtype=torch.int64, device=dev)
a[:, 3] &= 0x1FFFFFFFFFFFFFFF
tmp = torch.empty_like(a)
def step():
    check(L.h2_dev_ntt(a.data_ptr(), tmp.data_ptr(), vp(w_f), log_n, stream), "ntt")
    check(L.h2_dev_intt(a.data_ptr(), tmp.data_ptr(), vp(w_i), vp(n_inv), log_n, stream), "intt")
for trial in range(2):
    torch.cuda.synchronize(); time.sleep(1.0)
    N = 120
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    ev[0].record()
    for i in range(N):
        step(); ev[i + 1].record()
    torch.cuda.synchronize()
    d = [ev[i].elapsed_time(ev[i + 1]) for i in range(N)]
    print("trial", trial, "per-step ms:", " ".join("%.2f" % x for x in d[:30]), "...", " ".join("%.2f" % x for x in d[-10:]))
    print("  mean first 10 %.3f  10-25 %.3f  25-60 %.3f  60-120 %.3f" % (sum(d[:10]) / 10, sum(d[10:25]) / 15, sum(d[25:60]) / 35, sum(d[60:]) / 60))
