"""round trips of the largest transforms (2^26 .. 2^28): h2_dev_intt against h2_dev_ntt(omega^-1) * n^-1, and inverse o
forward = identity.   usage: python tools/experiments/ntt_big_roundtrip.py [log_n ...]"""
import os
import sys

ROOT_DIR = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT_DIR)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before HIP initialises (halo2-gpu-specific_amd/__init__.py says why)
import torch  # noqa: E402

from halo2_gpu_specific_amd import lib  # noqa: E402
from halo2_gpu_specific_amd._lib import check  # noqa: E402
from halo2_gpu_specific_amd.prover import _fr, R_MOD  # noqa: E402

ROOT = 0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C
L = lib()
dev = torch.device("cuda", 0)
import ctypes
TS = torch.cuda.Stream()                       # one stream for torch's ops AND the library's kernels: one order for all work
torch.cuda.set_stream(TS)                      # (the null stream would mean 'the library's own stream' to the C ABI)
ST = ctypes.c_void_p(TS.cuda_stream)
for log_n in [int(v) for v in sys.argv[1:]] or [26, 27, 28]:
    n = 1 << log_n
    omega = pow(ROOT, 1 << (28 - log_n), R_MOD)
    x = torch.empty((n, 4), dtype=torch.int64, device=dev)
    check(L.h2_dev_random_fr(b"\x07" * 32, n, x.data_ptr(), ST), "rnd")
    a, tmp = x.clone(), torch.empty_like(x)
    check(L.h2_dev_ntt(a.data_ptr(), tmp.data_ptr(), _fr(omega), log_n, ST), "ntt")
    fwd = a.clone()
    check(L.h2_dev_intt(a.data_ptr(), tmp.data_ptr(), _fr(pow(omega, -1, R_MOD)), _fr(pow(n, -1, R_MOD)), log_n, ST), "intt")
    torch.cuda.synchronize()
    bad = (a != x).any(dim=1)
    print("log_n %d: intt(ntt(x)) == x: %s (%d rows differ, first %s)" % (log_n, not bool(bad.any()), int(bad.sum()),
          bad.nonzero()[:4].flatten().tolist()))
    b = fwd.clone()
    check(L.h2_dev_ntt(b.data_ptr(), tmp.data_ptr(), _fr(pow(omega, -1, R_MOD)), log_n, ST), "ntt inv")
    check(L.h2_dev_eval_op(0, b.data_ptr(), b.data_ptr(), None, 0, 0, n, _fr(pow(n, -1, R_MOD)), ST), "scale")
    torch.cuda.synchronize()
    bad2 = (b != x).any(dim=1)
    print("          ntt(omega^-1) * n^-1 == x: %s (%d rows differ)" % (not bool(bad2.any()), int(bad2.sum())))
    del x, a, b, fwd, tmp
    torch.cuda.empty_cache()
