"""MSM timing on the scalar distributions a proof actually commits (one MI355X):
   uniform | sparse small values (a witness column) | mostly-constant 254-bit (a grand-product column with padding rows)
usage: python tools/sparse_msm_probe.py [k]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before HIP initialises (halo2-gpu-specific_amd/__init__.py says why)
import torch  # noqa: E402

torch.cuda.init()
import numpy as np  # noqa: E402

from halo2_gpu_specific_amd import prover  # noqa: E402
from halo2_gpu_specific_amd._lib import check  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << k
D = prover.Device()
L = D.L
params = prover.Params.synthetic(D, k)


def timed(name, col, bits, reps=3):
    D.msm(col, params.g_lagrange, n, bits)
    D.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        D.msm(col, params.g_lagrange, n, bits)
    D.sync()
    print("%-52s %7.2f ms" % (name, (time.perf_counter() - t0) / reps * 1e3))


uniform = D.empty(n)
check(L.h2_dev_random_fr(1, n, uniform.data_ptr(), D.stream), "rnd")
timed("uniform 254-bit", uniform, 254)
small = np.zeros((n, 4), dtype=np.uint64)
small[: n // 8, 0] = np.tile(np.array([5, 25, 30, 5], dtype=np.uint64), n // 32)
small[n - 6:, 0] = [60000, 12345, 3, 40000, 5, 65535]
t = D.upload(small)
check(L.h2_dev_batch_mont(t.data_ptr(), n, D.stream), "mont")
timed("sparse small values (16-bit bound)", t, 16)
timed("sparse small values (254-bit bound)", t, 254)
mostly = D.clone(uniform)
with torch.cuda.stream(D.tstream):
    mostly[n // 8:] = mostly[0]
timed("7/8 of the rows one 254-bit value", mostly, 254)
with torch.cuda.stream(D.tstream):
    mostly[:] = mostly[0]
timed("every row the same 254-bit value", mostly, 254)
boolean = np.zeros((n, 4), dtype=np.uint64)
boolean[::2, 0] = 1
t = D.upload(boolean)
check(L.h2_dev_batch_mont(t.data_ptr(), n, D.stream), "mont")
timed("boolean column (254-bit bound)", t, 254)
# a few distinct wide values (an opcode / state column of field-sized constants): heavy buckets in every window
for distinct in (4, 16, 256, 4096):
    col = D.clone(uniform)
    with torch.cuda.stream(D.tstream):
        idx = (torch.arange(n, device=D.dev) * 2654435761 % distinct)
        col[:] = uniform[idx]
    timed("%d distinct 254-bit values" % distinct, col, 254)
# sparse wide values: 1/10 of the rows random, the rest zero
col = D.clone(uniform)
with torch.cuda.stream(D.tstream):
    mask = (torch.arange(n, device=D.dev) % 10) != 0
    col[mask] = 0
timed("1/10 of the rows 254-bit, rest zero", col, 254)
# half the rows one value, half uniform (dominant value + uniform remainder)
col = D.clone(uniform)
with torch.cuda.stream(D.tstream):
    col[::2] = uniform[7]
timed("half the rows one 254-bit value", col, 254)
