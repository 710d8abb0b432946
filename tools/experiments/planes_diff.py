"""range-split MSMs over shifted-base tables for several column shapes: prints a digest per (k, column, range); run once with
H2_MSM_REDUCE_PLANES=0 and once with =1 and diff the outputs"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
torch.cuda.init()
from halo2_gpu_specific_amd import prover, parallel
from halo2_gpu_specific_amd._lib import check
from halo2_gpu_specific_amd.transcript import jacobian_to_affine
D = prover.Device()
for k in [int(a) for a in sys.argv[1:]] or (15, 16, 18, 20, 22):
    n = 1 << k
    params = prover.Params.unsafe_setup(D, k, 0x1D0C5F0A3B7E91C2A4D6F8091B2C3D4E5F60718293A4B5C6D7E8F9010203)
    cols = []
    for j in range(4):
        t = D.empty(n)
        check(D.L.h2_dev_random_fr(bytes([j + 1]) * 32, n, t.data_ptr(), D.stream), "rnd")
        cols.append(t)
    with torch.cuda.stream(D.tstream):
        cols[1][: n // 3] = 0
        cols[2][: 2 * n // 3] = 0
        cols[3][n // 4:] = 0
    D.sync()
    for lo, hi in ((0, n), (0, n // 2), (n // 2, n), (0, n // 4), (3 * n // 8, 5 * n // 8)):
        parts = D.msm_partial(cols, params.g_lagrange, lo, hi, 254)
        for j in range(4):
            print(k, j, lo, hi, hashlib.sha256(repr(jacobian_to_affine(parts[j])).encode()).hexdigest()[:16], flush=True)
