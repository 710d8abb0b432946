"""coeff_to_coset (clone + distribute_powers + n-point NTT) against coeff_to_extended, per polynomial: where does the coset
route of the extended-domain phase spend its time?   usage: python tools/experiments/coset_micro.py [k] [degree]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before HIP initialises (halo2-gpu-specific_amd/__init__.py says why)
import torch  # noqa: E402

from halo2_gpu_specific_amd import prover  # noqa: E402
from halo2_gpu_specific_amd._lib import check  # noqa: E402
from halo2_gpu_specific_amd.prover import _fr, R_MOD  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 22
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 3
D = prover.Device()
dom = prover.Domain(k, deg)
n = dom.n
poly = D.empty(n)
check(D.L.h2_dev_random_fr(b"\x01" * 32, n, poly.data_ptr(), D.stream), "rnd")


def timed(fn, reps=5):
    fn(); D.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    D.sync()
    return (time.perf_counter() - t0) / reps * 1e3


print("k=%d degree=%d extended_k=%d cosets needed=%d of %d" % (k, deg, dom.extended_k, dom.quotient_poly_degree, 1 << (dom.extended_k - k)))
print("coeff_to_extended      %.3f ms" % timed(lambda: D.coeff_to_extended(poly, dom)))
print("coeff_to_coset (j=1)   %.3f ms" % timed(lambda: D.coeff_to_coset(poly, dom, 1)))
out = D.clone(poly)
tmp = D.empty(n)
g = dom.g_coset * pow(dom.extended_omega, 1, R_MOD) % R_MOD
print("  clone                %.3f ms" % timed(lambda: D.clone(poly)))
print("  distribute_powers    %.3f ms" % timed(lambda: check(D.L.h2_dev_distribute_powers(out.data_ptr(), n, _fr(g), D.stream), "dp")))
print("  ntt n                %.3f ms" % timed(lambda: check(D.L.h2_dev_ntt(out.data_ptr(), tmp.data_ptr(), _fr(dom.omega), dom.k, D.stream), "ntt")))
print("  empty(n)             %.3f ms" % timed(lambda: D.empty(n)))
