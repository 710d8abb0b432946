"""Run in a child process with HALO2_PROOFS_N_GPU=4 (tests/test_gpu_pool_split.py): the in-call device splits of the host-slice
entry points on ONE visible GPU -- pool entries wrap modulo the visible devices as `devices[gpu_idx % devices.len()]` does
(arithmetic.rs:355), so four leases over two host-API slots of the one device run the real split."""
import copy
import sys

import numpy as np

sys.path.insert(0, sys.argv[1])
sys.path.insert(0, sys.argv[1] + "/tests")

import halo2_gpu_specific_amd as h2  # noqa: E402
from evalh_cases import oracle_evaluate_h, random_case  # noqa: E402
from h2util import Oracle  # noqa: E402
from halo2_gpu_specific_amd import evaluation as ev  # noqa: E402

oracle = Oracle.get()
L = h2.lib()
assert L.h2_device_count() == 4, L.h2_device_count()
aff = lambda r: oracle.to_affine(r.reshape(1, 12)).tobytes()  # noqa: E731

# ---- gpu_multiexp_bound's split (arithmetic.rs:413-440): ceil(n / 4) chunks, host fold
for n, seed in ((1 << 15) + 3, 501), (1 << 16, 503), (4, 505), (3, 507), (1, 509), (5, 511):
    s, p = oracle.random_fr(seed, n), oracle.random_g1(seed + 1, n)
    out = np.zeros(12, dtype=np.uint64)
    assert L.h2_msm_multi(s.ctypes.data, p.ctypes.data, n, 254, out.ctypes.data) == 0, L.h2_last_error()
    assert aff(out) == aff(oracle.best_multiexp(s, p)), ("h2_msm_multi", n)
# identity parts: the scalars of two of the four chunks are zero, then all of them
n = 1 << 12
s, p = oracle.random_fr(521, n), oracle.random_g1(522, n)
s[n // 4: 3 * n // 4] = 0
out = np.zeros(12, dtype=np.uint64)
assert L.h2_msm_multi(s.ctypes.data, p.ctypes.data, n, 254, out.ctypes.data) == 0
assert aff(out) == aff(oracle.best_multiexp(s, p))
s[:] = 0
assert L.h2_msm_multi(s.ctypes.data, p.ctypes.data, n, 254, out.ctypes.data) == 0
assert aff(out) == aff(oracle.best_multiexp(s, p))
print("h2_msm_multi: 4 parts ok")

# ---- h2_evaluate_h_coeff: the cosets of the extended domain dealt over the pool (evaluation.rs:1262-1275,1513-1520)
before = ev.generated_launches()
for seed, j, k, kwargs in ((31, 3, 5, {}), (32, 5, 8, {}), (33, 9, 11, dict(n_calcs=60)), (34, 2, 6, {}), (36, 4, 14, dict(lookup_sets=(2,), n_shuffles=1)),
                           (37, 17, 9, {})):
    d, _ = oracle.domain(j, k)
    ek = d.extended_k
    kw = random_case(seed, k, ek, oracle, **({"n_calcs": 40} | kwargs))
    kw["zeta"], kw["extended_omega"] = d.fr("g_coset"), d.fr("extended_omega")
    n = 1 << k
    names = ("fixed", "advice", "instance", "perm_z", "perm_sigma", "lookup_z", "lookup_m", "shuffle_z")
    coeff = copy.copy(kw)
    for name in names:
        coeff[name] = [np.ascontiguousarray(col[:n]) for col in kw[name]]
    coeff["l0"], coeff["l_last"] = np.ascontiguousarray(kw["l0"][:n]), np.ascontiguousarray(kw["l_last"][:n])
    ext = copy.copy(kw)
    for name in names:
        ext[name] = [oracle.coeff_to_extended(col, d, threads=8) for col in coeff[name]]
    ext["l0"], ext["l_last"] = oracle.coeff_to_extended(coeff["l0"], d, threads=8), oracle.coeff_to_extended(coeff["l_last"], d, threads=8)
    want = oracle_evaluate_h(oracle, ev.Builder().build(**ext))
    got = ev.evaluate_h_coeff(ev.Builder().build(**coeff))
    assert np.array_equal(got, want), ("h2_evaluate_h_coeff", seed, 1 << (ek - k))
    print("h2_evaluate_h_coeff: k=%d, %d cosets over the pool ok" % (k, 1 << (ek - k)))
assert ev.generated_launches() > before
print("POOL-SPLIT-OK")
