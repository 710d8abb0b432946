"""GPU parity of evaluate_h (h2_evaluate_h through the C ABI) against the CPU oracle, bit-exact."""
import numpy as np
import pytest

from evalh_cases import oracle_evaluate_h, random_case
from halo2_gpu_specific_amd import evaluation as ev

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,k,ek", [(1, 2, 3), (2, 5, 7), (3, 8, 10), (4, 10, 12), (5, 12, 14), (6, 13, 13)])
def test_evaluate_h_vs_oracle(oracle, seed, k, ek):
    kw = random_case(seed, k, ek, oracle, n_calcs=40)
    b = ev.Builder().build(**kw)
    assert np.array_equal(ev.evaluate_h(b), oracle_evaluate_h(oracle, b))


def test_evaluate_h_shapes(oracle):
    for kwargs in (dict(with_perm=False), dict(lookup_sets=()), dict(n_shuffles=0), dict(lookup_sets=(2,), n_shuffles=1),
                   dict(with_perm=False, lookup_sets=(), n_shuffles=0, n_calcs=3)):
        kw = random_case(11, 6, 8, oracle, **kwargs)
        b = ev.Builder().build(**kw)
        assert np.array_equal(ev.evaluate_h(b), oracle_evaluate_h(oracle, b)), kwargs


def test_evaluate_h_k16(oracle):
    kw = random_case(21, 16, 18, oracle, n_calcs=60)
    b = ev.Builder().build(**kw)
    assert np.array_equal(ev.evaluate_h(b), oracle_evaluate_h(oracle, b))


@pytest.mark.timeout(900)
def test_evaluate_h_k20(oracle):
    """the size class of the proofs (2^20 rows, 2^22 extended points, permutation + lookups + shuffles)"""
    kw = random_case(23, 20, 22, oracle, n_calcs=40)
    b = ev.Builder().build(**kw)
    assert np.array_equal(ev.evaluate_h(b), oracle_evaluate_h(oracle, b))


GENERATOR_SETTINGS = [
    {},                                   # the defaults: one stage, factors grouped
    {"H2_JIT_FACTOR": "0"},               # the reference's own fold order (every term folded by y, factors multiplied per term)
    {"H2_JIT_STAGE_PRODUCTS": "8"},       # cut into many stages that each add their share into `values`
    {"H2_JIT_STAGE_PRODUCTS": "20", "H2_JIT_GROUP": "3", "H2_JIT_MAX_AHEAD": "2", "H2_JIT_GAP": "4"},
    {"H2_JIT_INLINE_MULS": "1000"},       # the multiplier inlined at every product
    {"H2_JIT_MUL2": "0"},                 # no a b + c d fusion: every product with its own reduction
    {"H2_JIT_LDS_ARGS": "1"},             # scalars and column pointers read through an LDS copy of the argument block
    {"H2_JIT_LIVE": "6", "H2_JIT_GAP": "200"},   # a tiny live budget: loaded values dropped and loaded again by Belady's rule
]


@pytest.mark.parametrize("seed,k,ek,kwargs", [
    (1, 2, 3, {}), (2, 5, 7, {}), (4, 10, 12, {}), (6, 13, 13, {}), (7, 6, 8, dict(with_perm=False, lookup_sets=(), n_shuffles=0, n_calcs=3)),
    (8, 9, 11, dict(lookup_sets=(2,), n_shuffles=1, n_calcs=80)), (9, 16, 18, dict(n_calcs=60))])
def test_generated_kernels_match_interpreter_and_oracle(oracle, monkeypatch, tmp_path, seed, k, ek, kwargs):
    """the program as straight-line HIP generated, compiled (hipRTC) and cached INSIDE the library (csrc/evalh_gen.cpp) the first
    time its descriptor arrives -- nothing passed in, `reserved` NULL: every opcode, challenge powers, rotations, the lookup /
    shuffle result calculations, the argument terms; grouped by factor or in the reference's fold order, in one stage or many --
    same bits as the interpreter kernels (H2_EVALH_INTERPRET) and as the CPU oracle, and the generated kernels are what ran"""
    monkeypatch.setenv("H2_JIT_CACHE", str(tmp_path))
    kw = random_case(seed, k, ek, oracle, **({"n_calcs": 40} | kwargs))
    b = ev.Builder().build(**kw)
    want = oracle_evaluate_h(oracle, b)
    before = ev.generated_launches()
    assert np.array_equal(ev.evaluate_h(ev.Builder().build(**kw, flags=ev.EVALH_INTERPRET)), want)
    assert ev.generated_launches() == before, "H2_EVALH_INTERPRET must keep the interpreter kernels"
    for env in GENERATOR_SETTINGS:
        for name, value in env.items():
            monkeypatch.setenv(name, value)
        info = ev.prepare(b)
        assert info["stages"] >= (2 if "H2_JIT_STAGE_PRODUCTS" in env and info["products_per_row"] > 40 else 1) and info["scratch_bytes"] == 0
        before = ev.generated_launches()
        assert np.array_equal(ev.evaluate_h(b), want), env
        assert ev.generated_launches() == before + info["stages"], "the generated kernels did not run"
        assert ev.prepare(b)["from_cache"] == 1
        for name in env:
            monkeypatch.delenv(name)
    monkeypatch.setenv("H2_EVALH_JIT", "0")
    before = ev.generated_launches()
    assert np.array_equal(ev.evaluate_h(b), want)
    assert ev.generated_launches() == before


def test_generated_kernels_disk_cache(oracle, monkeypatch, tmp_path):
    """a second process (here: a second option set that hashes differently, then the same one again after the memory cache is
    bypassed by a fresh hash) finds the code objects on disk: one file per program, written atomically, mode of a private dir"""
    import os

    monkeypatch.setenv("H2_JIT_CACHE", str(tmp_path))
    kw = random_case(77, 6, 8, oracle, n_calcs=30)
    b = ev.Builder().build(**kw)
    info = ev.compile_only(b)
    assert info["from_cache"] == 0 and len(os.listdir(tmp_path)) == 1
    assert ev.compile_only(b)["from_cache"] == 2
    assert ev.prepare(b)["from_cache"] == 2               # the first load on this device: from the file, no compile
    assert np.array_equal(ev.evaluate_h(b), oracle_evaluate_h(oracle, b))


def test_plan_cache_is_bounded(tmp_path):
    """the in-memory cache of loaded code objects has a bound (H2_EVALH_PLANS_MAX; least recently used, unheld plans are
    unloaded after their device has drained): tests/evalh_plans_worker.py in a process of its own with a bound of two"""
    import os
    import subprocess
    import sys

    env = dict(os.environ, H2_EVALH_PLANS_MAX="2", H2_JIT_CACHE=str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "evalh_plans_worker.py")],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "plans worker ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_without_hiprtc_the_interpreter_kernels_run(tmp_path):
    """a machine without libhiprtc.so (H2_HIPRTC_LIB names a file that is not there; a process of its own, the library loads
    hipRTC once): one warning per program, the interpreter kernels, the oracle's bits -- this path used to crash building its
    own error message"""
    import os
    import subprocess
    import sys

    env = dict(os.environ, H2_HIPRTC_LIB=str(tmp_path / "no_such_libhiprtc.so"), H2_JIT_CACHE=str(tmp_path / "c"))
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "evalh_plans_worker.py")],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "no-hiprtc worker ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert "runs on the interpreter kernels" in r.stderr and "could not be loaded" in r.stderr, r.stderr[-2000:]


@pytest.mark.parametrize("seed,j,k,kwargs", [(31, 3, 5, {}), (32, 5, 8, {}), (33, 9, 11, dict(n_calcs=60)), (34, 2, 6, {}),
                                             (35, 5, 7, dict(with_perm=False, lookup_sets=(), n_shuffles=0, n_calcs=5)),
                                             (36, 4, 14, dict(lookup_sets=(2,), n_shuffles=1))])
def test_evaluate_h_from_coefficient_forms(oracle, seed, j, k, kwargs):
    """h2_evaluate_h_coeff -- coefficient forms in, the numerator on the extended domain out, computed coset by coset
    (the cuda evaluate_h's shape, plonk/evaluation.rs:1229-1241) -- against the oracle's evaluate_h on the oracle's own
    extended cosets (coeff_to_extended, poly/domain.rs:270-287): 2, 4, 4, 1 (extended_k = k), 4 and 4 cosets"""
    import copy

    d, _ = oracle.domain(j, k)
    ek = d.extended_k
    kw = random_case(seed, k, ek, oracle, **({"n_calcs": 40} | kwargs))
    kw["zeta"], kw["extended_omega"] = d.fr("g_coset"), d.fr("extended_omega")
    n = 1 << k
    names = ("fixed", "advice", "instance", "perm_z", "perm_sigma", "lookup_z", "lookup_m", "shuffle_z")
    coeff = copy.copy(kw)
    for name in names:
        coeff[name] = [np.ascontiguousarray(col[:n]) for col in kw[name]]           # random coefficient vectors
    coeff["l0"], coeff["l_last"] = np.ascontiguousarray(kw["l0"][:n]), np.ascontiguousarray(kw["l_last"][:n])
    ext = copy.copy(kw)
    for name in names:
        ext[name] = [oracle.coeff_to_extended(col, d, threads=8) for col in coeff[name]]
    ext["l0"], ext["l_last"] = oracle.coeff_to_extended(coeff["l0"], d, threads=8), oracle.coeff_to_extended(coeff["l_last"], d, threads=8)
    want = oracle_evaluate_h(oracle, ev.Builder().build(**ext))
    got = ev.evaluate_h_coeff(ev.Builder().build(**coeff))
    assert np.array_equal(got, want)
    assert np.array_equal(ev.evaluate_h(ev.Builder().build(**ext)), want)            # the extended-coset entry point agrees
    # h2_quotient_poly_coeff: the same evaluation, divided by the vanishing polynomial and taken back to coefficient form without
    # leaving the device -- against the oracle's divide_by_vanishing_poly (poly/domain.rs:354-373) and extended_to_coeff
    # (:328-350) applied to the oracle's numerator
    import ctypes

    from halo2_gpu_specific_amd._lib import lib

    _, t = oracle.domain(j, k)
    divided = want.copy()
    oracle.lib.oracle_divide_by_vanishing_poly(divided.ctypes.data, len(divided), t.ctypes.data, len(t), 8)
    want_coeff = oracle.extended_to_coeff(divided, d, threads=8)
    b = ev.Builder().build(**coeff)
    out = np.empty((len(want_coeff), 4), dtype=np.uint64)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)                                  # noqa: E731
    scal = [d.fr(name) for name in ("g_coset", "g_coset_inv", "extended_omega_inv", "extended_ifft_divisor")]
    assert lib().h2_quotient_poly_coeff(ctypes.byref(b.desc), vp(t), len(t), *[vp(v) for v in scal], vp(out), len(out)) == 0
    assert np.array_equal(out, want_coeff)
    assert lib().h2_quotient_poly_coeff(ctypes.byref(b.desc), vp(t), 3, *[vp(v) for v in scal], vp(out), len(out)) != 0   # t_len must divide


def test_host_lincomb(oracle):
    """h2_lincomb (host buffers; the GWC cuda branch's shape, gwc/prover.rs:57-151): res = sum_j coeffs[j] * polys[j]"""
    import ctypes

    import halo2_gpu_specific_amd as h2
    from halo2_gpu_specific_amd import arithmetic as ar

    n, count = 5000, 5
    polys = [oracle.random_fr(700 + i, n) for i in range(count)]
    coeffs = oracle.random_fr(777, count)
    want = np.zeros((n, 4), dtype=np.uint64)
    for p, c in zip(polys, coeffs):
        want = oracle.eval_op(ar.OP_SUM, want, oracle.eval_op(ar.OP_MUL_C, p, None, 0, 0, c), 0, 0, None)
    res = np.zeros((n, 4), dtype=np.uint64)
    ptrs = (ctypes.c_void_p * count)(*[p.ctypes.data for p in polys])
    assert h2.lib().h2_lincomb(res.ctypes.data, ptrs, coeffs.ctypes.data, count, n) == 0
    assert np.array_equal(res, want)


@pytest.mark.parametrize("seed,k,ek", [(41, 6, 8), (42, 10, 12), (43, 13, 14)])
def test_row_ranges_of_the_device_evaluator(oracle, monkeypatch, seed, k, ek):
    """h2_evalh_desc::row_begin / row_count (one evaluation split over several devices by row range): the interpreter, the
    generated kernel and its many-stage form (every stage after the first ADDS into `values`) write exactly the rows asked
    for -- the oracle's values there, the buffer untouched elsewhere -- for ranges at the start, across the wrap of the
    rotations, a single row and the whole domain"""
    import ctypes

    import torch

    import halo2_gpu_specific_amd as h2
    from halo2_gpu_specific_amd._lib import check

    kw = random_case(seed, k, ek, oracle, n_calcs=30, lookup_sets=(1, 2), n_shuffles=1)
    want = oracle_evaluate_h(oracle, ev.Builder().build(**kw))
    size = 1 << ek
    dev = torch.device("cuda", 0)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(dev)  # noqa: E731
    names = ("fixed", "advice", "instance", "perm_z", "perm_sigma", "lookup_z", "lookup_m", "shuffle_z")
    tensors = {name: [up(c) for c in kw[name]] for name in names}
    singles = {name: up(kw[name]) for name in ("l0", "l_last", "l_active_row")}
    dkw = dict(kw)
    for name in names:
        dkw[name] = [t.data_ptr() for t in tensors[name]]
    for name in singles:
        dkw[name] = singles[name].data_ptr()
    L = h2.lib()
    sentinel = 0x5A5A5A5A5A5A5A5A
    for flags, env in ((ev.EVALH_INTERPRET, {}), (0, {}), (0, {"H2_JIT_STAGE_PRODUCTS": "10"})):
        for name, value in env.items():
            monkeypatch.setenv(name, value)
        for lo, cnt in ((0, size // 4), (size - 5, 5), (size // 2 - 3, 1), (7, size - 7), (0, 0)):
            b = ev.Builder().build(**dkw, flags=flags, row_begin=lo, row_count=cnt)
            out = torch.full((size, 4), sentinel, dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            check(L.h2_dev_evaluate_h(ctypes.byref(b.desc), out.data_ptr(), None), "h2_dev_evaluate_h")
            torch.cuda.synchronize()
            got = out.cpu().numpy().view(np.uint64)
            hi = lo + cnt if cnt else size
            lo_ = lo if cnt else 0
            assert np.array_equal(got[lo_:hi], want[lo_:hi]), (flags, env, lo, cnt)
            rest = np.concatenate([got[:lo_], got[hi:]])
            assert (rest == np.uint64(sentinel)).all(), "rows outside the range were written"
    # the host-buffer entry points take no row range
    bad = ev.Builder().build(**kw, row_begin=0, row_count=4)
    assert L.h2_evaluate_h(ctypes.byref(bad.desc), np.zeros((size, 4), dtype=np.uint64).ctypes.data) != 0
