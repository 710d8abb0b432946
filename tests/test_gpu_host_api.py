"""The literal drop-in's data flow (halo2-gpu-specific_amd/host_api.py): every polynomial in host memory, every vector
operation one host-slice entry point of include/halo2_hip.h -- the calls `integration/hip.rs` binds -- including the cuda
shape of the evaluator (coefficient forms in, one h2_evaluate_h_coeff call).  Same SRS, witness and randomness as the
device-resident prover: the proof bytes must be equal (mini-PLONK, the lookup / shuffle / instance circuit, the wide
circuit; GWC and SHPLONK; two circuit instances)."""
import numpy as np
import pytest

import ref_plonk as rp
from test_gpu_plonk import cols_to_arr, srs
from test_plonk_host import lookup_shuffle_cs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def device():
    from halo2_gpu_specific_amd import prover

    return prover.Device()


@pytest.mark.parametrize("which,k", [("mini", 9), ("lookup", 8), ("wide", 9), ("mini", 14)])
def test_host_slice_api_gives_the_resident_provers_bytes(oracle, device, which, k):
    from halo2_gpu_specific_amd import circuits, host_api, prover
    from halo2_gpu_specific_amd.rng import ProverRng

    inst = ()
    if which == "mini":
        cs, (adv, fixed, copies) = circuits.mini_plonk(), circuits.mini_plonk_synthesize(k)
    elif which == "wide":
        cs, (adv, fixed, copies) = circuits.wide(4), circuits.wide_synthesize(k, 4)
    else:
        cs = lookup_shuffle_cs()
        syn = rp.LookupShuffle.synthesize(k)
        adv, fixed = cols_to_arr(syn[0]), cols_to_arr(syn[1])
        copies, inst = [(l[0], l[1], r[0], r[1]) for l, r in syn[2]], syn[3]
    params = srs(oracle, device, k)
    pk = prover.keygen(device, params, cs, fixed, copies)
    H = host_api.HostApiDevice()
    hparams = host_api.params_like(H, params)
    hpk = prover.keygen(H, hparams, cs, fixed, copies)
    assert hpk.transcript_repr == pk.transcript_repr and hpk.fixed_commitments == pk.fixed_commitments
    assert hpk.perm_commitments == pk.perm_commitments
    for seed, gwc in ((1, False), (2, True)):
        want = prover.create_proof_ext(device, params, pk, adv, ProverRng(seed), gwc, instances=inst)
        got = prover.create_proof_ext(H, hparams, hpk, adv, ProverRng(seed), gwc, instances=inst)
        first = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), None)
        assert len(got) == len(want) and first is None, "host-slice proof differs at byte %s" % first
    calls = H.L.calls
    assert calls["h2_quotient_poly_coeff"] == 2 and "h2_evaluate_h_coeff" not in calls and calls["h2_msm"] > 10 and calls["h2_intt_to"] > 3 and calls["h2_permutation_product"] >= 2
    assert calls["h2_msm_intt"] >= 2                   # the product columns: commitment + coefficient form in one call each
    assert calls["h2_quotient_sum"] >= 3               # (SHPLONK: the rotation sets' quotients + the final one; GWC: one per point)
    assert "h2_kate_division" not in calls
    if which in ("lookup", "wide"):                    # the lookups' grand sums: one call per input set, nothing step by step
        assert calls["h2_logup_grand_sum"] >= 2 and "h2_prefix_sum" not in calls
    assert "h2_permutation_terms" not in calls and not any(name.startswith("oracle") for name in calls)
    # the permutation products step by step (h2_permutation_terms, the shared batch inversion, h2_eval_op, h2_prefix_product):
    # the same bytes
    S = host_api.HostApiDevice(fused_permutation=False)
    sparams = host_api.params_like(S, params)
    spk = prover.keygen(S, sparams, cs, fixed, copies)
    assert prover.create_proof_ext(S, sparams, spk, adv, ProverRng(2), True, instances=inst) == want
    assert S.L.calls["h2_permutation_terms"] >= 1 and "h2_permutation_product" not in S.L.calls and "h2_quotient_sum" not in S.L.calls
    assert S.L.calls["h2_evaluate_h_coeff"] == 1 and S.L.calls["h2_divide_by_vanishing_poly"] == 1 and S.L.calls["h2_extended_to_coeff"] == 1
    if which == "mini" and k == 9:          # two circuit instances in one proof
        adv2 = circuits.mini_plonk_synthesize(k, a=9)[0]
        want = prover.create_proof_ext(device, params, pk, [adv, adv2], ProverRng(5), False, instances=[(), ()])
        assert prover.create_proof_ext(H, hparams, hpk, [adv, adv2], ProverRng(5), False, instances=[(), ()]) == want


def test_host_slice_device_needs_a_gpu_and_touches_no_oracle():
    import inspect

    from halo2_gpu_specific_amd import host_api

    src = inspect.getsource(host_api)
    assert "oracle" not in src.replace("no oracle", "") and "liboracle" not in src


def _ptr(a):
    import ctypes

    return a.ctypes.data_as(ctypes.c_void_p)


def _evalpoly(L, poly, x):
    out = np.zeros(4, dtype=np.uint64)
    assert L.h2_eval_polynomial(_ptr(poly), len(poly), _ptr(x), _ptr(out)) == 0
    return out


def _oracle_evalpoly(oracle, poly, x):
    out = np.zeros(4, dtype=np.uint64)
    oracle.lib.oracle_eval_polynomial(_ptr(poly), len(poly), _ptr(x), _ptr(out))
    return out


@pytest.mark.parametrize("n", [1 << 12, (1 << 21) + 4097])
def test_registered_polynomials_are_read_on_the_device_and_never_served_stale(oracle, n):
    """h2_poly_register (the proving key's fixed / sigma / l_0 / l_last coefficient forms, plonk.rs:226-240, read by every
    proof): a registered vector gives the entry points that READ it -- h2_eval_polynomial, h2_lincomb operands (whole and
    sub-ranges), h2_kate_division, h2_eval_op -- the oracle's results from a device copy made once; after unregister the host
    contents count again; registering the same address anew (another generation) uploads the new contents: a copy made for an
    earlier registration is never used."""
    import ctypes

    import halo2_gpu_specific_amd as h2
    from halo2_gpu_specific_amd import arithmetic as ar

    L = h2.lib()
    p, q = oracle.random_fr(9000 + n % 97, n), oracle.random_fr(9100 + n % 97, n)
    x, coeffs = oracle.random_fr(9200, 1)[0], oracle.random_fr(9300, 2)
    assert L.h2_poly_register(_ptr(p), n) == 0
    try:
        for _ in range(2):                                  # the second round runs from the device copy
            assert np.array_equal(_evalpoly(L, p, x), _oracle_evalpoly(oracle, p, x))
        # operands: one registered (whole), one not; then a sub-range of the registered vector
        for lo, m in ((0, n), (n // 4, n // 2)):
            want = oracle.eval_op(ar.OP_SUM, oracle.eval_op(ar.OP_MUL_C, p[lo:lo + m], None, 0, 0, coeffs[0]),
                                  oracle.eval_op(ar.OP_MUL_C, q[lo:lo + m], None, 0, 0, coeffs[1]), 0, 0, None)
            res = np.zeros((m, 4), dtype=np.uint64)
            ptrs = (ctypes.c_void_p * 2)(p[lo:].ctypes.data, q[lo:].ctypes.data)
            assert L.h2_lincomb(_ptr(res), ptrs, _ptr(coeffs), 2, m) == 0
            assert np.array_equal(res, want), (lo, m)
        got = np.zeros((n, 4), dtype=np.uint64)
        want = np.zeros((n, 4), dtype=np.uint64)
        assert L.h2_kate_division(_ptr(p), n, _ptr(x), _ptr(got)) == 0
        oracle.lib.oracle_kate_division(_ptr(p), n, _ptr(x), _ptr(want))
        assert np.array_equal(got[:n - 1], want[:n - 1])
        assert np.array_equal(ar.eval_op(ar.OP_MUL, p, q), oracle.eval_op(ar.OP_MUL, p, q, 0, 0, None))
        assert np.array_equal(ar.eval_op(ar.OP_SUM, q, p, 0, 3), oracle.eval_op(ar.OP_SUM, q, p, 0, 3, None))   # rotated: single shot
    finally:
        assert L.h2_poly_unregister(_ptr(p)) == 0
    # unregistered: the host contents are what counts again
    p[5] = q[7]
    assert np.array_equal(_evalpoly(L, p, x), _oracle_evalpoly(oracle, p, x))
    # a new registration of the same address: new generation, new upload -- never the copy of the first one
    assert L.h2_poly_register(_ptr(p), n) == 0
    try:
        assert np.array_equal(_evalpoly(L, p, x), _oracle_evalpoly(oracle, p, x))
        assert L.h2_poly_register(_ptr(p), n // 2) == 0     # registered again (replaced) without unregistering: half the length
        p2 = p.copy()
        assert np.array_equal(_evalpoly(L, p, x), _oracle_evalpoly(oracle, p2, x))      # whole vector: outside the range, uploaded
        assert np.array_equal(_evalpoly(L, p[:n // 2], x), _oracle_evalpoly(oracle, p2[:n // 2], x))
    finally:
        assert L.h2_poly_unregister(_ptr(p)) == 0
    assert L.h2_poly_unregister(_ptr(p)) == 0               # idempotent
    assert L.h2_poly_register(None, 4) != 0 and L.h2_poly_register(_ptr(p), 0) != 0


class _Pinned:
    """numpy (n, 4) u64 arrays in page-locked host memory (h2_host_alloc_pinned): what the chunk pipeline needs to run at all"""

    def __init__(self, L):
        self.L, self.blocks = L, []

    def like(self, a):
        import ctypes

        p = ctypes.c_void_p()
        assert self.L.h2_host_alloc_pinned(a.nbytes, ctypes.byref(p)) == 0
        self.blocks.append(p)
        out = np.ctypeslib.as_array((ctypes.c_uint64 * a.size).from_address(p.value)).reshape(a.shape)
        out[...] = a
        return out

    def free(self):
        for p in self.blocks:
            self.L.h2_host_free_pinned(p)


@pytest.mark.parametrize("pinned", [True, False])
def test_pipelined_elementwise_entry_points_at_long_sizes_vs_oracle(oracle, pinned):
    """from 2^21 elements on, h2_lincomb / h2_eval_op / h2_batch_mont / h2_batch_unmont / h2_divide_by_vanishing_poly /
    h2_permutation_terms run chunk by chunk over three streams when their host vectors are PAGE-LOCKED (upload of chunk c + 1,
    kernel of chunk c, download of chunk c - 1) and single shot from ordinary memory: a ragged length (the last chunk short),
    in-place forms, the table of the vanishing division and the omega powers of the permutation terms across chunk boundaries,
    both ways against the oracle"""
    import ctypes

    import halo2_gpu_specific_amd as h2
    from halo2_gpu_specific_amd import arithmetic as ar

    L = h2.lib()
    pool = _Pinned(L)
    host = pool.like if pinned else (lambda v: v)
    try:
        _pipelined_cases(oracle, L, ar, host, ctypes)
    finally:
        pool.free()


def _pipelined_cases(oracle, L, ar, host, ctypes):
    n = (1 << 21) + (1 << 19) + 12345
    a, b, c3 = host(oracle.random_fr(9500, n)), host(oracle.random_fr(9501, n)), host(oracle.random_fr(9502, n))
    coeffs = oracle.random_fr(9503, 3)
    want = np.zeros((n, 4), dtype=np.uint64)
    for v, cf in zip((a, b, c3), coeffs):
        want = oracle.eval_op(ar.OP_SUM, want, oracle.eval_op(ar.OP_MUL_C, v, None, 0, 0, cf), 0, 0, None)
    res = host(np.zeros((n, 4), dtype=np.uint64))
    ptrs = (ctypes.c_void_p * 3)(a.ctypes.data, b.ctypes.data, c3.ctypes.data)
    assert L.h2_lincomb(_ptr(res), ptrs, _ptr(coeffs), 3, n) == 0
    assert np.array_equal(res, want)
    assert L.h2_eval_op(ar.OP_MUL, _ptr(res), _ptr(a), _ptr(b), 0, 0, n, None) == 0
    assert np.array_equal(res, oracle.eval_op(ar.OP_MUL, a, b, 0, 0, None))
    assert L.h2_eval_op(ar.OP_LCTHETA, _ptr(res), _ptr(a), _ptr(b), 0, 0, n, _ptr(coeffs[0])) == 0
    assert np.array_equal(res, oracle.eval_op(ar.OP_LCTHETA, a, b, 0, 0, coeffs[0]))
    inplace = host(a.copy())
    assert L.h2_eval_op(ar.OP_SUB, _ptr(inplace), _ptr(inplace), _ptr(b), 0, 0, n, None) == 0          # res aliases l
    assert np.array_equal(inplace, oracle.eval_op(ar.OP_SUB, a, b, 0, 0, None))
    raw = oracle.random_fr(9504, n)
    m = host(raw.copy())
    assert L.h2_batch_mont(_ptr(m), n) == 0
    one = np.zeros((n, 4), dtype=np.uint64)
    one[:, 0] = 1
    mont_one = ar.gpu_mont(one)
    assert np.array_equal(oracle.eval_op(ar.OP_MUL, np.array(m), mont_one, 0, 0, None), m)                 # m * mont(1) = m
    assert L.h2_batch_unmont(_ptr(m), n) == 0
    assert np.array_equal(m, raw)
    d, t = oracle.domain(5, 20)                       # extended_k = 22: 2^22 values, t_len = 4
    ext = oracle.random_fr(9505, 1 << d.extended_k)
    got = host(ext.copy())
    assert L.h2_divide_by_vanishing_poly(_ptr(got), len(got), _ptr(t), len(t)) == 0
    oracle.lib.oracle_divide_by_vanishing_poly(ext.ctypes.data, len(ext), t.ctypes.data, len(t), 16)
    assert np.array_equal(got, ext)
    # the permutation argument's numerator / denominator products (permutation/prover.rs:89-128): first column of a set, then a
    # second one multiplied in; the numerator's delta^c omega^i must continue across the chunk boundaries
    from h2util import R_MOD, fr_mont

    value, sigma = a, b
    beta, gamma = oracle.random_fr(9506, 2)
    omega = fr_mont(pow(0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C, 1 << (28 - 22), R_MOD))
    dpow = fr_mont(pow(0x09226B6E22C6F0CA64EC26AAD4C86E715B5F898E5E963F25870E56BBE533E9A2, 3, R_MOD))
    num, den = host(np.zeros((n, 4), dtype=np.uint64)), host(np.zeros((n, 4), dtype=np.uint64))
    wn, wd = np.zeros((n, 4), dtype=np.uint64), np.zeros((n, 4), dtype=np.uint64)
    for first, (v_, s_) in ((1, (value, sigma)), (0, (c3, a))):
        assert L.h2_permutation_terms(_ptr(num), _ptr(den), _ptr(v_), _ptr(s_), n, _ptr(beta), _ptr(gamma), _ptr(dpow), _ptr(omega), first) == 0
        oracle.lib.oracle_permutation_terms(_ptr(wn), _ptr(wd), _ptr(np.array(v_)), _ptr(np.array(s_)), n, _ptr(beta), _ptr(gamma), _ptr(dpow),
                                            _ptr(omega), first)
        assert np.array_equal(num, wn) and np.array_equal(den, wd), first


@pytest.mark.parametrize("register", [True, False])
def test_host_slice_proof_at_a_pipelined_size(oracle, device, register):
    """the literal drop-in at k = 21 (every vector 64 MiB: the chunked pipeline and, with `register`, the device copies of the
    proving key's and the proof's final polynomials carry every read): the resident prover's bytes, SHPLONK and GWC"""
    from halo2_gpu_specific_amd import circuits, host_api, prover
    from halo2_gpu_specific_amd.rng import ProverRng

    k = 21
    cs, (adv, fixed, copies) = circuits.mini_plonk(), circuits.mini_plonk_synthesize(k)
    params = prover.Params.unsafe_setup(device, k, 0x1D0C5F0A3B7E91C2A4D6F8091B2C3D4E5F60718293A4B5C6D7E8F9010203)
    pk = prover.keygen(device, params, cs, fixed, copies)
    H = host_api.HostApiDevice(pinned=register, register_polys=register, fused_permutation=register)
    hparams = host_api.params_like(H, params)
    hpk = prover.keygen(H, hparams, cs, fixed, copies)
    for seed, gwc in ((3, False), (4, True)):
        want = prover.create_proof_ext(device, params, pk, adv, ProverRng(seed), gwc)
        assert prover.create_proof_ext(H, hparams, hpk, adv, ProverRng(seed), gwc) == want
    assert not H._retained, "the per-proof registrations are released at the end of the proof"


def test_two_threads_through_the_chunk_pipeline_at_once(oracle):
    """the reference's entry points are called from rayon workers: two threads, each pushing its own page-locked vectors through
    the chunk-pipelined entry points (h2_lincomb, h2_eval_op in place) at the same time -- the two host-API slots of the device
    each run their own three-stream pipeline -- give the oracle's results"""
    import ctypes
    import threading

    import halo2_gpu_specific_amd as h2
    from halo2_gpu_specific_amd import arithmetic as ar

    L = h2.lib()
    pool = _Pinned(L)
    n = (1 << 21) + 77
    errors = []
    try:
        jobs = []
        for t in range(2):
            a, b = pool.like(oracle.random_fr(9700 + t, n)), pool.like(oracle.random_fr(9710 + t, n))
            coeffs = oracle.random_fr(9720 + t, 2)
            want = oracle.eval_op(ar.OP_SUM, oracle.eval_op(ar.OP_MUL_C, a, None, 0, 0, coeffs[0]),
                                  oracle.eval_op(ar.OP_MUL_C, b, None, 0, 0, coeffs[1]), 0, 0, None)
            want_mul = oracle.eval_op(ar.OP_MUL, a, b, 0, 0, None)
            jobs.append((a, b, coeffs, pool.like(np.zeros((n, 4), dtype=np.uint64)), pool.like(np.zeros((n, 4), dtype=np.uint64)), want, want_mul))

        def work(job):
            a, b, coeffs, res, res2, want, want_mul = job
            try:
                ptrs = (ctypes.c_void_p * 2)(a.ctypes.data, b.ctypes.data)
                for _ in range(3):
                    assert L.h2_lincomb(_ptr(res), ptrs, _ptr(coeffs), 2, n) == 0
                    assert np.array_equal(res, want)
                    assert L.h2_eval_op(ar.OP_MUL, _ptr(res2), _ptr(a), _ptr(b), 0, 0, n, None) == 0
                    assert np.array_equal(res2, want_mul)
            except Exception as e:                          # noqa: BLE001
                errors.append(e)

        threads = [threading.Thread(target=work, args=(j,)) for j in jobs]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        assert not errors, errors
    finally:
        pool.free()


@pytest.mark.parametrize("n,count,registered", [(1, 1, False), (2, 2, False), (4097, 3, False), ((1 << 21) + 5, 2, True)])
def test_permutation_product_in_one_call(oracle, n, count, registered):
    """h2_permutation_product (one grand-product column of the permutation argument: the per-column products, the batch inversion,
    the product and the running product on the device, the set's columns up once and z down once) against the oracle's
    steps taken one by one (permutation/prover.rs:72-165); sigma columns registered with h2_poly_register are read on the device"""
    import ctypes

    import halo2_gpu_specific_amd as h2
    from h2util import R_MOD, fr_mont

    L = h2.lib()
    values = [oracle.random_fr(9800 + j, n) for j in range(count)]
    sigmas = [oracle.random_fr(9810 + j, n) for j in range(count)]
    beta, gamma, init = oracle.random_fr(9820, 3)
    delta = 0x09226B6E22C6F0CA64EC26AAD4C86E715B5F898E5E963F25870E56BBE533E9A2
    omega = fr_mont(pow(0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C, 1 << (28 - 22), R_MOD))
    first_col = 4
    num, den = np.zeros((n, 4), dtype=np.uint64), np.zeros((n, 4), dtype=np.uint64)
    for j in range(count):
        dpow = fr_mont(pow(delta, first_col + j, R_MOD))
        oracle.lib.oracle_permutation_terms(_ptr(num), _ptr(den), _ptr(values[j]), _ptr(sigmas[j]), n, _ptr(beta), _ptr(gamma), _ptr(dpow),
                                            _ptr(omega), 1 if j == 0 else 0)
    oracle.lib.oracle_batch_invert(_ptr(den), n)
    import halo2_gpu_specific_amd.arithmetic as ar

    f = oracle.eval_op(ar.OP_MUL, num, den, 0, 0, None)
    want = np.zeros((n, 4), dtype=np.uint64)
    want[0] = init
    if n > 1:
        from oracle_prover import OracleLib

        OracleLib().O.oracle_prefix_product(_ptr(f), n, _ptr(init), _ptr(want))
    if registered:
        for sg in sigmas:
            assert L.h2_poly_register(_ptr(sg), n) == 0
    try:
        z = np.empty((n, 4), dtype=np.uint64)
        vp = (ctypes.c_void_p * count)(*[v.ctypes.data for v in values])
        sp = (ctypes.c_void_p * count)(*[v.ctypes.data for v in sigmas])
        for _ in range(2 if registered else 1):            # (the second call finds the device copies)
            z[:] = 0
            assert L.h2_permutation_product(_ptr(z), vp, sp, count, n, _ptr(beta), _ptr(gamma), _ptr(fr_mont(pow(delta, first_col, R_MOD))),
                                            _ptr(fr_mont(delta)), _ptr(omega), _ptr(init)) == 0
            assert np.array_equal(z, want)
    finally:
        if registered:
            for sg in sigmas:
                assert L.h2_poly_unregister(_ptr(sg)) == 0
    assert L.h2_permutation_product(_ptr(z), vp, None, count, n, _ptr(beta), _ptr(gamma), _ptr(init), _ptr(init), _ptr(omega), _ptr(init)) != 0


@pytest.mark.parametrize("n,registered", [(1, False), (2, False), (4099, False), ((1 << 21) + 3, True)])
def test_quotient_sum_in_one_call(oracle, n, registered):
    """h2_quotient_sum (a multi-point opening's quotient contributions: per rotation set a linear combination, minus a few low
    coefficients, divided by the set's points one after the other; the sets summed; everything on the device, one vector down)
    against the oracle's lincomb / subtraction / Kate divisions taken step by step (shplonk/prover.rs:95-153, :205-219), and the
    remainders it reports against the oracle's Horner evaluations of the dividends"""
    import ctypes

    import halo2_gpu_specific_amd as h2
    from oracle_prover import OracleLib

    L, O = h2.lib(), OracleLib().O
    polys = [oracle.random_fr(9900 + j, n) for j in range(4)]
    sets = [([0, 1, 2], oracle.random_fr(9910, 3), oracle.random_fr(9911, min(2, n)), oracle.random_fr(9912, 2)),
            ([3], oracle.random_fr(9913, 1), np.zeros((0, 4), dtype=np.uint64), oracle.random_fr(9915, 1)),
            ([1, 3], oracle.random_fr(9916, 2), oracle.random_fr(9917, min(1, n)), oracle.random_fr(9918, 3))]
    import halo2_gpu_specific_amd.arithmetic as ar

    want, want_rem = np.zeros((n, 4), dtype=np.uint64), []
    for idx, coeffs, low, points in sets:
        cur = np.zeros((n, 4), dtype=np.uint64)
        ptrs = (ctypes.c_void_p * len(idx))(*[polys[i].ctypes.data for i in idx])
        O.oracle_lincomb(_ptr(cur), ptrs, _ptr(coeffs), len(idx), n)
        if len(low):
            cur[:len(low)] = oracle.eval_op(ar.OP_SUB, cur[:len(low)].copy(), low, 0, 0, None)
        for pt in points:
            rem = np.zeros(4, dtype=np.uint64)
            oracle.lib.oracle_eval_polynomial(_ptr(cur), n, _ptr(pt), _ptr(rem))
            want_rem.append(rem)
            q = np.zeros((n, 4), dtype=np.uint64)
            if n >= 2:
                O.oracle_kate_division(_ptr(cur), n, _ptr(pt), _ptr(q))
            cur = q
        want = oracle.eval_op(ar.OP_SUM, want, cur, 0, 0, None)
    sz = ctypes.c_size_t
    counts = (sz * 3)(*[len(s_[0]) for s_ in sets])
    lows = (sz * 3)(*[len(s_[2]) for s_ in sets])
    pts = (sz * 3)(*[len(s_[3]) for s_ in sets])
    ptrs = (ctypes.c_void_p * 6)(*[polys[i].ctypes.data for s_ in sets for i in s_[0]])
    coeffs = np.concatenate([s_[1] for s_ in sets])
    low = np.concatenate([s_[2].reshape(-1, 4) for s_ in sets])
    points = np.concatenate([s_[3] for s_ in sets])
    if registered:
        for p_ in polys[:3]:
            assert L.h2_poly_register(_ptr(p_), n) == 0
    try:
        for _ in range(2 if registered else 1):
            out, rem = np.empty((n, 4), dtype=np.uint64), np.zeros((6, 4), dtype=np.uint64)
            assert L.h2_quotient_sum(_ptr(out), n, 3, counts, ptrs, _ptr(coeffs), lows, _ptr(low), pts, _ptr(points), _ptr(rem)) == 0
            assert np.array_equal(out, want)
            assert np.array_equal(rem, np.array(want_rem))
            out2 = np.empty((n, 4), dtype=np.uint64)
            assert L.h2_quotient_sum(_ptr(out2), n, 3, counts, ptrs, _ptr(coeffs), lows, _ptr(low), pts, _ptr(points), None) == 0
            assert np.array_equal(out2, want)
    finally:
        if registered:
            for p_ in polys[:3]:
                assert L.h2_poly_unregister(_ptr(p_)) == 0
    assert L.h2_quotient_sum(_ptr(out), n, 3, counts, None, _ptr(coeffs), lows, _ptr(low), pts, _ptr(points), None) != 0
    zero = np.ones((n, 4), dtype=np.uint64)
    assert L.h2_quotient_sum(_ptr(zero), n, 0, None, None, None, None, None, None, None, None) == 0 and not zero.any()


@pytest.mark.parametrize("n", [1, 5, 4097, (1 << 20) + 9])
def test_eval_polynomial_batch_on_host_vectors(oracle, n):
    """h2_eval_polynomial_batch: polynomial j at point j in one call -- a polynomial evaluated at several points goes up once,
    a registered one not at all -- against the oracle's Horner, value by value"""
    import ctypes

    import halo2_gpu_specific_amd as h2

    L = h2.lib()
    polys = [oracle.random_fr(9950 + j, n) for j in range(3)]
    which = [0, 1, 0, 2, 2, 0, 1]
    points = oracle.random_fr(9960, len(which))
    want = np.array([_oracle_evalpoly(oracle, polys[w], points[j]) for j, w in enumerate(which)])
    ptrs = (ctypes.c_void_p * len(which))(*[polys[w].ctypes.data for w in which])
    assert L.h2_poly_register(_ptr(polys[1]), n) == 0
    try:
        for _ in range(2):
            out = np.zeros((len(which), 4), dtype=np.uint64)
            assert L.h2_eval_polynomial_batch(ptrs, len(which), n, _ptr(points), _ptr(out)) == 0
            assert np.array_equal(out, want)
    finally:
        assert L.h2_poly_unregister(_ptr(polys[1])) == 0
    assert L.h2_eval_polynomial_batch(ptrs, 0, n, None, None) == 0
    assert L.h2_eval_polynomial_batch(None, 2, n, _ptr(points), _ptr(out)) != 0


def test_random_fr_into_a_host_vector():
    """h2_random_fr: the keyed stream of h2_dev_random_fr (the vanishing argument's blinding polynomial) in a host vector"""
    import ctypes

    import torch

    import halo2_gpu_specific_amd as h2

    L = h2.lib()
    key = (ctypes.c_uint8 * 32)(*range(7, 39))
    for n in (1, 4097, (1 << 18) + 3):
        dev = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()                      # (torch's fill runs on torch's stream, the generator on the library's)
        assert L.h2_dev_random_fr(key, n, dev.data_ptr(), None) == 0
        torch.cuda.synchronize()
        host = np.zeros((n, 4), dtype=np.uint64)
        assert L.h2_random_fr(key, n, _ptr(host)) == 0
        assert np.array_equal(host, dev.cpu().numpy().view(np.uint64))
    assert L.h2_random_fr(None, 4, _ptr(host)) != 0


def test_fused_entry_points_from_two_threads_at_once(oracle):
    """two threads, each with its own vectors, inside h2_permutation_product / h2_quotient_sum / h2_eval_polynomial_batch at the same
    time (the two host-API slots of the device): every result equals the one the same call gives alone"""
    import ctypes
    import threading

    import halo2_gpu_specific_amd as h2
    from h2util import R_MOD, fr_mont

    L = h2.lib()
    n = (1 << 19) + 11
    delta = fr_mont(0x09226B6E22C6F0CA64EC26AAD4C86E715B5F898E5E963F25870E56BBE533E9A2)
    omega = fr_mont(pow(0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C, 1 << (28 - 20), R_MOD))
    sz = ctypes.c_size_t

    def calls(t):
        vals = [oracle.random_fr(9980 + 10 * t + j, n) for j in range(3)]
        sc = oracle.random_fr(9990 + t, 8)
        vp = (ctypes.c_void_p * 2)(vals[0].ctypes.data, vals[1].ctypes.data)
        sp = (ctypes.c_void_p * 2)(vals[1].ctypes.data, vals[2].ctypes.data)
        z = np.empty((n, 4), dtype=np.uint64)
        assert L.h2_permutation_product(_ptr(z), vp, sp, 2, n, _ptr(sc[0]), _ptr(sc[1]), _ptr(delta), _ptr(delta), _ptr(omega), _ptr(sc[2])) == 0
        q = np.empty((n, 4), dtype=np.uint64)
        ptrs = (ctypes.c_void_p * 3)(*[v.ctypes.data for v in vals])
        assert L.h2_quotient_sum(_ptr(q), n, 2, (sz * 2)(2, 1), ptrs, _ptr(sc[:3]), (sz * 2)(1, 0), _ptr(sc[3:4]), (sz * 2)(2, 1), _ptr(sc[4:7]), None) == 0
        ev = np.zeros((3, 4), dtype=np.uint64)
        assert L.h2_eval_polynomial_batch(ptrs, 3, n, _ptr(sc[:3]), _ptr(ev)) == 0
        return z, q, ev

    alone = [calls(t) for t in range(2)]
    got, errors = [None, None], []

    def work(t):
        try:
            for _ in range(3):
                got[t] = calls(t)
                for a, b in zip(got[t], alone[t]):
                    assert np.array_equal(a, b)
        except Exception as e:                              # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


@pytest.mark.parametrize("n", [1, 2, 4099, (1 << 18) + 5])
def test_host_vector_twins_of_the_lookup_steps(oracle, n):
    """h2_prefix_sum, h2_distribute_powers, h2_permutation_sigma, h2_logup_multiplicity on host vectors against the device entry
    points they wrap (each of those is checked against the oracle in tests/test_gpu_numerics.py)"""
    import ctypes

    import torch

    import halo2_gpu_specific_amd as h2
    from h2util import R_MOD, fr_mont

    L = h2.lib()
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()            # noqa: E731
    host = lambda t: t.cpu().numpy().view(np.uint64)                                           # noqa: E731
    f, init, g = oracle.random_fr(9970, n), oracle.random_fr(9971, 1)[0], oracle.random_fr(9972, 1)[0]
    # grand sum
    z = np.empty((n, 4), dtype=np.uint64)
    assert L.h2_prefix_sum(_ptr(f), n, _ptr(init), _ptr(z)) == 0
    d_f, d_z = dev(f), torch.zeros((n, 4), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    assert L.h2_dev_prefix_sum(d_f.data_ptr(), n, _ptr(init), d_z.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(z, host(d_z))
    # a[i] *= g^i
    a = f.copy()
    assert L.h2_distribute_powers(_ptr(a), n, _ptr(g)) == 0
    d_a = dev(f)
    torch.cuda.synchronize()
    assert L.h2_dev_distribute_powers(d_a.data_ptr(), n, _ptr(g), None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(a, host(d_a))
    # one sigma column
    rng = np.random.default_rng(n)
    mc, mr = rng.integers(0, 5, size=n, dtype=np.uint32), rng.integers(0, n, size=n, dtype=np.uint32)
    delta = fr_mont(0x09226B6E22C6F0CA64EC26AAD4C86E715B5F898E5E963F25870E56BBE533E9A2)
    omega = fr_mont(pow(0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C, 1 << (28 - 19), R_MOD))
    sig = np.empty((n, 4), dtype=np.uint64)
    assert L.h2_permutation_sigma(_ptr(sig), _ptr(mc), _ptr(mr), n, _ptr(delta), _ptr(omega)) == 0
    d_mc, d_mr = torch.from_numpy(mc.view(np.int32)).cuda(), torch.from_numpy(mr.view(np.int32)).cuda()
    d_sig = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    assert L.h2_dev_permutation_sigma(d_sig.data_ptr(), d_mc.data_ptr(), d_mr.data_ptr(), n, _ptr(delta), _ptr(omega), None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(sig, host(d_sig))
    # multiplicities: a table of distinct values, two input columns drawn from it
    usable = max(n - 3, 1) if n > 4 else n
    table = oracle.random_fr(9973, n)
    table[usable:] = table[0]
    ins = [table[rng.integers(0, usable, size=n)] for _ in range(2)]
    m, bits = np.empty((n, 4), dtype=np.uint64), ctypes.c_uint32(99)
    ptrs = (ctypes.c_void_p * 2)(*[c.ctypes.data for c in ins])
    assert L.h2_logup_multiplicity(_ptr(table), ptrs, 2, usable, n, _ptr(m), ctypes.byref(bits)) == 0
    d_t, d_ins = dev(table), [dev(c) for c in ins]
    d_m = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
    nbytes = L.h2_logup_scratch_bytes(n)
    d_sc = torch.zeros(max(nbytes, 256), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    dptrs, dbits = (ctypes.c_void_p * 2)(*[t.data_ptr() for t in d_ins]), ctypes.c_uint32(0)
    assert L.h2_dev_logup_multiplicity_bits(d_t.data_ptr(), dptrs, 2, usable, n, d_m.data_ptr(), d_sc.data_ptr(), nbytes, ctypes.byref(dbits), None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(m, host(d_m)) and bits.value == dbits.value
    absent = [oracle.random_fr(9974, n)]
    ptr1 = (ctypes.c_void_p * 1)(absent[0].ctypes.data)
    assert L.h2_logup_multiplicity(_ptr(table), ptr1, 1, usable, n, _ptr(m), None) != 0          # a value that is not in the table


@pytest.mark.parametrize("n,count,with_table,registered", [(1, 1, True, False), (2, 2, False, False), (4099, 3, True, False), ((1 << 20) + 9, 2, True, True)])
def test_logup_grand_sum_in_one_call(oracle, n, count, with_table, registered):
    """h2_logup_grand_sum (one grand-sum column of a logup lookup: beta + f per input, the inversions, the table's m / (beta + t)
    and the running sum on the device; plonk/logup/prover.rs:243-347) against the oracle's steps taken one by one"""
    import ctypes

    import halo2_gpu_specific_amd as h2
    import halo2_gpu_specific_amd.arithmetic as ar
    from oracle_prover import OracleLib

    L, O = h2.lib(), OracleLib().O
    inputs = [oracle.random_fr(9930 + j, n) for j in range(count)]
    table, m = oracle.random_fr(9940, n), oracle.random_fr(9941, n)
    beta, init = oracle.random_fr(9942, 2)
    if n > 2:
        inputs[0][1] = oracle.eval_op(ar.OP_SUB, np.zeros((1, 4), dtype=np.uint64), beta.reshape(1, 4), 0, 0, None)[0]   # beta + f = 0: kept 0
    acc = np.zeros((n, 4), dtype=np.uint64)
    for f in inputs:
        t = oracle.eval_op(ar.OP_SUM_C, f, None, 0, 0, beta)
        oracle.lib.oracle_batch_invert(_ptr(t), n)
        acc = oracle.eval_op(ar.OP_SUM, acc, t, 0, 0, None)
    if with_table:
        t = oracle.eval_op(ar.OP_SUM_C, table, None, 0, 0, beta)
        oracle.lib.oracle_batch_invert(_ptr(t), n)
        acc = oracle.eval_op(ar.OP_SUB, acc, oracle.eval_op(ar.OP_MUL, t, m, 0, 0, None), 0, 0, None)
    want = np.zeros((n, 4), dtype=np.uint64)
    want[0] = init
    if n > 1:
        O.oracle_prefix_sum(_ptr(acc), n, _ptr(init), _ptr(want))
    ip = (ctypes.c_void_p * count)(*[f.ctypes.data for f in inputs])
    if registered:
        assert L.h2_poly_register(_ptr(table), n) == 0 and L.h2_poly_register(_ptr(inputs[0]), n) == 0
    try:
        for _ in range(2 if registered else 1):
            z = np.empty((n, 4), dtype=np.uint64)
            assert L.h2_logup_grand_sum(_ptr(z), ip, count, _ptr(table) if with_table else None, _ptr(m) if with_table else None, n,
                                        _ptr(beta), _ptr(init)) == 0
            assert np.array_equal(z, want)
    finally:
        if registered:
            assert L.h2_poly_unregister(_ptr(table)) == 0 and L.h2_poly_unregister(_ptr(inputs[0])) == 0
    assert L.h2_logup_grand_sum(_ptr(z), ip, count, _ptr(table), None, n, _ptr(beta), _ptr(init)) != 0        # a table without its m


@pytest.mark.parametrize("k", [3, 12, 20])
def test_intt_out_of_place(oracle, k):
    """h2_intt_to: the values are left alone, the coefficients land in another vector -- the same coefficients h2_intt leaves in place"""
    import halo2_gpu_specific_amd as h2
    from halo2_gpu_specific_amd import prover

    L = h2.lib()
    n = 1 << k
    dom = prover.Domain(k, 3)
    wi, dv = prover._fr(dom.omega_inv), prover._fr(dom.ifft_divisor)
    a = oracle.random_fr(9990 + k, n)
    keep, out = a.copy(), np.empty((n, 4), dtype=np.uint64)
    assert L.h2_intt_to(_ptr(a), _ptr(out), wi, dv, k) == 0
    assert np.array_equal(a, keep)
    assert L.h2_intt(_ptr(a), wi, dv, k) == 0
    assert np.array_equal(out, a)
    assert L.h2_poly_register(_ptr(keep), n) == 0            # a registered column is read on the device
    try:
        out2 = np.empty((n, 4), dtype=np.uint64)
        assert L.h2_intt_to(_ptr(keep), _ptr(out2), wi, dv, k) == 0 and np.array_equal(out2, a)
    finally:
        assert L.h2_poly_unregister(_ptr(keep)) == 0
