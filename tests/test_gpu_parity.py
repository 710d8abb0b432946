"""GPU parity: the HIP path (through the C ABI, libhalo2_hip.so) against the CPU oracle and
the committed golden vectors.  Bit-exact: everything here is integer / finite-field work.
MSM results are compared after affine normalisation (the reference's own `==` on C::Curve is
projective-aware, poly/commitment.rs:494)."""
import ctypes
import os

import numpy as np
import pytest

import halo2_gpu_specific_amd as h2
from halo2_gpu_specific_amd import arithmetic as ar
from h2util import (
    Q_MOD,
    R_MOD,
    arr_to_points,
    from_mont,
    fr_mont,
    golden_points,
    h2i,
    ints_to_arr,
    load_golden,
    points_to_arr,
    to_mont,
)

pytestmark = pytest.mark.gpu

S = 28
ROOT = 0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C


def omega_for(log_n):
    return pow(ROOT, 1 << (S - log_n), R_MOD)


def test_device_visible():
    assert h2.lib().h2_device_count() >= 1


# ------------------------------------------------------------------ Montgomery conversion
def test_batch_mont_unmont():
    vals = [0, 1, 2, R_MOD - 1, R_MOD - 2, (1 << 253) + 12345, 0xDEADBEEF << 100] + [
        (i * 0x9E3779B97F4A7C15F39CC0605CEDC834) % R_MOD for i in range(1, 300)
    ]
    a = ints_to_arr(vals)
    ar.gpu_mont(a)
    assert from_mont(a) == vals
    assert np.array_equal(a, to_mont(vals))
    ar.gpu_unmont(a)
    assert np.array_equal(a, ints_to_arr(vals))


def test_conversions_and_msm_scalars_accept_any_256_bit_value(oracle):
    """the generated multiplier's carry analysis assumes operands below 2^254, so the entry points that take RAW
    caller data -- h2_batch_mont / h2_batch_unmont and the scalars of an MSM -- go through the wide-operand product, which
    is exact for any 256-bit first operand: a non-canonical input gives its residue, not a silently wrong value"""
    vals = [R_MOD, R_MOD + 1, 2 * R_MOD - 1, 4 * R_MOD + 7, (1 << 256) - 1, (1 << 255) + 12345, (1 << 254), 5 * R_MOD + 3] + [
        ((i * 0x9E3779B97F4A7C15F39CC0605CEDC834A1B2C3D4E5F60718) << 40) % (1 << 256) for i in range(1, 200)]
    a = ints_to_arr(vals)
    ar.gpu_mont(a)
    assert from_mont(a) == [v % R_MOD for v in vals]
    b = ints_to_arr(vals)
    ar.gpu_unmont(b)
    rinv = pow(1 << 256, -1, R_MOD)
    assert [limb_int for limb_int in __import__("h2util").arr_to_ints(b)] == [v * rinv % R_MOD for v in vals]
    # MSM with scalars whose Montgomery images are not canonical: s_i and s_i + r (as memory images) are the same scalar
    n = 1 << 12
    pts = oracle.random_g1(4242, n)
    s = oracle.random_fr(4243, n)
    want = _affine(oracle, oracle.best_multiexp(s, pts))
    ints = __import__("h2util").arr_to_ints(s)
    bumped = ints_to_arr([v + R_MOD * (1 + i % 4) if v + R_MOD * (1 + i % 4) < (1 << 256) else v for i, v in enumerate(ints)])
    assert _affine(oracle, ar.gpu_multiexp_single_gpu_with_bound(bumped, pts, 254)) == want


# ------------------------------------------------------------------ elementwise
@pytest.mark.parametrize("size", [1, 7, 256, 1000, 1 << 14])
def test_eval_ops(oracle, size):
    l = oracle.random_fr(11, size)
    r = oracle.random_fr(12, size)
    c = oracle.random_fr(13, 1)[0]
    rots = [(0, 0), (1, -1), (-3, 5), (size - 1, -(size - 1))]
    for op in range(9):
        for l_rot, r_rot in rots:
            l_rot, r_rot = (l_rot % size if l_rot > 0 else -((-l_rot) % size)), (r_rot % size if r_rot > 0 else -((-r_rot) % size))
            want = oracle.eval_op(op, l, r, l_rot, r_rot, c)
            need_r = op in (ar.OP_SUM, ar.OP_MUL, ar.OP_SUB, ar.OP_LCTHETA, ar.OP_LCBETA)
            got = ar.eval_op(op, l if op != ar.OP_CONSTANT else None, r if need_r else None, l_rot, r_rot, c, size=size)
            assert np.array_equal(got, want), (op, l_rot, r_rot)


def test_eval_op_edge_values(oracle):
    vals = [0, 1, R_MOD - 1, R_MOD - 2, 2, (R_MOD + 1) // 2]
    l = to_mont(vals)
    r = to_mont(list(reversed(vals)))
    for op in (ar.OP_SUM, ar.OP_SUB, ar.OP_MUL):
        assert np.array_equal(ar.eval_op(op, l, r), oracle.eval_op(op, l, r, 0, 0, None))


def test_divide_by_vanishing(oracle):
    d, t = oracle.domain(5, 6)
    a = oracle.random_fr(21, 1 << d.extended_k)
    want = a.copy()
    oracle.lib.oracle_divide_by_vanishing_poly(want.ctypes.data, len(want), t.ctypes.data, len(t), 4)
    got = ar.divide_by_vanishing_poly(a.copy(), t)
    assert np.array_equal(got, want)


# ------------------------------------------------------------------ NTT
def test_ntt_golden():
    for case in load_golden("ntt_kat.json"):
        log_n = case["log_n"]
        x = to_mont([h2i(v) for v in case["input"]])
        out = ar.best_fft(x.copy(), fr_mont(h2i(case["omega"])), log_n)
        assert from_mont(out) == [h2i(v) for v in case["output"]], log_n
        back = ar.gpu_ifft(out, fr_mont(h2i(case["omega_inv"])), log_n, fr_mont(h2i(case["n_inv"])))
        assert np.array_equal(back, x), log_n


@pytest.mark.parametrize("log_n", list(range(0, 21)))
def test_ntt_vs_oracle(oracle, log_n):
    n = 1 << log_n
    x = oracle.random_fr(100 + log_n, n)
    w = fr_mont(omega_for(log_n))
    want = oracle.best_fft(x, w, log_n)
    got = ar.best_fft(x.copy(), w, log_n)
    assert np.array_equal(got, want)


def test_ntt_edge_inputs(oracle):
    log_n = 10
    n = 1 << log_n
    w = fr_mont(omega_for(log_n))
    zeros = np.zeros((n, 4), dtype=np.uint64)
    assert not ar.best_fft(zeros.copy(), w, log_n).any()
    delta = zeros.copy()
    delta[0] = fr_mont(1)
    out = ar.best_fft(delta, w, log_n)  # DFT of a delta is all ones
    assert np.array_equal(out, np.tile(fr_mont(1), (n, 1)))
    big = to_mont([R_MOD - 1] * n)
    assert np.array_equal(ar.best_fft(big.copy(), w, log_n), oracle.best_fft(big, w, log_n))


def test_ntt_linearity(oracle):
    log_n = 16
    n = 1 << log_n
    w = fr_mont(omega_for(log_n))
    a, b = oracle.random_fr(1, n), oracle.random_fr(2, n)
    c = oracle.random_fr(3, 1)[0]
    fa, fb = ar.best_fft(a.copy(), w, log_n), ar.best_fft(b.copy(), w, log_n)
    comb = oracle.eval_op(ar.OP_LCTHETA, a, b, 0, 0, c)  # a*c + b
    want = oracle.eval_op(ar.OP_LCTHETA, fa, fb, 0, 0, c)
    assert np.array_equal(ar.best_fft(comb, w, log_n), want)


@pytest.mark.parametrize("log_n", [22, 24])
def test_ntt_full_size(oracle, log_n):
    """BASELINE config 3: 2^24 forward + inverse; forward output == oracle on all elements."""
    n = 1 << log_n
    x = oracle.random_fr(0x48414C4F32 + 1, n)
    w = fr_mont(omega_for(log_n))
    want = oracle.best_fft(x, w, log_n)
    got = ar.best_fft(x.copy(), w, log_n)
    assert np.array_equal(got, want)
    back = ar.gpu_ifft(got, fr_mont(pow(omega_for(log_n), -1, R_MOD)), log_n, fr_mont(pow(n, -1, R_MOD)))
    assert np.array_equal(back, x)


def test_ntt_2p25_nine_bit_pass(oracle):
    """2^25 -- the extended domain of a k = 24 proof -- runs as passes of 8 + 8 + 9 bits: sampled outputs against
    the oracle's Horner evaluation X[k] = sum_j x_j w^(jk) (arithmetic.rs:714-735 restated), and the round trip"""
    log_n = 25
    n = 1 << log_n
    x = oracle.random_fr(0x2525, n)
    omega = omega_for(log_n)
    got = ar.best_fft(x.copy(), fr_mont(omega), log_n)
    for k in (0, 1, 255, 256, 65535, 65536 + 257, (1 << 24) + 12345, n - 1):
        point = fr_mont(pow(omega, k, R_MOD))
        want = np.zeros(4, dtype=np.uint64)
        oracle.lib.oracle_eval_polynomial(x.ctypes.data, n, point.ctypes.data, want.ctypes.data)
        assert np.array_equal(got[k], want), k
    back = ar.gpu_ifft(got, fr_mont(pow(omega, -1, R_MOD)), log_n, fr_mont(pow(n, -1, R_MOD)))
    assert np.array_equal(back, x)


def test_coset_golden(oracle):
    for case in load_golden("coset_kat.json"):
        d, _ = oracle.domain(case["j"], case["k"])
        coeffs = to_mont([h2i(v) for v in case["coeffs"]])
        ext = ar.coeff_to_extended(coeffs, d.k, d.extended_k, d.fr("g_coset"), d.fr("g_coset_inv"), d.fr("extended_omega"))
        assert from_mont(ext) == [h2i(v) for v in case["extended"]]
        back = ar.extended_to_coeff(
            ext, d.k, d.extended_k, d.quotient_poly_degree, d.fr("g_coset"), d.fr("g_coset_inv"), d.fr("extended_omega_inv"),
            d.fr("extended_ifft_divisor"),
        )
        assert np.array_equal(back[: 1 << d.k], coeffs) and not back[1 << d.k :].any()


@pytest.mark.parametrize("j,k", [(3, 10), (5, 12), (4, 15), (2, 9), (9, 11)])
def test_coset_vs_oracle(oracle, j, k):
    d, _ = oracle.domain(j, k)
    coeffs = oracle.random_fr(j * 100 + k, 1 << k)
    want = oracle.coeff_to_extended(coeffs, d)
    got = ar.coeff_to_extended(coeffs, d.k, d.extended_k, d.fr("g_coset"), d.fr("g_coset_inv"), d.fr("extended_omega"))
    assert np.array_equal(got, want)
    ext = oracle.random_fr(j * 1000 + k, 1 << d.extended_k)  # arbitrary extended-domain values
    want_c = oracle.extended_to_coeff(ext, d)
    got_c = ar.extended_to_coeff(
        ext, d.k, d.extended_k, d.quotient_poly_degree, d.fr("g_coset"), d.fr("g_coset_inv"), d.fr("extended_omega_inv"),
        d.fr("extended_ifft_divisor"),
    )
    assert np.array_equal(got_c, want_c)


# ------------------------------------------------------------------ MSM
def _affine(oracle, jac):
    return arr_to_points(oracle.to_affine(jac))[0]


def test_msm_golden(oracle):
    for case in load_golden("msm_kat.json"):
        scalars = to_mont([h2i(s) for s in case["scalars"]]).reshape(-1, 4)
        pts = points_to_arr(golden_points(case["points"])).reshape(-1, 8)
        want = golden_points([case["result"]])[0]
        got = _affine(oracle, ar.gpu_multiexp_single_gpu_with_bound(scalars, pts, 254))
        assert got == want, case["name"]
        got = _affine(oracle, ar.gpu_multiexp_bound(scalars, pts, 254))
        assert got == want, case["name"] + " (multi)"


@pytest.mark.parametrize("log_n", [0, 1, 5, 10, 14, 16])
def test_msm_vs_oracle(oracle, log_n):
    n = 1 << log_n
    scalars = oracle.random_fr(500 + log_n, n)
    pts = oracle.random_g1(600 + log_n, n)
    want = _affine(oracle, oracle.best_multiexp(scalars, pts))
    assert _affine(oracle, ar.best_multiexp(scalars, pts)) == want


def test_msm_ragged_sizes(oracle):
    for n in (3, 33, 1000, 4097, 70001):
        scalars = oracle.random_fr(n, n)
        pts = oracle.random_g1(n + 1, n)
        want = _affine(oracle, oracle.best_multiexp(scalars, pts))
        assert _affine(oracle, ar.gpu_multiexp_single_gpu_with_bound(scalars, pts, 254)) == want, n


def test_msm_max_bits_bound(oracle):
    """advice-column shape: all scalars < 2^16, max_bits = 16 (plonk/prover.rs:286,296-297)"""
    n = 1 << 14
    vals = [(i * 2654435761) % (1 << 16) for i in range(n)]
    scalars = to_mont(vals)
    pts = oracle.random_g1(77, n)
    want = _affine(oracle, oracle.best_multiexp(scalars, pts))
    for bits in (16, 17, 64, 254):
        assert _affine(oracle, ar.gpu_multiexp_single_gpu_with_bound(scalars, pts, bits)) == want, bits
    ident = ar.gpu_multiexp_single_gpu_with_bound(scalars, pts, 0)  # arithmetic.rs:346
    assert _affine(oracle, ident) == (0, 0)


def test_msm_skewed_and_adversarial(oracle):
    n = 1 << 13
    pts = oracle.random_g1(88, n)
    cases = {
        "boolean": [i & 1 for i in range(n)],
        "all_ones": [1] * n,
        "all_same": [0x1234567] * n,
        "zero_one_rm1": [(0, 1, R_MOD - 1)[i % 3] for i in range(n)],
        "half_zero": [0 if i % 2 else (i * 0x9E3779B97F4A7C15F39CC0605CEDC835) % R_MOD for i in range(n)],
    }
    for name, vals in cases.items():
        scalars = to_mont(vals)
        want = _affine(oracle, oracle.best_multiexp(scalars, pts))
        assert _affine(oracle, ar.best_multiexp(scalars, pts)) == want, name
    # repeated points and P / -P pairs
    rep = np.tile(pts[:4], (n // 4, 1))
    neg = rep.copy()
    ys = [(-y) % Q_MOD for _, y in arr_to_points(rep[:4])]
    neg_y = to_mont(ys, Q_MOD)
    neg[1::2, 4:] = np.tile(neg_y[1::2], (n // 4, 1))[: len(neg[1::2])]
    scalars = oracle.random_fr(99, n)
    for name, b in (("repeated", rep), ("neg_pairs", neg)):
        want = _affine(oracle, oracle.best_multiexp(scalars, b))
        assert _affine(oracle, ar.best_multiexp(scalars, b)) == want, name
    same = to_mont([5] * n)
    assert _affine(oracle, ar.best_multiexp(same, neg)) == _affine(oracle, oracle.best_multiexp(same, neg))


def test_msm_linearity(oracle):
    """size-independent property: MSM(a + b, P) == MSM(a, P) + MSM(b, P)"""
    n = 1 << 15
    pts = oracle.random_g1(5, n)
    a, b = oracle.random_fr(6, n), oracle.random_fr(7, n)
    ab = oracle.eval_op(ar.OP_SUM, a, b, 0, 0, None)
    ra, rb, rab = ar.best_multiexp(a, pts), ar.best_multiexp(b, pts), ar.best_multiexp(ab, pts)
    s = np.zeros(12, dtype=np.uint64)
    oracle.lib.oracle_g1_add(ra.ctypes.data, rb.ctypes.data, s.ctypes.data)
    assert _affine(oracle, s) == _affine(oracle, rab)


def test_msm_2_20(oracle):
    """BASELINE config 2: 2^20 random scalars / points, bit-exact vs the oracle"""
    n = 1 << 20
    scalars = oracle.random_fr(0x48414C4F32, n)
    pts = oracle.random_g1(0x48414C4F32, n)
    want = _affine(oracle, oracle.best_multiexp(scalars, pts))
    assert _affine(oracle, ar.gpu_multiexp_single_gpu_with_bound(scalars, pts, 254)) == want


def test_commit_lagrange_and_ifft(oracle):
    """gpu_multiexp_bound_and_fft (arithmetic.rs:375-410) + the relational KAT of
    poly/commitment.rs:481-495: commit(ifft(a)) == commit_lagrange(a)."""
    k = 8
    n = 1 << k
    case_s = fr_mont(0x1234567890ABCDEF1234567890ABCDEF)
    g = np.zeros((n, 8), dtype=np.uint64)
    gl = np.zeros((n, 8), dtype=np.uint64)
    oracle.lib.oracle_unsafe_setup(k, case_s.ctypes.data, g.ctypes.data, gl.ctypes.data)
    d, _ = oracle.domain(1, k)
    a = to_mont(list(range(n)))  # commitment.rs:489-491
    vals = a.copy()
    c_lagrange = ar.gpu_multiexp_bound_and_fft(vals, gl, 254, d.fr("omega_inv"), d.fr("ifft_divisor"), k)
    assert np.array_equal(vals, oracle.ifft(a, d.fr("omega_inv"), k, d.fr("ifft_divisor")))
    c_coeff = ar.best_multiexp(vals, g)
    assert _affine(oracle, c_lagrange) == _affine(oracle, c_coeff)
    assert _affine(oracle, c_lagrange) == _affine(oracle, oracle.best_multiexp(a, gl))


@pytest.mark.parametrize("k", [16, 20])
@pytest.mark.timeout(600)
def test_commit_lagrange_and_ifft_large(oracle, k):
    """h2_msm_intt at the sizes the prover calls it with (one z-polynomial at 2^16 / 2^20): the in-place iNTT equals the
    oracle's `ifft` on every element and the commitment equals the oracle's MSM over random bases; a bounded column
    (max_bits = 40) goes through the same entry point"""
    n = 1 << k
    d, _ = oracle.domain(1, k)
    pts = oracle.random_g1(600 + k, n)
    for bits in (254, 40):
        a = oracle.random_fr(610 + k + bits, n)
        if bits < 254:                                   # canonical values below 2^40, stored Montgomery
            a = to_mont_array(oracle, a, bits)
        vals = a.copy()
        got = ar.gpu_multiexp_bound_and_fft(vals, pts, 254, d.fr("omega_inv"), d.fr("ifft_divisor"), k)
        assert np.array_equal(vals, oracle.ifft(a, d.fr("omega_inv"), k, d.fr("ifft_divisor")))
        assert _affine(oracle, got) == _affine(oracle, oracle.best_multiexp(a, pts))


def to_mont_array(oracle, a, bits):
    """(n, 4) u64 random limbs -> Montgomery form of (value mod 2^bits)"""
    small = a.copy()
    full, rem = divmod(bits, 64)
    small[:, full + (1 if rem else 0):] = 0
    if rem:
        small[:, full] &= np.uint64((1 << rem) - 1)
    return ar.gpu_mont(small)


@pytest.mark.parametrize("j,k", [(3, 22), (5, 22)])
@pytest.mark.timeout(900)
def test_coset_divide_inverse_vs_oracle_large(oracle, j, k):
    """coeff_to_extended, divide_by_vanishing_poly and extended_to_coeff against the oracle at the prover's sizes:
    2^22 rows extended to 2^23 (degree 3, BASELINE configs[3]) / 2^24 (degree 5); every element"""
    d, t = oracle.domain(j, k)
    coeffs = oracle.random_fr(j * 100 + k, 1 << k)
    ext = ar.coeff_to_extended(coeffs, d.k, d.extended_k, d.fr("g_coset"), d.fr("g_coset_inv"), d.fr("extended_omega"))
    want = oracle.coeff_to_extended(coeffs, d)
    assert np.array_equal(ext, want)
    del want
    div = ar.divide_by_vanishing_poly(ext.copy(), t)
    oracle.lib.oracle_divide_by_vanishing_poly(ext.ctypes.data, len(ext), t.ctypes.data, len(t), 64)
    assert np.array_equal(div, ext)
    got_c = ar.extended_to_coeff(
        div, d.k, d.extended_k, d.quotient_poly_degree, d.fr("g_coset"), d.fr("g_coset_inv"), d.fr("extended_omega_inv"),
        d.fr("extended_ifft_divisor"),
    )
    assert np.array_equal(got_c, oracle.extended_to_coeff(ext, d))


def test_msm_batch_shared_bases(oracle):
    """h2_dev_msm_batch: several columns committed against the same bases (plonk/prover.rs:293-299),
    pipelined on two streams; each result must equal the oracle's MSM of that column."""
    import ctypes

    import torch

    L = h2.lib()
    n, count = 1 << 12, 5
    pts = oracle.random_g1(901, n)
    cols = [oracle.random_fr(910 + i, n) for i in range(count)]
    cols[2][: n // 2] = 0  # a half-empty column
    dev = torch.device("cuda", 0)
    d_pts = torch.from_numpy(pts.view(np.int64)).to(dev)
    d_cols = [torch.from_numpy(c.view(np.int64)).to(dev) for c in cols]
    per = (L.h2_msm_scratch_bytes(n, 254) + 255) // 256 * 256
    scratch = torch.empty(2 * per, dtype=torch.uint8, device=dev)
    ptrs = (ctypes.c_void_p * count)(*[t.data_ptr() for t in d_cols])
    out = np.zeros((count, 12), dtype=np.uint64)
    rc = L.h2_dev_msm_batch(ptrs, count, d_pts.data_ptr(), n, 254, scratch.data_ptr(), 2 * per, out.ctypes.data, None)
    assert rc == 0, L.h2_last_error()
    for i in range(count):
        assert _affine(oracle, out[i]) == _affine(oracle, oracle.best_multiexp(cols[i], pts)), i
    # too-small scratch is refused, not overrun
    assert L.h2_dev_msm_batch(ptrs, count, d_pts.data_ptr(), n, 254, scratch.data_ptr(), per, out.ctypes.data, None) == 1


def test_resident_bases(oracle):
    """h2_bases_register: MSMs over sub-ranges of a registered SRS reuse one device copy"""
    L = h2.lib()
    n = 1 << 16
    pts = oracle.random_g1(77, n)
    s = oracle.random_fr(78, n)
    want_full = _affine(oracle, oracle.best_multiexp(s, pts))
    want_half = _affine(oracle, oracle.best_multiexp(s[: n // 2], pts[n // 4 : n // 4 + n // 2]))
    assert L.h2_bases_register(pts.ctypes.data, n) == 0
    try:
        for _ in range(2):
            assert _affine(oracle, ar.gpu_multiexp_single_gpu_with_bound(s, pts, 254)) == want_full
            sub = pts[n // 4 : n // 4 + n // 2]  # a view into the registered range (commit with size < n)
            assert sub.ctypes.data == pts.ctypes.data + 64 * (n // 4)
            out = np.zeros(12, dtype=np.uint64)
            assert L.h2_msm(s.ctypes.data, sub.ctypes.data, n // 2, 254, out.ctypes.data) == 0
            assert _affine(oracle, out) == want_half
    finally:
        assert L.h2_bases_unregister(pts.ctypes.data) == 0
    assert _affine(oracle, ar.gpu_multiexp_single_gpu_with_bound(s, pts, 254)) == want_full  # back to per-call upload


def test_resident_bases_reregistered_address_is_reuploaded(oracle):
    """unregister -> refill the same host buffer -> register again (also with a larger length) must use the
    new points, not the device copy of the old registration"""
    L = h2.lib()
    n = 1 << 12
    buf = np.zeros((2 * n, 8), dtype=np.uint64)
    s = oracle.random_fr(178, 2 * n)
    first, second = oracle.random_g1(177, n), oracle.random_g1(179, 2 * n)
    buf[:n] = first
    assert L.h2_bases_register(buf.ctypes.data, n) == 0
    out = np.zeros(12, dtype=np.uint64)
    assert L.h2_msm(s.ctypes.data, buf.ctypes.data, n, 254, out.ctypes.data) == 0
    assert _affine(oracle, out) == _affine(oracle, oracle.best_multiexp(s[:n], first))
    assert L.h2_bases_unregister(buf.ctypes.data) == 0
    buf[:] = second                                            # same address, new (and longer) contents
    assert L.h2_bases_register(buf.ctypes.data, 2 * n) == 0
    try:
        assert L.h2_msm(s.ctypes.data, buf.ctypes.data, 2 * n, 254, out.ctypes.data) == 0
        assert _affine(oracle, out) == _affine(oracle, oracle.best_multiexp(s, second))
        # re-registering WITHOUT an unregister in between also replaces the copy
        buf[:n] = first
        assert L.h2_bases_register(buf.ctypes.data, n) == 0
        assert L.h2_msm(s.ctypes.data, buf.ctypes.data, n, 254, out.ctypes.data) == 0
        assert _affine(oracle, out) == _affine(oracle, oracle.best_multiexp(s[:n], first))
    finally:
        assert L.h2_bases_unregister(buf.ctypes.data) == 0


def test_concurrent_callers(oracle):
    """Threading contract: every entry point is called concurrently from rayon workers;
    the blocking device pool serialises them (arithmetic.rs:314-331).  8 threads mix NTTs and MSMs."""
    import threading

    log_n = 12
    n = 1 << log_n
    w = fr_mont(omega_for(log_n))
    xs = [oracle.random_fr(700 + i, n) for i in range(8)]
    pts = oracle.random_g1(800, n)
    want_ntt = [oracle.best_fft(x, w, log_n) for x in xs]
    want_msm = [_affine(oracle, oracle.best_multiexp(x, pts)) for x in xs]
    errors = []

    def worker(i):
        try:
            for _ in range(3):
                if i % 2 == 0:
                    assert np.array_equal(ar.best_fft(xs[i].copy(), w, log_n), want_ntt[i])
                else:
                    assert _affine(oracle, ar.best_multiexp(xs[i], pts)) == want_msm[i]
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


@pytest.mark.parametrize("n", [64, 1000, 1 << 13, 70001])
def test_msm_dominant_scalar(oracle, n):
    """columns that are constant over most rows (grand products over padding rows): the dominant value gets its own
    window -- one addition per row and one host scalar multiplication -- and the result is the same group element"""
    import random

    rnd = random.Random(n)
    pts = oracle.random_g1(123, n)
    rand = from_mont(oracle.random_fr(321, n))
    hot = rnd.randrange(R_MOD)
    cases = {
        "7/8 hot": [hot if i >= n // 8 else rand[i] for i in range(n)],
        "all hot": [hot] * n,
        "hot = r - 1": [R_MOD - 1 if i % 4 else rand[i] for i in range(n)],
        "hot = 1, rest zero": [1 if i % 3 else 0 for i in range(n)],
        "two frequent values": [(hot, 77, rand[i])[i % 3] for i in range(n)],
        "30 % hot, interleaved": [hot if (i * 7) % 10 < 3 else rand[i] for i in range(n)],
        "hot small, 16-bit bound": [5 if i % 8 else (i * 31) % 65536 for i in range(n)],
    }
    for name, vals in cases.items():
        scalars = to_mont(vals)
        want = _affine(oracle, oracle.best_multiexp(scalars, pts))
        bits = 16 if "16-bit" in name else 254
        assert _affine(oracle, ar.gpu_multiexp_single_gpu_with_bound(scalars, pts, bits)) == want, name
        assert _affine(oracle, ar.gpu_multiexp_bound(scalars, pts, bits)) == want, name + " (multi)"


def test_msm_batch_mixes_dominant_and_uniform_columns(oracle):
    import torch

    L = h2.lib()
    n = 1 << 12
    pts = oracle.random_g1(55, n)
    rand = [from_mont(oracle.random_fr(60 + j, n)) for j in range(4)]
    cols = [
        to_mont(rand[0]),
        to_mont([0xABCDEF0123456789 if i > 100 else rand[1][i] for i in range(n)]),
        to_mont(rand[2]),
        to_mont([3] * n),
        to_mont([rand[3][0] if i % 2 else rand[3][i] for i in range(n)]),
    ]
    d_pts = torch.from_numpy(pts.view(np.int64)).cuda()
    d_cols = [torch.from_numpy(c.view(np.int64)).cuda() for c in cols]
    per = (L.h2_msm_scratch_bytes(n, 254) + 255) // 256 * 256
    scratch = torch.empty(2 * per, dtype=torch.uint8, device="cuda")
    ptrs = (ctypes.c_void_p * len(cols))(*[t.data_ptr() for t in d_cols])
    out = np.zeros((len(cols), 12), dtype=np.uint64)
    assert L.h2_dev_msm_batch(ptrs, len(cols), d_pts.data_ptr(), n, 254, scratch.data_ptr(), 2 * per, out.ctypes.data, None) == 0
    for j, c in enumerate(cols):
        assert _affine(oracle, out[j]) == _affine(oracle, oracle.best_multiexp(c, pts)), j


def test_msm_batch_ex_per_column_bases_and_bounds(oracle):
    """h2_dev_msm_batch_ex: one pipelined call over columns with their own base table and max_bits (16-bit witness
    columns next to full-width ones, g_lagrange next to g); a zero bound yields the identity"""
    import torch

    L = h2.lib()
    n = 1 << 12
    tables = [oracle.random_g1(91, n), oracle.random_g1(92, n)]
    small = to_mont([(i * 2654435761) % 65536 for i in range(n)])
    cols = [oracle.random_fr(93, n), small, oracle.random_fr(94, n), small, oracle.random_fr(95, n)]
    which = [0, 0, 1, 1, 0]
    bits = [254, 16, 254, 17, 0]
    d_tab = [torch.from_numpy(t.view(np.int64)).cuda() for t in tables]
    d_cols = [torch.from_numpy(np.ascontiguousarray(c).view(np.int64)).cuda() for c in cols]
    per = max((L.h2_msm_scratch_bytes(n, b) + 255) // 256 * 256 for b in bits)
    scratch = torch.empty(2 * per, dtype=torch.uint8, device="cuda")
    count = len(cols)
    sp = (ctypes.c_void_p * count)(*[t.data_ptr() for t in d_cols])
    bp = (ctypes.c_void_p * count)(*[d_tab[w].data_ptr() for w in which])
    bb = (ctypes.c_uint32 * count)(*bits)
    out = np.zeros((count, 12), dtype=np.uint64)
    assert L.h2_dev_msm_batch_ex(sp, bp, bb, count, n, scratch.data_ptr(), 2 * per, out.ctypes.data, None) == 0, L.h2_last_error()
    for j in range(count):
        if bits[j] == 0:
            assert _affine(oracle, out[j]) == (0, 0)
        else:
            assert _affine(oracle, out[j]) == _affine(oracle, oracle.best_multiexp(cols[j], tables[which[j]])), j
    assert L.h2_dev_msm_batch_ex(sp, bp, bb, count, n, scratch.data_ptr(), per, out.ctypes.data, None) == 1


def test_msm_2p20_proof_shaped_columns(oracle):
    """BASELINE config 2's size on the scalar distributions a proof commits: a grand-product column (constant over
    7/8 of the rows: dominant-scalar window + skew path) and a sparse witness column (a few small values, 16-bit
    blinding rows: the empty-bucket bisection), against the oracle"""
    n = 1 << 20
    pts = oracle.random_g1(2020, n)
    z = oracle.random_fr(2021, n)
    z[n // 8:] = z[5]
    want = _affine(oracle, oracle.best_multiexp(z, pts))
    assert _affine(oracle, ar.gpu_multiexp_single_gpu_with_bound(z, pts, 254)) == want
    w = np.zeros((n, 4), dtype=np.uint64)
    w[: n // 8, 0] = np.tile(np.array([5, 25, 30, 5], dtype=np.uint64), n // 32)
    w[n - 6:, 0] = [60000, 12345, 3, 40000, 5, 65535]
    w = to_mont([int(v) for v in w[:, 0]])
    want = _affine(oracle, oracle.best_multiexp(w, pts))
    for bits in (16, 254):
        assert _affine(oracle, ar.gpu_multiexp_single_gpu_with_bound(w, pts, bits)) == want, bits


@pytest.mark.parametrize("which", ["zero_one_rm1", "half_zero", "repeated", "neg_pairs"])
def test_msm_2p20_adversarial(oracle, which):
    """BASELINE config 2's extra distributions AT ITS STATED SIZE ("each also at 2^20"; the 2^13 forms are
    test_msm_skewed_and_adversarial): scalars in {0, 1, r - 1}, every other scalar zero (commitment.rs:207-212), four points
    repeated 2^18 times, and P / -P pairs -- every bucket of a window receives the same few points, the doubling and the
    P + (-P) branches of the accumulation are the common case, not the exception"""
    n = 1 << 20
    pts = oracle.random_g1(2030, n)
    if which == "zero_one_rm1":
        three = np.concatenate([fr_mont(0).reshape(1, 4), fr_mont(1).reshape(1, 4), fr_mont(R_MOD - 1).reshape(1, 4)])
        scalars = np.ascontiguousarray(np.tile(three, ((n + 2) // 3, 1))[:n])
    else:
        scalars = oracle.random_fr(2031, n)
    if which == "half_zero":
        scalars[1::2] = 0
    if which == "repeated":
        pts = np.ascontiguousarray(np.tile(pts[:4], (n // 4, 1)))
    if which == "neg_pairs":                                   # P0, -P0, P1, -P1, P0, ...
        four = np.repeat(pts[:2], 2, axis=0)
        four[1::2, 4:] = to_mont([(-y) % Q_MOD for _, y in arr_to_points(pts[:2])], Q_MOD)
        pts = np.ascontiguousarray(np.tile(four, (n // 4, 1)))
    want = _affine(oracle, oracle.best_multiexp(scalars, pts))
    assert _affine(oracle, ar.gpu_multiexp_single_gpu_with_bound(scalars, pts, 254)) == want
    if which == "neg_pairs":
        same = np.ascontiguousarray(np.tile(fr_mont(5).reshape(1, 4), (n, 1)))   # sum of 5 P - 5 P pairs: the identity
        assert _affine(oracle, ar.best_multiexp(same, pts)) == _affine(oracle, oracle.best_multiexp(same, pts)) == (0, 0)


@pytest.mark.parametrize("n", [1 << 8, 1 << 12, 5000])
def test_msm_fused_group(oracle, n):
    """h2_dev_msm_batch with h2_msm_batch_scratch_bytes of scratch: the columns become the windows of ONE wide MSM
    (uniform, dominant-scalar, all-equal, zero and sparse columns side by side), same points as the oracle; with the
    pipeline-sized scratch the same call takes the two-stream path and returns the same points"""
    import torch

    L = h2.lib()
    pts = oracle.random_g1(55, n)
    rand = [from_mont(oracle.random_fr(160 + j, n)) for j in range(3)]
    cols = [
        to_mont(rand[0]),
        to_mont([0xABCDEF0123456789ABCDEF if i > n // 10 else rand[1][i] for i in range(n)]),
        to_mont([R_MOD - 1] * n),
        to_mont([0] * n),
        to_mont(rand[2]),
        to_mont([7 if i % 5 == 0 else 0 for i in range(n)]),
        to_mont([rand[2][3] if i % 2 else rand[2][i] for i in range(n)]),
    ]
    d_pts = torch.from_numpy(pts.view(np.int64)).cuda()
    d_cols = [torch.from_numpy(c.view(np.int64)).cuda() for c in cols]
    count = len(cols)
    ptrs = (ctypes.c_void_p * count)(*[t.data_ptr() for t in d_cols])
    want = [_affine(oracle, oracle.best_multiexp(c, pts)) for c in cols]
    fused_bytes = L.h2_msm_batch_scratch_bytes(n, 254, count)
    pipe_bytes = 2 * ((L.h2_msm_scratch_bytes(n, 254) + 255) // 256 * 256)
    assert fused_bytes > pipe_bytes
    for nbytes in (fused_bytes, pipe_bytes):
        scratch = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        out = np.zeros((count, 12), dtype=np.uint64)
        assert L.h2_dev_msm_batch(ptrs, count, d_pts.data_ptr(), n, 254, scratch.data_ptr(), nbytes, out.ctypes.data, None) == 0
        assert [_affine(oracle, out[j]) for j in range(count)] == want, nbytes


@pytest.mark.parametrize("n", [1 << 15, (1 << 17) + 4321])
def test_msm_narrow_columns_row_ranges(oracle, n):
    """narrow witness columns (booleans, bytes, a 10- / 12-bit code): one window of a few buckets, which the library
    cuts into row ranges acting as separate windows (msm_shape) -- with and without a dominant value, ragged n, and the
    bound given exactly or loosely; against the oracle"""
    pts = oracle.random_g1(3030 + n, n)
    rows = np.arange(n, dtype=np.uint64)
    cases = {
        "boolean_sparse": ((rows * 2654435761 >> 7) % 16 == 0).astype(np.uint64),        # 1/16 ones: not dominant
        "boolean_dense": ((rows * 2654435761 >> 7) % 4 != 0).astype(np.uint64),           # 3/4 ones: dominant value
        "bytes": (rows * 40503 >> 3) % 256,
        "ten_bits": (rows * 2654435761 >> 5) % 1000,
        "twelve_bits_mostly_seven": np.where(rows % 3 == 0, (rows * 48271) % 4096, 7).astype(np.uint64),
    }
    for name, vals in cases.items():
        scalars = to_mont([int(v) for v in vals])
        want = _affine(oracle, oracle.best_multiexp(scalars, pts))
        top = max(1, int(vals.max()).bit_length())
        for bits in (top, top + 1, 13):
            if bits < top:
                continue
            assert _affine(oracle, ar.gpu_multiexp_single_gpu_with_bound(scalars, pts, bits)) == want, (name, bits)


def test_msm_fused_narrow_columns_at_2p22(oracle):
    """16-bit witness columns at 2^22 rows are committed as ONE fused MSM (columns of one or two windows fuse up to 2^24
    entries each; wider ones stop at 2^22): the fused batch gives the points of the single-column path (row ranges, its own
    sort / finish / reduce) for every column -- dense 16-bit, a 12-bit code, 1/16 sparse, mostly one value -- and, for the
    sparse column, of the oracle over its non-zero rows"""
    import torch

    L = h2.lib()
    n = 1 << 22
    d_pts = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    assert L.h2_dev_random_points(424242, n, d_pts.data_ptr(), None) == 0
    rows = np.arange(n, dtype=np.uint64)
    vals = [
        (rows * np.uint64(2654435761) >> np.uint64(5)) % np.uint64(1 << 16),
        (rows * np.uint64(40503) >> np.uint64(3)) % np.uint64(4096),
        np.where(rows % np.uint64(16) == 0, (rows * np.uint64(48271)) % np.uint64(1 << 16), 0).astype(np.uint64),
        np.where(rows % np.uint64(3) == 0, (rows * np.uint64(69621)) % np.uint64(1 << 16), 7).astype(np.uint64),
    ]
    d_cols = []
    for v in vals:
        a = np.zeros((n, 4), dtype=np.uint64)
        a[:, 0] = v
        t = torch.from_numpy(a.view(np.int64)).cuda()
        assert L.h2_dev_batch_mont(t.data_ptr(), n, None) == 0
        d_cols.append(t)
    count = len(d_cols)
    ptrs = (ctypes.c_void_p * count)(*[t.data_ptr() for t in d_cols])
    nbytes = L.h2_msm_batch_scratch_bytes(n, 16, count)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    out = np.zeros((count, 12), dtype=np.uint64)
    assert L.h2_dev_msm_batch(ptrs, count, d_pts.data_ptr(), n, 16, scratch.data_ptr(), nbytes, out.ctypes.data, None) == 0
    single_bytes = L.h2_msm_scratch_bytes(n, 16)
    for j in range(count):
        one = np.zeros(12, dtype=np.uint64)
        assert L.h2_dev_msm(d_cols[j].data_ptr(), d_pts.data_ptr(), n, 16, scratch.data_ptr(), single_bytes, one.ctypes.data, None) == 0
        assert _affine(oracle, out[j]) == _affine(oracle, one), j
    live = np.nonzero(vals[2])[0]
    pts_live = d_pts[torch.from_numpy(live.astype(np.int64)).cuda()].cpu().numpy().view(np.uint64)
    sc_live = d_cols[2][torch.from_numpy(live.astype(np.int64)).cuda()].cpu().numpy().view(np.uint64)
    assert _affine(oracle, out[2]) == _affine(oracle, oracle.best_multiexp(np.ascontiguousarray(sc_live), np.ascontiguousarray(pts_live)))


@pytest.mark.timeout(300)
def test_msm_randomised_shapes():
    """tools/msm_fuzz.py for a short budget: random sizes, bounds and value distributions against the oracle, windowed and
    over shifted-base tables with random digit counts (the long runs -- thousands of cases -- are recorded in DESIGN.md)"""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "msm_fuzz.py"), "20", "11", "tables"], capture_output=True,
                         text=True, timeout=280)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "all equal to the oracle" in out.stdout and "over tables" in out.stdout
    # the one-lane-per-bucket fold (used from 2^21 buckets on) forced onto every shape
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "msm_fuzz.py"), "10", "12", "tables"], capture_output=True,
                         text=True, timeout=280, env=dict(os.environ, H2_MSM_FINISH_LANE_LOG="0"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "all equal to the oracle" in out.stdout


@pytest.mark.parametrize("k,j", [(0, 3), (1, 3), (3, 5), (5, 6), (8, 3), (9, 5), (11, 5), (12, 6), (13, 3), (16, 5), (18, 6), (20, 5)])
def test_fused_coset_transforms_vs_oracle(oracle, k, j):
    """h2_dev_coset_ntt / h2_dev_coset_intt (the powers of the coset generator fused into the first pass's load / the last
    pass's store): the values of a coefficient vector on EVERY coset g_c H of the extended domain (g_c = zeta
    extended_omega^c) are the entries c, c + 2^(extended_k - k), ... of the oracle's coeff_to_extended
    (poly/domain.rs:270-287); the inverse returns the coefficients; out-of-place reads leave the source untouched and the
    in-place form gives the same values"""
    import torch

    from h2util import R_MOD, from_mont
    from halo2_gpu_specific_amd._lib import check

    L = h2.lib()
    d, _ = oracle.domain(j, k)
    n, en = 1 << d.k, 1 << d.extended_k
    c = en // n
    coeffs = oracle.random_fr(4100 + k, n)
    ext = oracle.coeff_to_extended(coeffs, d)
    dev = torch.device("cuda", 0)
    src = torch.from_numpy(coeffs.view(np.int64)).to(dev)
    keep = src.clone()
    tmp = torch.empty_like(src)
    zeta, w_ext = from_mont(d.fr("g_coset"))[0], from_mont(d.fr("extended_omega"))[0]
    omega, omega_inv, n_inv = d.fr("omega"), d.fr("omega_inv"), d.fr("ifft_divisor")
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    for cs in sorted({0, 1, c - 1, c // 2}):
        g = zeta * pow(w_ext, cs, R_MOD) % R_MOD
        out = torch.empty_like(src)
        torch.cuda.synchronize()
        check(L.h2_dev_coset_ntt(src.data_ptr(), out.data_ptr(), tmp.data_ptr(), k, vp(fr_mont(g)), vp(omega), None), "h2_dev_coset_ntt")
        torch.cuda.synchronize()
        assert torch.equal(src, keep)
        got = out.cpu().numpy().view(np.uint64)
        assert np.array_equal(got, ext[cs::c]), (k, j, cs)
        inplace = src.clone()
        torch.cuda.synchronize()            # (the library's default stream is not torch's)
        check(L.h2_dev_coset_ntt(inplace.data_ptr(), inplace.data_ptr(), tmp.data_ptr(), k, vp(fr_mont(g)), vp(omega), None), "h2_dev_coset_ntt")
        torch.cuda.synchronize()
        assert torch.equal(inplace, out)
        check(L.h2_dev_coset_intt(out.data_ptr(), tmp.data_ptr(), k, vp(fr_mont(pow(g, -1, R_MOD))), vp(omega_inv), vp(n_inv), None),
              "h2_dev_coset_intt")
        torch.cuda.synchronize()
        assert torch.equal(out, keep), (k, j, cs)


@pytest.mark.parametrize("k", [0, 4, 9, 13, 18, 20])
def test_batched_transforms_equal_single_ones(oracle, k):
    """h2_dev_ntt_batch / h2_dev_intt_batch / h2_dev_coset_ntt_batch (up to 16 vectors per launch, 19 here: two chunks, a
    ragged one) against the oracle's transforms vector by vector"""
    import torch

    from h2util import R_MOD, from_mont
    from halo2_gpu_specific_amd._lib import check

    L = h2.lib()
    count, n = 19, 1 << k
    d, _ = oracle.domain(3, k)
    dev = torch.device("cuda", 0)
    cols = [oracle.random_fr(5200 + 31 * k + i, n) for i in range(count)]
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    omega, omega_inv, n_inv = d.fr("omega"), d.fr("omega_inv"), d.fr("ifft_divisor")
    ts = [torch.from_numpy(c.view(np.int64)).to(dev) for c in cols]
    tmp = torch.empty((16 * n, 4), dtype=torch.int64, device=dev)
    ptrs = (ctypes.c_void_p * count)(*[t.data_ptr() for t in ts])
    torch.cuda.synchronize()
    check(L.h2_dev_ntt_batch(ptrs, count, tmp.data_ptr(), vp(omega), k, None), "h2_dev_ntt_batch")
    torch.cuda.synchronize()
    fwd = [oracle.best_fft(c, omega, k, threads=8) for c in cols]
    for t, want in zip(ts, fwd):
        assert np.array_equal(t.cpu().numpy().view(np.uint64), want)
    check(L.h2_dev_intt_batch(ptrs, count, tmp.data_ptr(), vp(omega_inv), vp(n_inv), k, None), "h2_dev_intt_batch")
    torch.cuda.synchronize()
    for t, c in zip(ts, cols):
        assert np.array_equal(t.cpu().numpy().view(np.uint64), c)
    # one coset of the extended domain, out of place
    ext_c = 1 << (d.extended_k - d.k)
    zeta, w_ext = from_mont(d.fr("g_coset"))[0], from_mont(d.fr("extended_omega"))[0]
    cs = ext_c - 1
    g = zeta * pow(w_ext, cs, R_MOD) % R_MOD
    outs = [torch.empty_like(t) for t in ts]
    optrs = (ctypes.c_void_p * count)(*[t.data_ptr() for t in outs])
    torch.cuda.synchronize()
    check(L.h2_dev_coset_ntt_batch(ptrs, optrs, count, tmp.data_ptr(), k, vp(fr_mont(g)), vp(omega), None), "h2_dev_coset_ntt_batch")
    torch.cuda.synchronize()
    for i in (0, 7, 15, 16, 18):
        assert np.array_equal(outs[i].cpu().numpy().view(np.uint64), oracle.coeff_to_extended(cols[i], d)[cs::ext_c]), i
        assert np.array_equal(ts[i].cpu().numpy().view(np.uint64), cols[i])
    # the whole extended domain, zero-padded and zeta-scaled in the first pass: h2_dev_coeff_to_extended_batch
    if k <= 18:
        en = 1 << d.extended_k
        exts = [torch.empty((en, 4), dtype=torch.int64, device=dev) for _ in ts]
        etmp = torch.empty((16 * en, 4), dtype=torch.int64, device=dev)
        eptrs = (ctypes.c_void_p * count)(*[t.data_ptr() for t in exts])
        torch.cuda.synchronize()
        check(L.h2_dev_coeff_to_extended_batch(ptrs, eptrs, count, etmp.data_ptr(), d.k, d.extended_k, vp(d.fr("g_coset")),
                                               vp(d.fr("g_coset_inv")), vp(d.fr("extended_omega")), None), "h2_dev_coeff_to_extended_batch")
        torch.cuda.synchronize()
        for i in (0, 15, 16, 18):
            assert np.array_equal(exts[i].cpu().numpy().view(np.uint64), oracle.coeff_to_extended(cols[i], d)), i


@pytest.mark.parametrize("slots", [1, 2, 3])
def test_concurrent_host_slice_callers_over_the_host_api_slots(tmp_path, slots):
    """Two (or three) host-API slots per device (H2_HOST_SLOTS): concurrent host-slice callers -- the reference's entry
    points are invoked from rayon workers -- overlap one call's transfers with another's kernels.  Six threads mix
    transforms that share plans and last-pass tables (2^18), MSMs over ONE registered SRS (its device copy and table shared by
    the slots), coset extensions and Horner evaluations; every result equals the one the same call gives alone."""
    import subprocess
    import sys

    from h2util import ROOT as REPO

    script = tmp_path / "worker.py"
    script.write_text(r"""
import os, sys, threading
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch
torch.cuda.init()
import halo2_gpu_specific_amd as h2
from halo2_gpu_specific_amd import arithmetic as ar
from h2util import R_MOD, fr_mont
L = h2.lib()
ROOT_W = 0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C
rng = np.random.default_rng(9)
def vec(n):
    return rng.integers(0, 2**62, size=(n, 4), dtype=np.uint64)
log_n, m = 18, 1 << 16
w = fr_mont(pow(ROOT_W, 1 << (28 - log_n), R_MOD))
w_inv, n_inv = fr_mont(pow(pow(ROOT_W, 1 << (28 - log_n), R_MOD), -1, R_MOD)), fr_mont(pow(1 << log_n, -1, R_MOD))
d_pts = torch.empty((m, 8), dtype=torch.int64, device="cuda")
L.h2_dev_random_points(11, m, d_pts.data_ptr(), None); L.h2_synchronize()
pts = d_pts.cpu().numpy().view(np.uint64)
assert L.h2_bases_register(pts.ctypes.data, m) == 0
xs = [vec(1 << log_n) for _ in range(6)]
ss = [vec(m) for _ in range(6)]
point = fr_mont(123456789)
def job(i):
    if i %% 3 == 0:
        return ("ntt", ar.best_fft(xs[i].copy(), w, log_n), ar.gpu_ifft(xs[i].copy(), w_inv, log_n, n_inv))
    if i %% 3 == 1:
        return ("msm", ar.gpu_multiexp_single_gpu_with_bound(ss[i], pts, 254), ar.gpu_multiexp_single_gpu_with_bound(ss[i][:m // 2], pts[:m // 2], 64))
    return ("eval", ar.eval_polynomial(xs[i], point), ar.best_fft(xs[i].copy(), w, log_n))
alone = [job(i) for i in range(6)]
errors, got = [], [None] * 6
def worker(i):
    try:
        for _ in range(3):
            got[i] = job(i)
    except Exception as e:
        errors.append((i, repr(e)))
ts = [threading.Thread(target=worker, args=(i,)) for i in range(6)]
[t.start() for t in ts]; [t.join() for t in ts]
assert not errors, errors
def same(a, b):
    if a[0] == "msm":      # Jacobian representations may differ with the summation order: compare the group elements
        from bench import jac_eq
        return all(jac_eq(p, q) for p, q in zip(a[1:], b[1:]))
    return all(np.array_equal(p, q) for p, q in zip(a[1:], b[1:]))
assert all(same(a, b) for a, b in zip(alone, got)), "a concurrent call differs from the same call alone"
assert L.h2_bases_unregister(pts.ctypes.data) == 0
assert L.h2_release_plans() == 0
print("SLOTS OK")
""" % (REPO, REPO))
    res = subprocess.run([sys.executable, str(script)], env=dict(os.environ, H2_HOST_SLOTS=str(slots)), capture_output=True,
                         text=True, timeout=280, cwd=REPO)
    assert res.returncode == 0 and "SLOTS OK" in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]
