"""GPU parity AT THE METRIC'S SIZES (BASELINE.json: k = 24), every output against the CPU oracle -- not against another
form of the HIP path and not on sampled outputs:

* 2^24-point MSM vs `oracle.best_multiexp` (arithmetic.rs:20-108, :465-492 restated): uniform 254-bit scalars through the
  windowed pipeline (host-buffer entry point `h2_msm`) and over a shifted-base table (`h2_dev_msm` after
  `h2_dev_bases_precompute`), and a column that is 7/8 one value (the shape of a grand-product column);
* the full 2^25-point NTT (the extended domain of a k = 24 proof, 8 + 8 + 9-bit passes) vs `oracle.best_fft`
  (arithmetic.rs:556-705), all 2^25 elements;
* `coeff_to_extended` -> `divide_by_vanishing_poly` -> `extended_to_coeff` at k = 24 / extended 2^25
  (poly/domain.rs:270-287, :354-373, :328-350), all elements;
* `gpu_multiexp_bound_and_fft` (commit_lagrange_and_ifft) at 2^24: the point and every coefficient.

The oracle side costs ~10 s of host time per MSM / transform on the GPU box's cores."""
import ctypes
import os

import numpy as np
import pytest

from halo2_gpu_specific_amd import arithmetic as ar
from h2util import R_MOD, arr_to_points, fr_mont
from test_gpu_msm_table import DevMsm

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(1500)]

S = 28
ROOT = 0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C
LOG_N = 24
# libgomp does not scale the oracle's rayon-shaped FFT recursion past ~32 threads (65 s per 2^24 transform on the GPU box's
# 256 hardware threads against 1.6 s on 32: DESIGN.md section 5); bench.py probes the team size the same way
FFT_THREADS = min(32, os.cpu_count() or 1)
ctypes_vp, ctypes_sz, ctypes_i32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int


def _affine(oracle, jac):
    return arr_to_points(oracle.to_affine(np.asarray(jac, dtype=np.uint64)))[0]


@pytest.fixture(scope="module")
def bases24(oracle):
    """2^24 curve points.  The oracle's try-and-increment generator (the 2^20 cases) takes a minute of host time
    at this size: these come from the library's generator (h2_dev_random_points, the same construction on the device) and a
    sample of them is checked here with big integers -- on the curve, in the field, not all alike.  What the MSM is compared
    with stays the oracle's own sum over exactly these points."""
    import torch

    from h2util import Q_MOD

    n = 1 << LOG_N
    t = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    from halo2_gpu_specific_amd._lib import check, lib
    check(lib().h2_dev_random_points(0x48414C4F32 + 24, n, t.data_ptr(), None), "h2_dev_random_points")
    check(lib().h2_synchronize(), "h2_synchronize")
    pts = np.ascontiguousarray(t.cpu().numpy().view(np.uint64))
    del t
    idx = np.unique(np.concatenate([np.arange(64), np.random.default_rng(24).integers(0, n, 2048), [n - 1]]))
    sample = arr_to_points(pts[idx])
    assert all(0 <= x < Q_MOD and 0 <= y < Q_MOD and (y * y - x * x * x - 3) % Q_MOD == 0 for x, y in sample)
    assert len(set(sample)) == len(sample)
    return pts


@pytest.fixture(scope="module")
def columns24(oracle):
    n = 1 << LOG_N
    uniform = oracle.random_fr(0x48414C4F32 + 2424, n)
    dominant = oracle.random_fr(0x48414C4F32 + 2425, n)
    dominant[n // 8:] = dominant[7]                     # 7/8 of the rows hold one 254-bit value (whole 256-row blocks of it)
    return {"uniform": uniform, "7/8 dominant": dominant}


@pytest.fixture(scope="module")
def want24(oracle, bases24, columns24):
    return {name: _affine(oracle, oracle.best_multiexp(col, bases24)) for name, col in columns24.items()}


def test_msm_2p24_windowed_vs_oracle(oracle, bases24, columns24, want24):
    """BASELINE's k = 24 MSM through `h2_msm` (scalars and bases uploaded by the call: no table exists for an
    unregistered host buffer, so this is the windowed pipeline: c = 17, 15 windows)"""
    for name, col in columns24.items():
        got = _affine(oracle, ar.gpu_multiexp_single_gpu_with_bound(col, bases24, 254))
        assert got == want24[name], name


def test_msm_2p24_over_table_vs_oracle(oracle, bases24, columns24, want24):
    """the same columns over the 12 GiB shifted-base table (12 digits of 21 / 22 bits, one shared bucket set, two-level
    partition): the form `bench.py`'s `msm_k24.over_shifted_base_table` and the k = 24 proof use"""
    dev = DevMsm(bases24)
    dev.precompute()
    try:
        for name, col in columns24.items():
            assert _affine(oracle, dev.msm(col, 254)) == want24[name], name
    finally:
        dev.forget()


def test_commit_lagrange_and_ifft_2p24_vs_oracle(oracle, bases24, columns24, want24):
    """gpu_multiexp_bound_and_fft (arithmetic.rs:375-410; Params::commit_lagrange_and_ifft, poly/commitment.rs:144-197) at
    k = 24: one upload feeds the MSM and the in-place inverse transform; the point equals the oracle's MSM and every
    coefficient the oracle's `ifft`"""
    d, _ = oracle.domain(1, LOG_N)
    vals = columns24["uniform"].copy()
    got = ar.gpu_multiexp_bound_and_fft(vals, bases24, 254, d.fr("omega_inv"), d.fr("ifft_divisor"), LOG_N)
    assert _affine(oracle, got) == want24["uniform"]
    assert np.array_equal(vals, oracle.ifft(columns24["uniform"], d.fr("omega_inv"), LOG_N, d.fr("ifft_divisor"), threads=FFT_THREADS))


def test_ntt_2p25_every_element_vs_oracle(oracle):
    log_n = 25
    n = 1 << log_n
    omega = pow(ROOT, 1 << (S - log_n), R_MOD)
    x = oracle.random_fr(0x25250000, n)
    got = ar.best_fft(x.copy(), fr_mont(omega), log_n)
    want = oracle.best_fft(x, fr_mont(omega), log_n, threads=FFT_THREADS)
    assert np.array_equal(got, want)
    del want
    back = ar.gpu_ifft(got, fr_mont(pow(omega, -1, R_MOD)), log_n, fr_mont(pow(n, -1, R_MOD)))
    assert np.array_equal(back, x)


def test_coset_divide_inverse_k24_vs_oracle(oracle):
    """the extended-domain leg of a k = 24 degree-3 proof (BASELINE configs[4]): 2^24 coefficients -> the 2^25-point coset,
    divided by the vanishing polynomial, back to 2 * 2^24 coefficients"""
    d, t = oracle.domain(3, LOG_N)
    assert d.extended_k == 25
    coeffs = oracle.random_fr(0x2424C0, 1 << LOG_N)
    ext = ar.coeff_to_extended(coeffs, d.k, d.extended_k, d.fr("g_coset"), d.fr("g_coset_inv"), d.fr("extended_omega"))
    want = oracle.coeff_to_extended(coeffs, d, threads=FFT_THREADS)
    assert np.array_equal(ext, want)
    del want
    div = ar.divide_by_vanishing_poly(ext.copy(), t)
    oracle.lib.oracle_divide_by_vanishing_poly(ext.ctypes.data, len(ext), t.ctypes.data, len(t), 64)
    assert np.array_equal(div, ext)
    got_c = ar.extended_to_coeff(
        div, d.k, d.extended_k, d.quotient_poly_degree, d.fr("g_coset"), d.fr("g_coset_inv"), d.fr("extended_omega_inv"),
        d.fr("extended_ifft_divisor"),
    )
    assert np.array_equal(got_c, oracle.extended_to_coeff(ext, d, threads=FFT_THREADS))


def test_ntt_2p26_every_element_vs_oracle(oracle):
    """four passes (2 + 8 + 8 + 8 bits): the extended domain of a k = 24 circuit of degree 5 (the wide circuit's shape)"""
    log_n = 26
    n = 1 << log_n
    omega = pow(ROOT, 1 << (S - log_n), R_MOD)
    x = oracle.random_fr(0x26260000, n)
    got = ar.best_fft(x.copy(), fr_mont(omega), log_n)
    want = oracle.best_fft(x, fr_mont(omega), log_n, threads=FFT_THREADS)
    assert np.array_equal(got, want)


@pytest.mark.skipif(os.environ.get("H2_TEST_NTT_2P28", "1") == "0", reason="H2_TEST_NTT_2P28=0")
def test_ntt_2p28_the_whole_two_adic_subgroup(oracle):
    """the largest transform the field has (Fr's 2-adicity is 28; 8 GiB per vector, 4 + 8 + 8 + 8 bits): outputs sampled
    against the oracle's Horner evaluation at omega^i (the DFT's definition, arithmetic.rs:714-735), then inverse o forward
    = identity on every element"""
    import torch

    from halo2_gpu_specific_amd import lib
    from halo2_gpu_specific_amd._lib import check

    log_n = 28
    n = 1 << log_n
    omega = ROOT
    L = lib()
    x = oracle.random_fr(0x28280000, n)
    dev = torch.device("cuda", 0)
    a = torch.from_numpy(x.view(np.int64)).to(dev)
    tmp = torch.empty_like(a)
    w, w_inv, n_inv = fr_mont(omega), fr_mont(pow(omega, -1, R_MOD)), fr_mont(pow(n, -1, R_MOD))   # alive across the calls
    check(L.h2_dev_ntt(a.data_ptr(), tmp.data_ptr(), w.ctypes.data, log_n, None), "h2_dev_ntt")
    torch.cuda.synchronize()
    out = np.zeros(4, dtype=np.uint64)
    for i in (0, 1, 0x1234567, n // 2 + 3, n - 1):
        point = fr_mont(pow(omega, i, R_MOD))
        oracle.lib.oracle_eval_polynomial_par.argtypes = [ctypes_vp, ctypes_sz, ctypes_vp, ctypes_i32, ctypes_vp]
        oracle.lib.oracle_eval_polynomial_par.restype = None
        oracle.lib.oracle_eval_polynomial_par(x.ctypes.data, n, point.ctypes.data, FFT_THREADS, out.ctypes.data)
        assert np.array_equal(a[i].cpu().numpy().view(np.uint64), out), "output %d" % i
    check(L.h2_dev_intt(a.data_ptr(), tmp.data_ptr(), w_inv.ctypes.data, n_inv.ctypes.data, log_n, None), "h2_dev_intt")
    torch.cuda.synchronize()
    assert torch.equal(a.cpu(), torch.from_numpy(x.view(np.int64)))


@pytest.mark.skipif(os.environ.get("H2_TEST_MSM_2P26") != "1", reason="opt-in (H2_TEST_MSM_2P26=1): ~2 minutes of oracle time")
def test_msm_2p26_beyond_the_metric_vs_oracle(oracle):
    """past the metric's size: 2^26 points (a k = 26 SRS: 4 GiB of bases, a 48 GiB shifted-base table of 12 digits, bucket
    ids of 21 / 22 bits, 2^26 x 12 = 8 x 10^8 sorted 32-bit entries), windowed and over the table, against the oracle"""
    n = 1 << 26
    bases = oracle.random_g1(0x48414C4F32 + 26, n)
    col = oracle.random_fr(0x48414C4F32 + 2626, n)
    want = _affine(oracle, oracle.best_multiexp(col, bases))
    dev = DevMsm(bases)
    assert _affine(oracle, dev.msm(col, 254)) == want, "windowed"
    dev.precompute()
    try:
        assert _affine(oracle, dev.msm(col, 254)) == want, "over the table"
    finally:
        dev.forget()
