"""numpy-facing mirror of the reference's hot-path functions
(/root/reference/halo2_proofs/src/arithmetic.rs, poly/domain.rs, poly/commitment.rs), each a thin
call into the C ABI.  Arrays are uint64: Fr = (n, 4), G1Affine = (n, 8), G1 = (12,).
Same names, argument meaning and error behaviour as the reference (asserts -> AssertionError)."""
import ctypes

import numpy as np

from ._lib import check, lib

OP_MUL_C, OP_SUM_C, OP_SUM, OP_MUL, OP_SUB, OP_LCTHETA, OP_LCBETA, OP_ADDGAMMA, OP_CONSTANT = range(9)
NUM_BITS = 254  # Fr::NUM_BITS
GPU_MSM_THRESHOLD = 1 << 14  # best_multiexp_gpu_cond: arithmetic.rs:446


def _fr(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    assert a.ndim >= 1 and a.shape[-1] == 4, "Fr arrays are (n, 4) uint64"
    return a


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def best_fft(a, omega, log_n):
    """arithmetic.rs:546-554 -- in place on `a` (2^log_n Fr), returns `a`."""
    a = _fr(a)
    assert a.shape[0] == 1 << log_n  # arithmetic.rs:569
    check(lib().h2_ntt(_p(a), _p(_fr(omega)), log_n), "h2_ntt")
    return a


def gpu_ifft(a, omega_inv, log_n, divisor):
    """arithmetic.rs:515-534"""
    a = _fr(a)
    assert a.shape[0] == 1 << log_n
    check(lib().h2_intt(_p(a), _p(_fr(omega_inv)), _p(_fr(divisor)), log_n), "h2_intt")
    return a


def gpu_multiexp_bound(coeffs, bases, max_bits):
    """arithmetic.rs:413-440 (multi-device split + host fold)"""
    coeffs = _fr(coeffs).reshape(-1, 4)
    bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, 8)
    assert len(coeffs) == len(bases)  # arithmetic.rs:466
    out = np.zeros(12, dtype=np.uint64)
    check(lib().h2_msm_multi(_p(coeffs), _p(bases), len(coeffs), max_bits, _p(out)), "h2_msm_multi")
    return out


def gpu_multiexp_single_gpu_with_bound(coeffs, bases, max_bits):
    """arithmetic.rs:334-367"""
    coeffs = _fr(coeffs).reshape(-1, 4)
    bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, 8)
    assert len(coeffs) == len(bases)
    out = np.zeros(12, dtype=np.uint64)
    check(lib().h2_msm(_p(coeffs), _p(bases), len(coeffs), max_bits, _p(out)), "h2_msm")
    return out


def gpu_multiexp(coeffs, bases):
    """arithmetic.rs:370-372"""
    return gpu_multiexp_bound(coeffs, bases, NUM_BITS)


def best_multiexp(coeffs, bases):
    """arithmetic.rs:465-492 -- on this build every size goes to the device (the reference only
    switches above 2^14, arithmetic.rs:446; the group element returned is the same)."""
    return gpu_multiexp(coeffs, bases)


def gpu_multiexp_bound_and_fft(coeffs, bases, max_bits, omega_inv, divisor, log_n):
    """arithmetic.rs:375-410 -- returns the commitment; `coeffs` is overwritten with the iFFT."""
    coeffs = _fr(coeffs)
    bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, 8)
    assert len(coeffs) == len(bases) == 1 << log_n
    out = np.zeros(12, dtype=np.uint64)
    check(
        lib().h2_msm_intt(_p(coeffs), _p(bases), len(coeffs), max_bits, _p(_fr(omega_inv)), _p(_fr(divisor)), log_n, _p(out)),
        "h2_msm_intt",
    )
    return out


def gpu_mont(a):
    a = _fr(a)
    check(lib().h2_batch_mont(_p(a), len(a)), "h2_batch_mont")
    return a


def gpu_unmont(a):
    a = _fr(a)
    check(lib().h2_batch_unmont(_p(a), len(a)), "h2_batch_unmont")
    return a


def coeff_to_extended(coeffs, k, extended_k, g_coset, g_coset_inv, extended_omega):
    """poly/domain.rs:270-287"""
    coeffs = _fr(coeffs)
    assert coeffs.shape[0] == 1 << k  # domain.rs:274
    out = np.zeros((1 << extended_k, 4), dtype=np.uint64)
    check(
        lib().h2_coeff_to_extended(_p(coeffs), _p(out), k, extended_k, _p(_fr(g_coset)), _p(_fr(g_coset_inv)), _p(_fr(extended_omega))),
        "h2_coeff_to_extended",
    )
    return out


def extended_to_coeff(a, k, extended_k, quotient_poly_degree, g_coset, g_coset_inv, extended_omega_inv, extended_ifft_divisor):
    """poly/domain.rs:328-350"""
    a = _fr(a)
    assert a.shape[0] == 1 << extended_k  # domain.rs:329
    out_len = (1 << k) * quotient_poly_degree
    out = np.zeros((out_len, 4), dtype=np.uint64)
    check(
        lib().h2_extended_to_coeff(
            _p(a), _p(out), out_len, extended_k, _p(_fr(g_coset)), _p(_fr(g_coset_inv)), _p(_fr(extended_omega_inv)), _p(_fr(extended_ifft_divisor))
        ),
        "h2_extended_to_coeff",
    )
    return out


def divide_by_vanishing_poly(a, t_evaluations):
    """poly/domain.rs:354-373 (in place)"""
    a, t = _fr(a), _fr(t_evaluations)
    check(lib().h2_divide_by_vanishing_poly(_p(a), len(a), _p(t), len(t)), "h2_divide_by_vanishing_poly")
    return a


def eval_op(op, l=None, r=None, l_rot=0, r_rot=0, c=None, size=None, res=None):
    """the elementwise kernels of SURVEY.md section 2.3 (eval_mul_c, eval_sum, ...)"""
    l = _fr(l) if l is not None else None
    r = _fr(r) if r is not None else None
    if size is None:
        size = len(l) if l is not None else len(r)
    if res is None:
        res = np.zeros((size, 4), dtype=np.uint64)
    cc = _fr(c) if c is not None else None
    check(lib().h2_eval_op(op, _p(res), _p(l), _p(r), l_rot, r_rot, size, _p(cc)), "h2_eval_op")
    return res


def eval_polynomial(poly, point):
    """arithmetic.rs:714-735"""
    poly = _fr(poly).reshape(-1, 4)
    out = np.zeros(4, dtype=np.uint64)
    check(lib().h2_eval_polynomial(_p(poly), len(poly), _p(_fr(point)), _p(out)), "h2_eval_polynomial")
    return out


def batch_invert(a):
    """arithmetic.rs:840-844 (in place)"""
    a = _fr(a)
    check(lib().h2_batch_invert(_p(a), len(a)), "h2_batch_invert")
    return a


def kate_division(a, b):
    """arithmetic.rs:754-773: a(X) / (X - b), no remainder"""
    a = _fr(a).reshape(-1, 4)
    q = np.zeros((max(len(a) - 1, 0), 4), dtype=np.uint64)
    check(lib().h2_kate_division(_p(a), len(a), _p(_fr(b)), _p(q)), "h2_kate_division")
    return q


def prefix_product(f, init, n=None):
    """z[0] = init, z[i] = z[i-1] * f[i-1] (permutation/prover.rs:151-160)"""
    f = _fr(f).reshape(-1, 4)
    n = len(f) + 1 if n is None else n
    z = np.zeros((n, 4), dtype=np.uint64)
    check(lib().h2_prefix_product(_p(f), n, _p(_fr(init)), _p(z)), "h2_prefix_product")
    return z
