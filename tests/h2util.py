"""Shared test plumbing: integer <-> limb conversion, golden loading and the ctypes
binding of the CPU oracle (oracle/liboracle.so).

The oracle is test infrastructure; nothing under halo2-gpu-specific_amd/ imports this.
"""
import ctypes
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
Q_MOD = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
MONT_R = 1 << 256
MASK64 = (1 << 64) - 1


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def h2i(s):
    return int(s, 16)


# ------------------------------------------------------------------ limbs
def int_to_limbs(v):
    return [(v >> (64 * i)) & MASK64 for i in range(4)]


def limbs_to_int(l):
    return int(l[0]) | (int(l[1]) << 64) | (int(l[2]) << 128) | (int(l[3]) << 192)


def ints_to_arr(vals):
    """list of ints (< 2^256) -> uint64 array (n, 4)"""
    a = np.zeros((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        a[i] = int_to_limbs(v)
    return a


def arr_to_ints(a):
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 4)
    return [limbs_to_int(row) for row in a]


def to_mont(vals, p=R_MOD):
    """canonical ints -> Montgomery-form uint64 array (n, 4)"""
    return ints_to_arr([v * MONT_R % p for v in vals])


def from_mont(a, p=R_MOD):
    rinv = pow(MONT_R, -1, p)
    return [v * rinv % p for v in arr_to_ints(a)]


def fr_mont(v):
    return to_mont([v], R_MOD)[0].copy()


def points_to_arr(pts):
    """list of (x, y) canonical ints (identity = (0, 0)) -> uint64 array (n, 8) Montgomery Fq"""
    a = np.zeros((len(pts), 8), dtype=np.uint64)
    for i, (x, y) in enumerate(pts):
        a[i, :4] = int_to_limbs(x * MONT_R % Q_MOD)
        a[i, 4:] = int_to_limbs(y * MONT_R % Q_MOD)
    return a


def arr_to_points(a):
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 8)
    rinv = pow(MONT_R, -1, Q_MOD)
    return [(limbs_to_int(r[:4]) * rinv % Q_MOD, limbs_to_int(r[4:]) * rinv % Q_MOD) for r in a]


def golden_points(lst):
    return [(h2i(x), h2i(y)) for x, y in lst]


# ------------------------------------------------------------------ oracle binding
def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class DomainT(ctypes.Structure):
    _fields_ = [
        ("n", ctypes.c_uint64),
        ("k", ctypes.c_uint64),
        ("extended_k", ctypes.c_uint64),
        ("quotient_poly_degree", ctypes.c_uint64),
        ("t_len", ctypes.c_uint64),
    ] + [
        (name, ctypes.c_uint64 * 4)
        for name in (
            "omega",
            "omega_inv",
            "extended_omega",
            "extended_omega_inv",
            "g_coset",
            "g_coset_inv",
            "ifft_divisor",
            "extended_ifft_divisor",
            "barycentric_weight",
        )
    ]

    def fr(self, name):
        return np.array(list(getattr(self, name)), dtype=np.uint64)


class Oracle:
    """ctypes view of oracle/liboracle.so (built on demand with `make -C oracle`)."""

    _inst = None

    @classmethod
    def get(cls):
        if cls._inst is None:
            cls._inst = cls()
        return cls._inst

    def __init__(self):
        path = os.path.join(ROOT, "oracle", "liboracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
        self.lib = ctypes.CDLL(path)
        L = self.lib
        vp, sz, i32, u32, u64, i64 = (
            ctypes.c_void_p,
            ctypes.c_size_t,
            ctypes.c_int,
            ctypes.c_uint32,
            ctypes.c_uint64,
            ctypes.c_int64,
        )
        sigs = {
            "oracle_fr_mul": [vp, vp, vp],
            "oracle_fr_add": [vp, vp, vp],
            "oracle_fr_sub": [vp, vp, vp],
            "oracle_fr_inv": [vp, vp],
            "oracle_fq_mul": [vp, vp, vp],
            "oracle_fq_add": [vp, vp, vp],
            "oracle_fq_sub": [vp, vp, vp],
            "oracle_fq_inv": [vp, vp],
            "oracle_from_repr_batch": [vp, sz, i32],
            "oracle_to_repr_batch": [vp, sz, i32],
            "oracle_g1_to_affine": [vp, vp],
            "oracle_g1_add": [vp, vp, vp],
            "oracle_g1_add_affine": [vp, vp, vp],
            "oracle_g1_double": [vp, vp],
            "oracle_g1_mul": [vp, vp, vp],
            "oracle_random_fr": [u64, sz, vp],
            "oracle_random_g1": [u64, sz, vp],
            "oracle_multiexp_serial": [vp, vp, sz, vp],
            "oracle_best_multiexp": [vp, vp, sz, i32, vp],
            "oracle_best_multiexp_gpu_cond": [vp, vp, sz, i32, vp],
            "oracle_small_multiexp": [vp, vp, sz, vp],
            "oracle_commit_lagrange_with_bound": [vp, vp, sz, i32, vp],
            "oracle_best_fft": [vp, vp, u32, i32],
            "oracle_best_fft_st": [vp, vp, u32],
            "oracle_ifft": [vp, vp, u32, vp, i32],
            "oracle_distribute_powers_zeta": [vp, sz, vp, vp, i32, i32],
            "oracle_coeff_to_extended": [vp, u32, u32, vp, vp, vp, vp, i32],
            "oracle_extended_to_coeff": [vp, u32, u32, u64, vp, vp, vp, vp, i32],
            "oracle_divide_by_vanishing_poly": [vp, sz, vp, sz, i32],
            "oracle_poly_add": [vp, vp, sz, i32],
            "oracle_poly_sub": [vp, vp, sz, i32],
            "oracle_poly_scale": [vp, vp, sz, i32],
            "oracle_domain_new": [u32, u32, vp, ctypes.POINTER(DomainT), vp, sz],
            "oracle_l_i_range": [ctypes.POINTER(DomainT), vp, vp, vp, sz, vp],
            "oracle_eval_polynomial": [vp, sz, vp, vp],
            "oracle_kate_division": [vp, sz, vp, vp],
            "oracle_batch_invert": [vp, sz],
            "oracle_lagrange_interpolate": [vp, vp, sz, vp],
            "oracle_unsafe_setup": [u32, vp, vp, vp],
            "oracle_eval_op": [i32, vp, vp, vp, i64, i64, sz, vp],
        }
        for name, args in sigs.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = None
        L.oracle_g1_eq.argtypes = [vp, vp]
        L.oracle_g1_eq.restype = i32
        L.oracle_g1_on_curve.argtypes = [vp]
        L.oracle_g1_on_curve.restype = i32
        L.oracle_find_max_scalar_bits.argtypes = [vp, sz]
        L.oracle_find_max_scalar_bits.restype = u32
        L.oracle_domain_new.restype = i32
        L.oracle_extended_to_coeff.restype = sz
        L.oracle_points_compress.argtypes = [vp, sz, vp]
        L.oracle_points_compress.restype = None
        L.oracle_points_decompress.argtypes = [vp, sz, vp]
        L.oracle_points_decompress.restype = sz
        self.threads = os.cpu_count() or 1
        # libgomp does not scale the rayon-shaped FFT recursion past ~32 threads (65 s per 2^24 transform on 256 hardware
        # threads against 1.6 s on 32: DESIGN.md section 5); the MSM restatement does scale
        self.fft_threads = min(self.threads, 32)

    # ---- thin pythonic wrappers (arrays are uint64, C-contiguous) ----
    def op2(self, name, a, b):
        out = np.zeros(4, dtype=np.uint64)
        getattr(self.lib, name)(_ptr(np.ascontiguousarray(a)), _ptr(np.ascontiguousarray(b)), _ptr(out))
        return out

    def op1(self, name, a):
        out = np.zeros(4, dtype=np.uint64)
        getattr(self.lib, name)(_ptr(np.ascontiguousarray(a)), _ptr(out))
        return out

    def random_fr(self, seed, n):
        out = np.zeros((n, 4), dtype=np.uint64)
        self.lib.oracle_random_fr(seed, n, _ptr(out))
        return out

    def random_g1(self, seed, n):
        out = np.zeros((n, 8), dtype=np.uint64)
        self.lib.oracle_random_g1(seed, n, _ptr(out))
        return out

    def points_compress(self, pts):
        """(n, 8) affine Montgomery points -> (n, 32) uint8 encodings"""
        pts = np.ascontiguousarray(pts, dtype=np.uint64).reshape(-1, 8)
        out = np.zeros((len(pts), 32), dtype=np.uint8)
        self.lib.oracle_points_compress(_ptr(pts), len(pts), out.ctypes.data_as(ctypes.c_void_p))
        return out

    def points_decompress(self, raw):
        """(n, 32) uint8 encodings -> ((n, 8) affine Montgomery points, number of invalid encodings)"""
        raw = np.ascontiguousarray(raw, dtype=np.uint8).reshape(-1, 32)
        out = np.zeros((len(raw), 8), dtype=np.uint64)
        bad = self.lib.oracle_points_decompress(raw.ctypes.data_as(ctypes.c_void_p), len(raw), _ptr(out))
        return out, int(bad)

    def best_fft(self, a, omega, log_n, threads=None):
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        self.lib.oracle_best_fft(_ptr(a), _ptr(np.ascontiguousarray(omega)), log_n, threads or self.fft_threads)
        return a

    def best_fft_st(self, a, omega, log_n):
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        self.lib.oracle_best_fft_st(_ptr(a), _ptr(np.ascontiguousarray(omega)), log_n)
        return a

    def ifft(self, a, omega_inv, log_n, divisor, threads=None):
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        self.lib.oracle_ifft(
            _ptr(a), _ptr(np.ascontiguousarray(omega_inv)), log_n, _ptr(np.ascontiguousarray(divisor)), threads or self.fft_threads
        )
        return a

    def best_multiexp(self, coeffs, bases, threads=None):
        coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64)
        bases = np.ascontiguousarray(bases, dtype=np.uint64)
        out = np.zeros(12, dtype=np.uint64)
        self.lib.oracle_best_multiexp_gpu_cond(_ptr(coeffs), _ptr(bases), len(coeffs), threads or self.threads, _ptr(out))
        return out

    def multiexp_serial(self, coeffs, bases):
        coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64)
        bases = np.ascontiguousarray(bases, dtype=np.uint64)
        out = np.zeros(12, dtype=np.uint64)
        out[4:8] = to_mont([1], Q_MOD)[0]  # identity (0, 1, 0)
        self.lib.oracle_multiexp_serial(_ptr(coeffs), _ptr(bases), len(coeffs), _ptr(out))
        return out

    def small_multiexp(self, coeffs, bases):
        coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64)
        bases = np.ascontiguousarray(bases, dtype=np.uint64)
        out = np.zeros(12, dtype=np.uint64)
        self.lib.oracle_small_multiexp(_ptr(coeffs), _ptr(bases), len(coeffs), _ptr(out))
        return out

    def to_affine(self, jac):
        out = np.zeros(8, dtype=np.uint64)
        self.lib.oracle_g1_to_affine(_ptr(np.ascontiguousarray(jac, dtype=np.uint64)), _ptr(out))
        return out

    def g1_mul(self, p_aff, k_mont):
        out = np.zeros(12, dtype=np.uint64)
        self.lib.oracle_g1_mul(_ptr(np.ascontiguousarray(p_aff)), _ptr(np.ascontiguousarray(k_mont)), _ptr(out))
        return out

    def domain(self, j, k, zeta=None):
        d = DomainT()
        t_cap = 1 << 8
        t = np.zeros((t_cap, 4), dtype=np.uint64)
        rc = self.lib.oracle_domain_new(j, k, _ptr(zeta) if zeta is not None else None, ctypes.byref(d), _ptr(t), t_cap)
        assert rc == 0, rc
        return d, t[: d.t_len].copy()

    def coeff_to_extended(self, a, d, threads=None):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        out = np.zeros((1 << d.extended_k, 4), dtype=np.uint64)
        self.lib.oracle_coeff_to_extended(
            _ptr(a), d.k, d.extended_k, _ptr(d.fr("g_coset")), _ptr(d.fr("g_coset_inv")), _ptr(d.fr("extended_omega")),
            _ptr(out), threads or self.fft_threads,
        )
        return out

    def extended_to_coeff(self, a, d, threads=None):
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        m = self.lib.oracle_extended_to_coeff(
            _ptr(a), d.k, d.extended_k, d.quotient_poly_degree, _ptr(d.fr("g_coset")), _ptr(d.fr("g_coset_inv")),
            _ptr(d.fr("extended_omega_inv")), _ptr(d.fr("extended_ifft_divisor")), threads or self.fft_threads,
        )
        return a[:m].copy()

    def eval_op(self, op, l, r, l_rot, r_rot, c):
        size = len(l) if l is not None else len(r)
        res = np.zeros((size, 4), dtype=np.uint64)
        self.lib.oracle_eval_op(
            op, _ptr(res), _ptr(l) if l is not None else None, _ptr(r) if r is not None else None, l_rot, r_rot, size,
            _ptr(c) if c is not None else None,
        )
        return res
