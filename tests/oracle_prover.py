"""tests/oracle_prover.py -- TEST INFRASTRUCTURE: a whole keygen + create_proof on the CPU, every vector pass computed by
the C oracle (oracle/oracle.c: the restatement of the reference's rayon CPU path).

What it is for
  * "proof bytes == CPU" (BASELINE configs[3]) at the sizes of the metric: the device prover's bytes against a CPU run
    over the same SRS, witness and randomness at k = 20 .. 24 (tests/test_gpu_cpu_prover.py).  The big-integer prover of
    tests/ref_plonk.py is the INDEPENDENT restatement of the protocol (own orchestration, Python integers) and pins the
    bytes up to k = 22 through hashes; this one shares the host orchestration of halo2-gpu-specific_amd/prover.py and
    swaps every kernel for the oracle's loop, so it checks the kernels IN CONTEXT (every launch of a proof, its sizes,
    aliasing and ordering) at any size the host memory holds.
  * bench.py's CPU create_proof baseline, timed on the GPU box's host cores in the same run.

How: `OracleDevice` is a `prover.Device` whose buffers are CPU torch tensors and whose `L` is `OracleLib`, an object with
the h2_* entry points the host prover calls, each forwarding to the oracle function that restates the same reference
lines.  Nothing in the product package selects it: `prover.Device()` still refuses to run without a HIP device.
"""
import contextlib
import ctypes
import os
import sys

import numpy as np

from h2util import Oracle, ROOT

sys.path.insert(0, ROOT)
from halo2_gpu_specific_amd import prover as P  # noqa: E402
from halo2_gpu_specific_amd.rng import chacha20_blocks  # noqa: E402

_vp, _sz, _i32, _u32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint32


def _a(x):
    """whatever the host prover hands a pointer parameter -> something a c_void_p argtype accepts"""
    if x is None or isinstance(x, int):
        return x
    if isinstance(x, (ctypes.Array, ctypes.Structure)):
        return ctypes.addressof(x)
    return x                                     # byref(...) objects pass through


class OracleLib:
    """the h2_* entry points of include/halo2_hip.h that prover.py calls, over host memory, by the oracle"""

    def __init__(self, threads=None):
        self.o = Oracle.get()
        O = self.o.lib
        # the host cores the CPU run uses: up to 32 (a k = 20 proof on a 256-thread host: 7.4 / 7.2 / 8.2 / 9.4 / 16.5 s on
        # 16 / 32 / 64 / 128 / 256 threads -- libgomp's fork-join over the short loops of a proof and the serial scans
        # dominate beyond that: profiles/r3_cpu_prover_threads.txt)
        self.threads = threads or int(os.environ.get("H2_ORACLE_THREADS", "0")) or min(self.o.threads, 32)
        self.fft_threads = min(self.threads, self.o.fft_threads)
        O.oracle_set_threads.argtypes, O.oracle_set_threads.restype = [_i32], None
        O.oracle_set_threads(self.threads)
        sigs = {
            "oracle_prefix_product": [_vp, _sz, _vp, _vp], "oracle_prefix_sum": [_vp, _sz, _vp, _vp],
            "oracle_lincomb": [_vp, _vp, _vp, _sz, _sz],
            "oracle_permutation_terms": [_vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _i32],
            "oracle_permutation_sigma": [_vp, _vp, _vp, _sz, _vp, _vp],
            "oracle_distribute_powers": [_vp, _sz, _vp],
            "oracle_eval_polynomial_par": [_vp, _sz, _vp, _i32, _vp],
            "oracle_batch_invert_par": [_vp, _sz, _i32],
            "oracle_reduce_wide_253": [_vp, _sz, _vp],
            "oracle_evaluate_h": [_vp, _vp],
        }
        for name, args in sigs.items():
            fn = getattr(O, name)
            fn.argtypes, fn.restype = args, None
        O.oracle_max_bits_canonical.argtypes, O.oracle_max_bits_canonical.restype = [_vp, _sz], _u32
        O.oracle_logup_multiplicity.argtypes = [_vp, _vp, _sz, _sz, _sz, _vp]
        O.oracle_logup_multiplicity.restype = _sz
        self.O = O

    # -- transforms (arithmetic.rs:556-705, poly/domain.rs:233-350) ------------------------------------------------------
    def h2_dev_ntt(self, a, tmp, omega, k, stream):
        self.O.oracle_best_fft(a, _a(omega), k, self.fft_threads)
        return 0

    def h2_dev_intt(self, a, tmp, omega_inv, divisor, k, stream):
        self.O.oracle_ifft(a, _a(omega_inv), k, _a(divisor), self.fft_threads)
        return 0

    def h2_dev_coeff_to_extended(self, a, out, tmp, k, ek, g, g_inv, ext_omega, stream):
        self.O.oracle_coeff_to_extended(a, k, ek, _a(g), _a(g_inv), _a(ext_omega), out, self.fft_threads)
        return 0

    def h2_dev_extended_to_coeff(self, a, tmp, ek, g, g_inv, ext_omega_inv, ext_divisor, stream):
        self.O.oracle_ifft(a, _a(ext_omega_inv), ek, _a(ext_divisor), self.fft_threads)
        self.O.oracle_distribute_powers_zeta(a, 1 << ek, _a(g), _a(g_inv), 0, self.threads)
        return 0

    def h2_dev_coset_ntt(self, a, out, tmp, k, g, omega, stream):
        """coeff_to_extended (poly/domain.rs:270-287) restricted to one coset: a[t] g^t, then the n-point transform"""
        if out != a:
            ctypes.memmove(out, a, 32 << k)
        self.O.oracle_distribute_powers(out, 1 << k, _a(g))
        self.O.oracle_best_fft(out, _a(omega), k, self.fft_threads)
        return 0

    def h2_dev_coset_intt(self, a, tmp, k, g_inv, omega_inv, divisor, stream):
        self.O.oracle_ifft(a, _a(omega_inv), k, _a(divisor), self.fft_threads)
        self.O.oracle_distribute_powers(a, 1 << k, _a(g_inv))
        return 0

    def h2_dev_coset_ntt_batch(self, srcs, dsts, count, tmp, k, g, omega, stream):
        for i in range(count):
            self.h2_dev_coset_ntt(srcs[i], dsts[i], tmp, k, g, omega, stream)
        return 0

    def h2_dev_coeff_to_extended_batch(self, srcs, dsts, count, tmp, k, ek, g, g_inv, ext_omega, stream):
        for i in range(count):
            self.h2_dev_coeff_to_extended(srcs[i], dsts[i], tmp, k, ek, g, g_inv, ext_omega, stream)
        return 0

    def h2_dev_intt_batch(self, ptrs, count, tmp, omega_inv, divisor, k, stream):
        for i in range(count):
            self.h2_dev_intt(ptrs[i], tmp, omega_inv, divisor, k, stream)
        return 0

    def h2_dev_distribute_powers(self, a, n, g, stream):
        self.O.oracle_distribute_powers(a, n, _a(g))
        return 0

    def h2_dev_divide_by_vanishing_poly(self, a, size, t_evals, t_len, stream):
        self.O.oracle_divide_by_vanishing_poly(a, size, t_evals, t_len, self.threads)
        return 0

    # -- commitments (arithmetic.rs:465-492 best_multiexp) ---------------------------------------------------------------
    def h2_msm_scratch_bytes(self, n, bits):
        return 64

    def h2_msm_batch_scratch_bytes(self, n, bits, count):
        return 64

    def h2_logup_scratch_bytes(self, n):
        return 64

    def h2_dev_msm(self, scalars, bases, n, max_bits, scratch, nbytes, out, stream):
        self.O.oracle_best_multiexp_gpu_cond(scalars, bases, n, self.threads, out)
        return 0

    def h2_dev_msm_batch_ex(self, sp, bp, bits, count, n, scratch, nbytes, out, stream):
        for j in range(count):
            self.O.oracle_best_multiexp_gpu_cond(sp[j], bp[j], n, self.threads, out + 96 * j)
        return 0

    def h2_dev_bases_precompute_bytes(self, n, digits):
        return 0

    def h2_dev_bases_precompute(self, bases, n, digits, stream):
        return 0

    def h2_dev_bases_forget(self, bases):
        return 0

    def h2_set_table_budget(self, nbytes):
        return 0

    # -- elementwise, scans, evaluations ---------------------------------------------------------------------------------
    def h2_dev_eval_op(self, op, res, l, r, l_rot, r_rot, size, c, stream):
        self.O.oracle_eval_op(op, res, l, r, l_rot, r_rot, size, _a(c))
        return 0

    def h2_dev_lincomb(self, res, ptrs, coeffs, count, size, stream):
        self.O.oracle_lincomb(res, _a(ptrs), coeffs, count, size)
        return 0

    def h2_dev_eval_polynomial(self, poly, n, point, out, stream):
        self.O.oracle_eval_polynomial_par(poly, n, _a(point), self.threads, _a(out))
        return 0

    def h2_dev_eval_polynomial_batch(self, ptrs, count, n, points, out, stream):
        for j in range(count):
            self.O.oracle_eval_polynomial_par(ptrs[j], n, points + 32 * j, self.threads, out + 32 * j)
        return 0

    def h2_dev_kate_division(self, a, n, b, q, stream):
        self.O.oracle_kate_division(a, n, _a(b), q)
        return 0

    def h2_dev_prefix_product(self, f, n, init, z, stream):
        self.O.oracle_prefix_product(f, n, _a(init), z)
        return 0

    def h2_dev_prefix_sum(self, f, n, init, z, stream):
        self.O.oracle_prefix_sum(f, n, _a(init), z)
        return 0

    def h2_dev_batch_invert(self, a, tmp, n, stream):
        self.O.oracle_batch_invert_par(a, n, self.threads)
        return 0

    def h2_dev_batch_mont(self, a, n, stream):
        self.O.oracle_from_repr_batch(a, n, 0)
        return 0

    def h2_dev_batch_unmont(self, a, n, stream):
        self.O.oracle_to_repr_batch(a, n, 0)
        return 0

    def h2_dev_max_scalar_bits(self, ptrs, count, n, words, out, stream):
        for j in range(count):
            out[j] = self.O.oracle_max_bits_canonical(ptrs[j], n)
        return 0

    # -- the arguments' own passes ---------------------------------------------------------------------------------------
    def h2_dev_permutation_terms(self, num, den, value, sigma, n, beta, gamma, delta_pow, omega, first, stream):
        self.O.oracle_permutation_terms(num, den, value, sigma, n, _a(beta), _a(gamma), _a(delta_pow), _a(omega), first)
        return 0

    def h2_dev_permutation_sigma(self, out, map_col, map_row, n, delta, omega, stream):
        self.O.oracle_permutation_sigma(out, map_col, map_row, n, _a(delta), _a(omega))
        return 0

    def h2_dev_logup_multiplicity(self, table, ptrs, n_inputs, usable, n, m, scratch, nbytes, stream):
        return 1 if self.O.oracle_logup_multiplicity(table, _a(ptrs), n_inputs, usable, n, m) else 0

    def h2_dev_evaluate_h(self, desc, out, stream):
        self.O.oracle_evaluate_h(desc, out)
        return 0

    def h2_dev_random_fr(self, key, n, out, stream):
        step = 1 << 18                                      # keystream of 2^18 blocks at a time: 16 MiB
        for first in range(0, n, step):
            cnt = min(step, n - first)
            words = chacha20_blocks(key, cnt, first=first)
            self.O.oracle_reduce_wide_253(words.ctypes.data, cnt, out + 32 * first)
        return 0


class _Null:
    """a stream / an event of a device that has neither"""
    cuda_stream = 0

    def __init__(self, *a, **k):
        pass

    def record(self, *a):
        pass

    def wait_event(self, *a):
        pass

    def synchronize(self):
        pass


class _Cuda:
    Stream = Event = _Null

    @staticmethod
    def stream(_):
        return contextlib.nullcontext()

    @staticmethod
    def set_device(_):
        pass


class _Torch:
    """torch with the stream plumbing of a device taken out"""
    cuda = _Cuda

    def __init__(self, torch):
        self._torch = torch

    def __getattr__(self, name):
        return getattr(self._torch, name)


class OracleDevice(P.Device):
    """prover.Device over host memory and the oracle (see the module docstring)"""

    def __init__(self, threads=None, eval_cache=None, force_cosets=False):
        import torch

        self.torch = _Torch(torch)
        self.dev = torch.device("cpu")
        self.L = OracleLib(threads)
        self.tstream = self.copy_stream = _Null()
        self.stream = None
        self._scratch, self._pinned = None, {}
        self.group, self.group_size, self.group_rank, self.force_collective = None, 1, 0, False
        self.force_cosets, self.mem_budget, self.eval_cache = force_cosets, None, eval_cache

    def upload(self, a):
        # a copy, as a transfer to a device is: the prover converts and blinds its columns in place
        a = np.ascontiguousarray(a)
        if a.ndim == 1:                          # a compact column: limb 0 only
            wide = np.zeros((a.shape[0], 4), dtype=np.uint64)
            wide[:, 0] = a
            a = wide
        return self.torch.from_numpy(a.copy().view(np.int64))

    def pinned_columns(self, count, n):
        return [np.zeros((n, 4), dtype=np.uint64) for _ in range(count)]

    def residency(self, cs, dom, instances=1):
        if self.eval_cache is not None:
            return "cosets", max(0, min(dom.quotient_poly_degree, self.eval_cache))
        return "extended", None

    def msm_async(self, scalars, bases, n, max_bits=254):
        import concurrent.futures

        fut = concurrent.futures.Future()
        fut.set_result(self.msm(scalars, bases, n, max_bits))
        return fut

    def intt_on_side_stream(self, cols, dom, extend=False):
        out = [self.intt(self.clone(t), dom) for t in cols]
        ext = [self.coeff_to_extended(c, dom) for c in out] if extend else None
        return out, ext, _Null()


def params_like(device, params):
    """the SRS of `params` (tables of a HIP device) for the oracle device: the same points, host memory"""
    d = OracleDevice() if device is None else device
    return P.Params(d, params.k, params.g.cpu().numpy(), params.g_lagrange.cpu().numpy(), tables=False)


def keygen(device, params, cs, fixed, copies, **kw):
    """prover.keygen over the oracle; the gate program stays with the oracle's interpreter (no generated kernel)"""
    prev = os.environ.get("H2_EVALH_JIT")
    os.environ["H2_EVALH_JIT"] = "0"
    try:
        pk = P.keygen(device, params, cs, fixed, copies, **kw)
    finally:
        if prev is None:
            del os.environ["H2_EVALH_JIT"]
        else:
            os.environ["H2_EVALH_JIT"] = prev
    pk.evalh_stats = None
    return pk
