"""create_proof as the LITERAL drop-in executes it: every polynomial lives in HOST memory (the reference's `Vec<F>`,
poly.rs:33-64; `Params { g, g_lagrange }`, poly/commitment.rs:23-29) and every vector operation is one call of a
host-slice entry point of include/halo2_hip.h -- the calls `integration/hip.rs` binds for `--features hip`:
`h2_ntt` / `h2_intt` (arithmetic.rs:495-534), `h2_msm` over a registered SRS (arithmetic.rs:334-367 + `register_params`),
`h2_coeff_to_extended` / `h2_extended_to_coeff` / `h2_divide_by_vanishing_poly` (poly/domain.rs:270-373),
`h2_evaluate_h_coeff` (plonk/evaluation.rs:1229-1241: coefficient forms in, extended values out), `h2_evaluate_h` with
y := theta (evaluate_with_theta, :2330-2398), `h2_lincomb` (gwc/prover.rs:57-151) -- i.e. one PCIe round trip per call.

This is a MEASUREMENT device (bench.py: `create_proof.host_slice_api`; INTEGRATION.md states the ratio to the
device-resident flow), built from product code only: there is no CPU arithmetic in it.  The passes the reference leaves
to rayon between its GPU calls also go through host-slice entry points: whole steps where the library has them
(`h2_permutation_product`, `h2_msm_intt`, `h2_quotient_poly_coeff`, `h2_eval_polynomial_batch`, `h2_quotient_sum`; the
step-by-step forms `h2_permutation_terms`, `h2_batch_invert`, `h2_prefix_product`, `h2_eval_polynomial`, `h2_kate_division`,
`h2_eval_op`, `h2_lincomb` with `fused_permutation=False`), single passes otherwise (`h2_prefix_sum`,
`h2_logup_multiplicity`, `h2_permutation_sigma`, `h2_distribute_powers`, `h2_random_fr`).  Every call is counted in
`HostSliceLib.calls`; nothing is staged through device tensors by hand any more (`STAGED` is empty).  The proof bytes are
those of the resident prover.
"""
import ctypes

import numpy as np

from . import prover as P
from ._lib import check, lib

_vp = ctypes.c_void_p
STAGED = ()


def _addr(x):
    if x is None or isinstance(x, int):
        return x
    if isinstance(x, (ctypes.Array, ctypes.Structure)):
        return ctypes.addressof(x)
    return x


def _bytes_at(addr, nbytes):
    """a numpy view of host memory"""
    return np.ctypeslib.as_array((ctypes.c_uint8 * nbytes).from_address(addr))


class _TimedLib:
    """the C library with a stopwatch on it: `busy_seconds` is the wall time during which at least one of its entry points was
    executing (calls of the worker threads overlap: the union, not the sum), `summed_seconds` the sum over calls -- what is left
    of a proof's wall time is the host's own handling of its vectors (allocating, first-touching and freeing them)"""

    def __init__(self, inner):
        import threading

        self._inner, self._mu = inner, threading.Lock()
        self._active, self._since = 0, 0.0
        self.busy_seconds = self.summed_seconds = 0.0
        self.by_call = {}

    def reset(self):
        with self._mu:
            self.busy_seconds = self.summed_seconds = 0.0
            self.by_call = {}

    def __getattr__(self, name):
        import time

        fn = getattr(self._inner, name)
        if not callable(fn):
            return fn

        def timed(*args):
            t0 = time.perf_counter()
            with self._mu:
                if self._active == 0:
                    self._since = t0
                self._active += 1
            try:
                return fn(*args)
            finally:
                t1 = time.perf_counter()
                with self._mu:
                    self._active -= 1
                    self.summed_seconds += t1 - t0
                    self.by_call[name] = self.by_call.get(name, 0.0) + (t1 - t0)
                    if self._active == 0:
                        self.busy_seconds += t1 - self._since

        self.__dict__[name] = timed
        return timed


class HostSliceLib:
    """the entry points prover.py calls (h2_dev_* signatures), each forwarded to the host-slice entry point the Rust patch
    binds; pointers are host addresses"""

    def __init__(self, torch, dev, workers=4):
        import concurrent.futures

        self.R = _TimedLib(lib())
        self.torch, self.dev = torch, dev
        self.calls = {}
        # the reference runs its per-column loops as rayon par_iters (commitments plonk/prover.rs:293-299, inverse transforms
        # :643-646, evaluations :731-737): the calls of such a loop are issued from a few threads here too, and the library's
        # two host-API slots per device overlap one call's transfers with another's kernels
        self.pool = concurrent.futures.ThreadPoolExecutor(max_workers=workers) if workers > 1 else None

    def _each(self, fn, items):
        """fn(item) for every item, from the worker threads when there are any; the first non-zero status wins"""
        if self.pool is None or len(items) < 2:
            rcs = [fn(it) for it in items]
        else:
            rcs = list(self.pool.map(fn, items))
        return next((rc for rc in rcs if rc), 0)

    def _count(self, name):
        self.calls[name] = self.calls.get(name, 0) + 1      # (dict updates of the worker threads: GIL-atomic enough for a tally)

    # -- transforms ----------------------------------------------------------------------------------------------------
    def h2_dev_ntt(self, a, tmp, omega, k, stream):
        self._count("h2_ntt")
        return self.R.h2_ntt(a, _addr(omega), k)

    def h2_dev_intt(self, a, tmp, omega_inv, divisor, k, stream):
        self._count("h2_intt")
        return self.R.h2_intt(a, _addr(omega_inv), _addr(divisor), k)

    def h2_dev_intt_batch(self, ptrs, count, tmp, omega_inv, divisor, k, stream):
        return self._each(lambda i: self.h2_dev_intt(ptrs[i], None, omega_inv, divisor, k, stream), list(range(count)))

    def h2_dev_coeff_to_extended_batch(self, srcs, dsts, count, tmp, k, ek, g, g_inv, ext_omega, stream):
        return self._each(lambda i: self.h2_dev_coeff_to_extended(srcs[i], dsts[i], None, k, ek, g, g_inv, ext_omega, stream),
                          list(range(count)))

    def h2_dev_coeff_to_extended(self, a, out, tmp, k, ek, g, g_inv, ext_omega, stream):
        self._count("h2_coeff_to_extended")
        return self.R.h2_coeff_to_extended(a, out, k, ek, _addr(g), _addr(g_inv), _addr(ext_omega))

    def h2_dev_extended_to_coeff(self, a, tmp, ek, g, g_inv, ext_omega_inv, ext_divisor, stream):
        self._count("h2_extended_to_coeff")
        return self.R.h2_extended_to_coeff(a, a, 1 << ek, ek, _addr(g), _addr(g_inv), _addr(ext_omega_inv), _addr(ext_divisor))

    def h2_dev_divide_by_vanishing_poly(self, a, size, t_evals, t_len, stream):
        self._count("h2_divide_by_vanishing_poly")
        return self.R.h2_divide_by_vanishing_poly(a, size, t_evals, t_len)

    def h2_dev_distribute_powers(self, a, n, g, stream):
        self._count("h2_distribute_powers")
        return self.R.h2_distribute_powers(a, n, _addr(g))

    # -- commitments: the SRS is registered once (Params below), each call ships the scalars ----------------------------
    def h2_msm_scratch_bytes(self, n, bits):
        return 256

    def h2_msm_batch_scratch_bytes(self, n, bits, count):
        return 512

    def h2_logup_scratch_bytes(self, n):
        return self.R.h2_logup_scratch_bytes(n)

    def h2_dev_msm(self, scalars, bases, n, max_bits, scratch, nbytes, out, stream):
        self._count("h2_msm")
        return self.R.h2_msm(scalars, bases, n, max_bits, _addr(out))

    def h2_dev_msm_batch_ex(self, sp, bp, bits, count, n, scratch, nbytes, out, stream):
        return self._each(lambda i: self.h2_dev_msm(sp[i], bp[i], n, bits[i], None, 0, _addr(out) + 96 * i, stream),
                          list(range(count)))

    def h2_dev_bases_precompute_bytes(self, n, digits):
        return 0

    def h2_dev_bases_precompute(self, bases, n, digits, stream):
        return 0

    def h2_dev_bases_forget(self, bases):
        return 0

    def h2_set_table_budget(self, nbytes):
        return self.R.h2_set_table_budget(nbytes)

    # -- Montgomery form, elementwise, scans, Horner, division ---------------------------------------------------------
    def h2_dev_batch_mont(self, a, n, stream):
        self._count("h2_batch_mont")
        return self.R.h2_batch_mont(a, n)

    def h2_dev_batch_unmont(self, a, n, stream):
        self._count("h2_batch_unmont")
        return self.R.h2_batch_unmont(a, n)

    def h2_dev_eval_op(self, op, res, l, r, l_rot, r_rot, size, c, stream):
        self._count("h2_eval_op")
        return self.R.h2_eval_op(op, res, l, r, l_rot, r_rot, size, _addr(c))

    def h2_dev_lincomb(self, res, ptrs, coeffs, count, size, stream):
        self._count("h2_lincomb")
        return self.R.h2_lincomb(res, ptrs, coeffs, count, size)

    def h2_dev_eval_polynomial(self, poly, n, point, out, stream):
        self._count("h2_eval_polynomial")
        return self.R.h2_eval_polynomial(poly, n, _addr(point), _addr(out))

    def h2_dev_eval_polynomial_batch(self, ptrs, count, n, points, out, stream):
        self._count("h2_eval_polynomial_batch")
        return self.R.h2_eval_polynomial_batch(ptrs, count, n, points, out)

    def h2_dev_kate_division(self, a, n, b, q, stream):
        self._count("h2_kate_division")
        return self.R.h2_kate_division(a, n, _addr(b), q)

    def h2_dev_prefix_product(self, f, n, init, z, stream):
        self._count("h2_prefix_product")
        return self.R.h2_prefix_product(f, n, _addr(init), z)

    def h2_dev_batch_invert(self, a, tmp, n, stream):
        self._count("h2_batch_invert")
        return self.R.h2_batch_invert(a, n)

    def h2_dev_prefix_sum(self, f, n, init, z, stream):
        self._count("h2_prefix_sum")
        return self.R.h2_prefix_sum(f, n, _addr(init), z)

    def h2_dev_permutation_terms(self, num, den, value, sigma, n, beta, gamma, delta_pow, omega, first, stream):
        self._count("h2_permutation_terms")
        return self.R.h2_permutation_terms(num, den, value, sigma, n, _addr(beta), _addr(gamma), _addr(delta_pow), _addr(omega), first)

    def h2_dev_permutation_sigma(self, out, map_col, map_row, n, delta, omega, stream):
        self._count("h2_permutation_sigma")
        return self.R.h2_permutation_sigma(out, map_col, map_row, n, _addr(delta), _addr(omega))

    def h2_dev_logup_multiplicity(self, table, ptrs, n_inputs, usable, n, m, scratch, nbytes, stream):
        self._count("h2_logup_multiplicity")
        return self.R.h2_logup_multiplicity(table, ptrs, n_inputs, usable, n, m, None)

    def h2_dev_logup_multiplicity_bits(self, table, ptrs, n_inputs, usable, n, m, scratch, nbytes, bits_out, stream):
        self._count("h2_logup_multiplicity")
        return self.R.h2_logup_multiplicity(table, ptrs, n_inputs, usable, n, m, bits_out)

    def h2_dev_random_fr(self, key, n, out, stream):
        self._count("h2_random_fr")
        return self.R.h2_random_fr(key, n, out)

    # -- the quotient numerator ---------------------------------------------------------------------------------------
    def h2_dev_evaluate_h(self, desc, out, stream):
        self._count("h2_evaluate_h")
        return self.R.h2_evaluate_h(desc, out)

    def h2_evaluate_h_coeff(self, desc, out):
        self._count("h2_evaluate_h_coeff")
        return self.R.h2_evaluate_h_coeff(desc, out)


class _NullStream:
    cuda_stream = 0

    def __init__(self, *a, **k):
        pass

    def record(self, *a):
        pass

    def wait_event(self, *a):
        pass

    def synchronize(self):
        pass


class _HostCuda:
    """the stream plumbing of prover.Device with host tensors: nothing is asynchronous (every entry point returns with
    its result in host memory)"""
    Stream = Event = _NullStream

    def __init__(self, torch):
        self._torch = torch

    def stream(self, _):
        import contextlib

        return contextlib.nullcontext()

    def set_device(self, _):
        pass

    def __getattr__(self, name):
        return getattr(self._torch.cuda, name)


class _HostTorch:
    def __init__(self, torch):
        self._torch = torch
        self.cuda = _HostCuda(torch)

    def __getattr__(self, name):
        return getattr(self._torch, name)


class HostApiDevice(P.Device):
    """prover.Device whose vectors are host tensors and whose library is `HostSliceLib` (see the module docstring).
    `quotient_from_coeffs`: create_proof_ext hands the evaluator COEFFICIENT forms and makes one h2_evaluate_h_coeff
    call per circuit instance, as `Evaluator::evaluate_h` under the cuda / hip feature does."""
    quotient_from_coeffs = True

    def __init__(self, device=0, pinned=False, workers=4, register_polys=True, fused_permutation=True):
        """`pinned`: every host vector lives in page-locked memory (a Rust-side allocator over h2_host_alloc_pinned for
        `Polynomial::values`): the same calls, but their transfers are DMA instead of staged pageable copies.  `workers`:
        threads that issue the calls of a per-column loop (the reference's rayon par_iters); 1 = strictly sequential"""
        self.pinned = pinned
        self.workers = workers
        self.register_polys, self._retained = register_polys, []     # register_polys=False: the data flow of rounds 4-5
        if not fused_permutation:
            self.permutation_product = None                          # the permutation products step by step (rounds 4-6)
            self.commit_lagrange_and_ifft = None                     # ... and h2_msm + h2_intt for the product columns
            self.quotient_sum = None                                 # ... and the multiopen's folds and divisions one by one
            self.quotient_poly_coeff = None                          # ... and evaluate_h / divide / extended_to_coeff as three calls
            self.logup_grand_sum = None                              # ... and the lookups' terms, inversion and scans one by one
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("the host-slice API needs a HIP device behind it: there is no CPU path")
        self.gpu = torch.device("cuda", device)
        torch.cuda.set_device(self.gpu)
        self.torch = _HostTorch(torch)
        self.dev = torch.device("cpu")
        self.L = HostSliceLib(torch, self.gpu, self.workers)
        self.tstream = self.copy_stream = _NullStream()
        self.stream = None
        self._scratch, self._pinned = None, {}
        self.group, self.group_size, self.group_rank, self.force_collective = None, 1, 0, False
        self.force_cosets, self.mem_budget, self.eval_cache = False, None, None

    def _pin(self, t):
        return t.pin_memory() if self.pinned else t

    def empty(self, n):
        return self.torch.empty((n, 4), dtype=self.torch.int64, pin_memory=self.pinned)

    def zeros(self, n):
        return self.torch.zeros((n, 4), dtype=self.torch.int64, pin_memory=self.pinned)

    def clone(self, t):
        out = self.torch.empty(t.shape, dtype=t.dtype, pin_memory=self.pinned)
        out.copy_(t)
        return out

    def upload(self, a):
        a = np.ascontiguousarray(a)
        if a.ndim == 1:
            wide = np.zeros((a.shape[0], 4), dtype=np.uint64)
            wide[:, 0] = a
            a = wide
        return self.clone(self.torch.from_numpy(a.view(np.int64)))

    def upload_async(self, a):
        return (self.clone(a) if self.torch.is_tensor(a) else self.upload(a)), None

    def widen(self, small, stream=None):
        out = self.zeros(small.shape[0])
        out[:, 0] = small
        return out

    def pinned_columns(self, count, n, compact=False):
        return [np.zeros((n,) if compact else (n, 4), dtype=np.uint64) for _ in range(count)]

    def max_scalar_bits_many(self, cols, n):
        """find_max_scalar_bits (plonk/prover.rs:237-254: a rayon fold over the column there) per column, each column cut into
        row ranges for the worker threads (numpy's reductions release the interpreter lock)"""
        parts = self.workers if self.L.pool is not None and n >= (1 << 16) else 1
        step = (n + parts - 1) // parts
        # (a job carries an address, not a tensor: a pool thread keeps its last job alive until the next one arrives, and a
        # page-locked block that cannot go back to the allocator's cache costs the next proof a fresh hipHostMalloc)
        jobs = [(ci, c.data_ptr() + 32 * lo, min(step, n - lo)) for ci, c in enumerate(cols) for lo in range(0, n, step)]
        one = lambda job: (job[0], P.max_scalar_bits(_bytes_at(job[1], 32 * job[2]).view(np.uint64).reshape(-1, 4)))       # noqa: E731
        done = [one(j) for j in jobs] if parts == 1 else list(self.L.pool.map(one, jobs))
        bits = [0] * len(cols)
        for ci, b in done:
            bits[ci] = max(bits[ci], b)
        return bits

    def residency(self, cs, dom, instances=1):
        return "extended", None

    def msm_async(self, scalars, bases, n, max_bits=254):
        import concurrent.futures

        fut = concurrent.futures.Future()
        fut.set_result(self.msm(scalars, bases, n, max_bits))
        return fut

    def intt(self, t, dom):
        """lagrange_to_coeff in place (poly/domain.rs:233-266): one h2_intt call (no scratch vector on the host side)"""
        from .prover import _fr

        check(self.L.h2_dev_intt(t.data_ptr(), None, _fr(dom.omega_inv), _fr(dom.ifft_divisor), dom.k, None), "h2_intt")
        return t

    def intt_to(self, t, dom):
        """the coefficient form of a column whose values are kept: h2_intt_to reads `t` and writes a new vector (the reference
        clones the column and transforms the clone in place, plonk/prover.rs:643-646)"""
        from .prover import _fr

        out = self.empty(dom.n)
        wi, dv = _fr(dom.omega_inv), _fr(dom.ifft_divisor)
        self.L._count("h2_intt_to")
        check(self.L.R.h2_intt_to(t.data_ptr(), out.data_ptr(), _addr(wi), _addr(dv), dom.k), "h2_intt_to")
        return out

    def intt_on_side_stream(self, cols, dom, extend=False):
        ptrs = [(t.data_ptr(), self.empty(dom.n)) for t in cols]                 # (addresses for the worker threads)
        wi, dv = P._fr(dom.omega_inv), P._fr(dom.ifft_divisor)

        def one(job):
            self.L._count("h2_intt_to")
            return self.L.R.h2_intt_to(job[0], job[1].data_ptr(), _addr(wi), _addr(dv), dom.k)

        check(self.L._each(lambda i: one(ptrs[i]), list(range(len(ptrs)))), "h2_intt_to")
        return [o for _, o in ptrs], None, _NullStream()

    def sync(self):
        pass

    def permutation_product(self, values, sigmas, n, beta, gamma, delta_pow, omega, init, probe):
        """one grand-product column of the permutation argument by ONE host-slice call (h2_permutation_product: the set's
        value / sigma columns in, z out; permutation/prover.rs:72-165) -> (z, z[probe]).  `fused_permutation=False` at
        construction keeps the step-by-step calls (h2_permutation_terms, h2_batch_invert, h2_eval_op, h2_prefix_product)."""
        from .prover import DELTA, _fr

        z = self.empty(n)
        vp = (_vp * len(values))(*[t.data_ptr() for t in values])
        sp = (_vp * len(sigmas))(*[t.data_ptr() for t in sigmas])
        self.L._count("h2_permutation_product")
        scalars = [_fr(v) for v in (beta, gamma, delta_pow, DELTA, omega, init)]       # (alive until the call has returned)
        check(self.L.R.h2_permutation_product(z.data_ptr(), vp, sp, len(values), n, *[_addr(v) for v in scalars]), "h2_permutation_product")
        return z, self.get_rows(z, probe, 1)[0]

    def logup_grand_sum(self, inputs, table, m, n, beta, init, probe):
        """one grand-sum column of a logup lookup by ONE host-slice call (h2_logup_grand_sum: the set's compressed inputs -- and,
        for the first set, the table and its multiplicities -- in, z out; plonk/logup/prover.rs:243-347) -> (z, z[probe])"""
        from .prover import _fr

        z = self.empty(n)
        ip = (_vp * len(inputs))(*[t.data_ptr() for t in inputs])
        scalars = [_fr(beta), _fr(init)]
        self.L._count("h2_logup_grand_sum")
        check(self.L.R.h2_logup_grand_sum(z.data_ptr(), ip, len(inputs), table.data_ptr() if table is not None else None,
                                          m.data_ptr() if m is not None else None, n, *[_addr(v) for v in scalars]), "h2_logup_grand_sum")
        return z, self.get_rows(z, probe, 1)[0]

    def quotient_poly_coeff(self, desc, dom, t_evaluations):
        """h(X) in coefficient form by ONE host-slice call (h2_quotient_poly_coeff: evaluate_h from coefficient forms, the
        division by the vanishing polynomial and extended_to_coeff on the device; only n * quotient_poly_degree coefficients
        cross PCIe)"""
        from .prover import _fr

        out_len = dom.n * dom.quotient_poly_degree
        out = self.empty(out_len)
        scalars = [_fr(v) for v in (dom.g_coset, dom.g_coset_inv, dom.extended_omega_inv, dom.extended_ifft_divisor)]
        self.L._count("h2_quotient_poly_coeff")
        check(self.L.R.h2_quotient_poly_coeff(ctypes.byref(desc), t_evaluations.data_ptr(), len(dom.t_evaluations),
                                              *[_addr(v) for v in scalars], out.data_ptr(), out_len), "h2_quotient_poly_coeff")
        return out

    def quotient_sum(self, n, sets, remainders=False):
        """sum over `sets` = [(polys, coeffs, low, points)] of (sum_i coeffs[i] polys[i] - low) / prod_j (X - points[j]) by ONE
        host-slice call (h2_quotient_sum; poly/multiopen/shplonk/prover.rs:95-153, :205-219) -> (the n coefficients, the
        remainders of the divisions as integers or None)"""
        from .prover import R_MOD, fr_from_mont_limbs, fr_to_mont_limbs

        def flat(values):
            return np.array([fr_to_mont_limbs(v % R_MOD) for v in values], dtype=np.uint64).reshape(-1, 4)

        sz = ctypes.c_size_t
        counts = (sz * len(sets))(*[len(p) for p, _, _, _ in sets])
        lows = (sz * len(sets))(*[len(lo) for _, _, lo, _ in sets])
        pts = (sz * len(sets))(*[len(pt) for _, _, _, pt in sets])
        tensors = [t for p, _, _, _ in sets for t in p]
        ptrs = (_vp * len(tensors))(*[t.data_ptr() for t in tensors])
        coeffs = flat([c for _, cs, _, _ in sets for c in cs])
        low = flat([c for _, _, lo, _ in sets for c in lo])
        points = flat([c for _, _, _, pt in sets for c in pt])
        rem = np.zeros((max(len(points), 1), 4), dtype=np.uint64) if remainders else None
        out = self.empty(n)
        self.L._count("h2_quotient_sum")
        check(self.L.R.h2_quotient_sum(out.data_ptr(), n, len(sets), counts, ptrs, coeffs.ctypes.data, lows, low.ctypes.data if len(low) else None,
                                       pts, points.ctypes.data if len(points) else None, rem.ctypes.data if remainders else None),
              "h2_quotient_sum")
        return out, ([fr_from_mont_limbs(r) for r in rem[:len(points)]] if remainders else None)

    def commit_lagrange_and_ifft(self, cols, bases, dom):
        """Params::commit_lagrange_and_ifft (poly/commitment.rs:144-197 -> gpu_multiexp_bound_and_fft, arithmetic.rs:375-410):
        one h2_msm_intt call per column -- the commitment over `bases` and, sharing the one upload, the column taken to its
        coefficient form in place.  -> the commitments (affine), as msm_batch returns them"""
        from .prover import _fr, jacobians_to_affine

        if not cols:
            return []
        out = np.zeros((len(cols), 12), dtype=np.uint64)
        wi, dv = _fr(dom.omega_inv), _fr(dom.ifft_divisor)
        ptrs, bp, n_, k_ = [c.data_ptr() for c in cols], bases.data_ptr(), dom.n, dom.k     # (addresses: see max_scalar_bits_many)

        def one(i):
            self.L._count("h2_msm_intt")
            return self.L.R.h2_msm_intt(ptrs[i], bp, n_, 254, _addr(wi), _addr(dv), k_, out[i].ctypes.data)

        check(self.L._each(one, list(range(len(cols)))), "h2_msm_intt")
        return jacobians_to_affine(out)

    # -- h2_poly_register: what the Rust side does at the same points (integration/hip.rs register_polys / unregister_polys) --
    def retain(self, vectors, owner=None):
        """final host vectors -> registered with the library: every later host-slice call that READS one of them (the evaluator's
        columns, h2_eval_polynomial, h2_lincomb operands, h2_kate_division) uses a device copy uploaded once.  `owner`: kept
        for the owner's life (the proving key); otherwise until release_retained (the end of the proof)."""
        import weakref

        if not self.register_polys:
            return
        R = self.L.R
        ptrs = []
        for t in vectors:
            if t is None or t.numel() == 0:
                continue
            assert t.is_contiguous()
            check(R.h2_poly_register(t.data_ptr(), t.shape[0]), "h2_poly_register")
            ptrs.append(t.data_ptr())
        if owner is not None:
            weakref.finalize(owner, _unregister_polys, R, ptrs).atexit = False
        else:
            self._retained += ptrs

    def release_retained(self):
        _unregister_polys(self.L.R, self._retained)
        self._retained = []


def _unregister_polys(R, ptrs):
    for p in ptrs:
        R.h2_poly_unregister(p)


def _unregister(R, ptrs):
    for p in ptrs:
        R.h2_bases_unregister(p)


def params_like(device, params):
    """`params` (tables of a HIP device) as the reference's Params: host vectors, registered with the library once
    (crate::hip::register_params in the patch) -- one device copy + shifted-base table per process"""
    import weakref

    g, gl = device._pin(params.g.cpu().contiguous()), device._pin(params.g_lagrange.cpu().contiguous())
    out = P.Params(device, params.k, g, gl, tables=False)
    R = device.L.R
    for t in (g, gl):
        check(R.h2_bases_register(t.data_ptr(), params.n), "h2_bases_register")
    weakref.finalize(out, _unregister, R, [g.data_ptr(), gl.data_ptr()]).atexit = False
    return out
