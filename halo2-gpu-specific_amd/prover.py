"""keygen + create_proof on the device-resident C ABI: the caller of the hot path.

Every polynomial lives in HBM from upload to the last opening; the host sees only what the protocol hashes
(commitments, evaluations) plus the handful of low coefficients SHPLONK adjusts.  Orchestration follows

  plonk/keygen.rs:330-440              keygen_pk: fixed polys, l0 / l_last / l_active_row, permutation pk
  plonk/permutation/keygen.rs:112-261  cycle -> mapping -> sigma polynomials
  plonk/prover.rs:206-850              create_proof_ext (advice, challenges, permutation products, vanishing
                                       argument, evaluations, multiopen)
  plonk/permutation/prover.rs:47-330   commit / evaluate / open
  plonk/vanishing/prover.rs:40-160     random poly, h pieces, h(x)
  poly/multiopen/shplonk.rs:58-135, shplonk/prover.rs:89-225

All arithmetic on vectors runs in libhalo2_hip.so (`h2_dev_*`); torch only owns the device buffers and the
stream.  There is no CPU path in this file: without the library or a GPU it raises.
"""
import ctypes
import hashlib
import os

import numpy as np

from . import evaluation as ev
from ._lib import H2Error, check, lib
from .circuit import compile_compress, compile_evaluator
from .transcript import (Blake2bWrite, R_MOD, fr_from_mont_limbs, fr_to_mont_limbs, g1_add_affine, jacobian_to_affine, jacobians_to_affine,
                         point_to_bytes)

ROOT_OF_UNITY = 0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C
DELTA = 0x09226B6E22C6F0CA64EC26AAD4C86E715B5F898E5E963F25870E56BBE533E9A2
ZETA = 0x30644E72E131A029048B6E193FD84104CC37A73FEC2BC5E9B8CA0B2D36636F23
S = 28
_vp = ctypes.c_void_p
_FR = ctypes.c_uint64 * 4


# H2_POISON_EMPTY=1 (tests): every "uninitialised" vector starts as all-ones words -- not even a field element -- so a pass
# that reads rows nobody wrote (a partially valid vector of the multi-rank paths: only a row range, a row slice and its halo)
# changes the proof on every run instead of only when the allocator hands back dirty memory
POISON_EMPTY = os.environ.get("H2_POISON_EMPTY") == "1"
TRACE_TRANSCRIPT = os.environ.get("H2_TRACE_TRANSCRIPT") == "1"     # create_proof prints a hash of the proof stream per phase


def _fr(v):
    """canonical integer -> Montgomery limbs for the C ABI"""
    return _FR(*fr_to_mont_limbs(v % R_MOD))


def _inv(v):
    return pow(v, -1, R_MOD)


class Domain:
    """EvaluationDomain::new (poly/domain.rs:44-149) -- the scalars only"""

    def __init__(self, k, degree):
        self.k, self.n = k, 1 << k
        self.quotient_poly_degree = degree - 1
        ek = k
        while (1 << ek) < self.n * self.quotient_poly_degree:
            ek += 1
        self.extended_k, self.extended_n = ek, 1 << ek
        self.extended_omega = pow(ROOT_OF_UNITY, 1 << (S - ek), R_MOD)
        self.omega = pow(self.extended_omega, 1 << (ek - k), R_MOD)
        self.omega_inv, self.extended_omega_inv = _inv(self.omega), _inv(self.extended_omega)
        self.ifft_divisor, self.extended_ifft_divisor = _inv(self.n), _inv(self.extended_n)
        self.g_coset, self.g_coset_inv = ZETA, ZETA * ZETA % R_MOD
        # t_evaluations: 1 / (ZETA^n * extended_omega^(n*i) - 1), i < 2^(extended_k - k)  (:91-131)
        t_len = 1 << (ek - k)
        zn, wn = pow(ZETA, self.n, R_MOD), pow(self.extended_omega, self.n, R_MOD)
        self.t_evaluations = [_inv((zn * pow(wn, i, R_MOD) - 1) % R_MOD) for i in range(t_len)]

    def rotate_omega(self, x, rot):
        return x * pow(self.omega if rot >= 0 else self.omega_inv, abs(rot), R_MOD) % R_MOD


def parse_bytes(text):
    """'12G', '512M', '4096' -> bytes; None / '' -> None"""
    if text is None or str(text).strip() == "":
        return None
    t = str(text).strip()
    mult = {"K": 1 << 10, "M": 1 << 20, "G": 1 << 30, "T": 1 << 40}.get(t[-1].upper())
    return int(float(t[:-1]) * mult) if mult else int(float(t))


def footprint(cs, dom, cached_cosets=None, instances=1):
    """Device bytes of polynomial data a proof of this circuit holds at its peak, by residency mode:
    'extended' = every extended coset resident (the proving key's fixed / sigma / l tables for the life of the key, the
    witness-dependent ones during the quotient phase); 'cosets' = coefficient forms only, the extended domain visited
    one coset of the n-th roots of unity at a time with `cached_cosets` sets of proving-key tables retained.
    `instances`: circuit instances proved together (plonk/prover.rs:206-232): every witness-dependent polynomial exists once
    per instance, the proving key's once.
    A planning estimate (it decides the mode against H2_DEVICE_MEM_BUDGET), not an allocator."""
    n, en = dom.n, dom.extended_n
    chunk = max(cs.degree() - 2, 1)
    nsets = (len(cs.perm_columns) + chunk - 1) // chunk
    F, A, I, P = cs.num_fixed, cs.num_advice, cs.num_instance, len(cs.perm_columns)
    lk_sets = sum(len(sets) for _, _, sets in cs.lookups)
    witness_polys = (A + I + nsets + lk_sets + len(cs.lookups) + len(cs.shuffles)) * max(1, instances)
    key_polys = 2 * (F + P)                                   # Lagrange values and coefficient forms
    c = dom.quotient_poly_degree
    cached = c if cached_cosets is None else max(1, min(c, cached_cosets))
    # + random / h / scratch vectors, the c per-coset quotients, and the scratch of two commitments in flight (sorted digit
    # entries, slice partials, buckets: ~370 B per point each = 23 n-vectors' worth; k = 25 measured: 97 GiB at the peak)
    base = 32 * n * (key_polys + 2 * witness_polys + 4 + c + 23)
    # (the extended cosets of the witness exist for one instance at a time: the quotient is evaluated circuit by circuit)
    return {"extended": base + 32 * en * ((F + P + 3) + witness_polys // max(1, instances) + 2),
            "cosets": base + 32 * n * ((F + P + 3) * cached + witness_polys // max(1, instances) + 3)}


class CosetTables:
    """The proving key's tables on single cosets of the extended domain (fixed / sigma columns, l0, l_last,
    l_active_row: n values each), built on demand from the coefficient forms and retained least-recently-used up to
    `keep` cosets -- this build's counterpart of the reference's extended-FFT cache (plonk/evaluation_gpu.rs:335-468,
    HALO2_PROOF_GPU_EVAL_CACHE): a miss costs (fixed + sigma + 3) n-point coset transforms (1.8 ms each at 2^24)."""

    def __init__(self, build, cosets, keep=None):
        self.build, self.cosets, self.keep = build, list(cosets), keep
        self.tabs, self.tick, self.hits, self.misses = {}, 0, 0, 0

    def __iter__(self):
        return iter(self.cosets)

    def __getitem__(self, j):
        hit = self.tabs.get(j)
        if hit is None:
            self.misses += 1
            if self.keep is not None:
                while self.tabs and len(self.tabs) >= max(self.keep, 1):
                    del self.tabs[min(self.tabs, key=lambda i: self.tabs[i][1])]
            hit = self.tabs[j] = [self.build(j), 0]
        else:
            self.hits += 1
        self.tick += 1
        hit[1] = self.tick
        return hit[0]

    def trim(self):
        """after a proof: keep = 0 retains nothing between proofs"""
        if self.keep is not None:
            while len(self.tabs) > self.keep:
                del self.tabs[min(self.tabs, key=lambda i: self.tabs[i][1])]


_PROCESS_GROUPS = {}


def _process_group_cache(key, make):
    """communicators made once per process (a Device per bench leg / per fuzzed circuit used to call
    dist.new_group() each, piling up RCCL communicators and turning Device() into an implicit collective every time); dropped
    when the default group they were made under is gone"""
    import torch.distributed as dist

    world = dist.get_world_size()
    alive = _PROCESS_GROUPS.get("_default")
    if alive is None or alive is not dist.group.WORLD:
        _PROCESS_GROUPS.clear()
        _PROCESS_GROUPS["_default"] = dist.group.WORLD
    if (key, world) not in _PROCESS_GROUPS:
        _PROCESS_GROUPS[(key, world)] = make()
    return _PROCESS_GROUPS[(key, world)]


_DEVICE_STREAMS = {}        # (GPU index, stream priorities) -> {"compute", "copy", "side"}: see Device.__init__


class Device:
    """Buffers (torch) + stream + thin typed wrappers over the h2_dev_* entry points."""

    def __init__(self, device=0, group=None, force_collective=False, force_cosets=False, mem_budget=None, eval_cache=None):
        """`mem_budget` (bytes; default H2_DEVICE_MEM_BUDGET, K / M / G suffixes; None = the device's memory): what the
        polynomial data of keygen + one proof may occupy.  A circuit whose extended cosets do not fit runs the
        extended-domain phase coset by coset from coefficient forms (`footprint`, `CosetTables`) -- the same proof bytes;
        `eval_cache` (default HALO2_PROOF_GPU_EVAL_CACHE, the reference's name) = how many cosets' worth of proving-key
        tables stay resident in that mode (setting it selects the mode; unset = as many as the budget holds).
        `group`: a torch.distributed process group (None = the default group when one is initialised) over which
        one proof is spread: every MSM is range-split over the ranks and the extended-domain phase is split by coset
        (DESIGN.md section 6); all ranks must then run the same proof on the same inputs.  `force_cosets` runs the
        coset path on a single device (all cosets locally): the same proof bytes by another route, for tests."""
        import torch  # plumbing only: device memory and the stream

        if not torch.cuda.is_available():
            raise RuntimeError("create_proof needs a HIP device: there is no CPU path")
        self.torch = torch
        self.dev = torch.device("cuda", device)
        torch.cuda.set_device(self.dev)
        self.L = lib()
        # H2_STREAM_PRIORITY = "<compute>,<side>" (torch / HIP stream priorities: lower = more urgent): experiment knob
        import os as _os

        pr = _os.environ.get("H2_STREAM_PRIORITY", "")
        self._prio = tuple(int(x) for x in pr.split(",")) if pr else (0, 0)
        # The streams belong to the PROCESS, not to the Device object: HIP multiplexes streams onto a few hardware queues
        # (GPU_MAX_HW_QUEUES) in creation order, and streams that share a queue serialise.  A process that made a Device per
        # workload kept drawing new streams, and which of them ended up sharing a queue depended on how many had been made before:
        # the 64-column proof at k = 22 ran 0.344 s as a process's first workload and 0.386 s after the k = 20 legs (its uploads
        # and commitments no longer overlapped: `advice commit` 170 -> 220 ms).  One set per (GPU, priorities), made once.
        streams = _DEVICE_STREAMS.setdefault((self.dev.index, self._prio), {})
        if not streams:
            streams["compute"] = torch.cuda.Stream(device=self.dev, priority=self._prio[0])
            streams["copy"] = torch.cuda.Stream(device=self.dev)
        self._streams = streams
        self.tstream, self.copy_stream = streams["compute"], streams["copy"]
        self.stream = _vp(self.tstream.cuda_stream)
        self._scratch = None
        self._pinned = {}
        self.group, self.group_size, self.group_rank, self.force_collective = group, 1, 0, force_collective
        import os

        self.force_cosets = force_cosets or os.environ.get("H2_FORCE_COSETS") == "1"   # experiment knob
        self.mem_budget = mem_budget if mem_budget is not None else parse_bytes(os.environ.get("H2_DEVICE_MEM_BUDGET"))
        if mem_budget is None and self.mem_budget is not None and not os.environ.get("H2_NTT_TABLE_BUDGET"):
            # a process-wide budget from the environment also bounds what the library keeps for itself (the optional
            # last-pass twiddle tables of the transforms): a sixteenth of it
            self.L.h2_set_table_budget(self.mem_budget // 16)
        env_cache = os.environ.get("HALO2_PROOF_GPU_EVAL_CACHE")
        self.eval_cache = eval_cache if eval_cache is not None else (int(env_cache) if env_cache not in (None, "") else None)
        import torch.distributed as dist

        self.bulk_group = group
        if dist.is_available() and dist.is_initialized():
            self.group_size, self.group_rank = dist.get_world_size(group), dist.get_rank(group)
            if self.group_size > 1 and group is None:
                # a second communicator for the bulk column traffic (parallel.broadcast_columns_begin): collectives of one
                # communicator run in issue order, and the 96-byte all-gathers of the commitments must not wait behind
                # half a gigabyte of coefficients.  ONE per process, made by the first Device of a multi-rank process
                # (collective: every rank constructs its first Device at the same point) and reused by every later one --
                # bench.py and the fuzzers build a Device per leg / per circuit, and communicators are never freed by torch
                # before destroy_process_group.
                self.bulk_group = _process_group_cache("bulk", dist.new_group)

    @classmethod
    def replica(cls, device=0, **kw):
        """a Device that proves ALONE inside a multi-rank process group -- one proof per GPU, no collective on the data path
        (the throughput form of N GPUs; `Device()` in such a process spreads ONE proof over the ranks).  Collective the first
        time: every rank builds the singleton groups of all ranks, in order."""
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return cls(device, **kw)
        groups = _process_group_cache("singletons", lambda: [dist.new_group(ranks=[r]) for r in range(dist.get_world_size())])
        return cls(device, group=groups[dist.get_rank()], **kw)

    # -- memory -----------------------------------------------------------------------------------------
    def empty(self, n):
        with self.torch.cuda.stream(self.tstream):
            t = self.torch.empty((n, 4), dtype=self.torch.int64, device=self.dev)
            if POISON_EMPTY:
                t.fill_(-1)
            return t

    def zeros(self, n):
        with self.torch.cuda.stream(self.tstream):
            return self.torch.zeros((n, 4), dtype=self.torch.int64, device=self.dev)

    def upload(self, a):
        a = np.ascontiguousarray(a)
        with self.torch.cuda.stream(self.tstream):
            src = self._pinned.get(a.ctypes.data)
            if src is not None and src.numel() == a.size:
                t = src.to(self.dev, non_blocking=True)
            else:
                import warnings

                with warnings.catch_warnings():      # read-only sources (a memory-mapped witness file) are only read
                    warnings.simplefilter("ignore", UserWarning)
                    t = self.torch.from_numpy(a.view(np.int64)).to(self.dev)
        return self.widen(t) if a.ndim == 1 else t

    def widen(self, small, stream=None):
        """a compact column (n u64 values on the device, 8 B per cell over PCIe) -> canonical (n, 4) scalars"""
        n = small.shape[0]
        out = self.empty(n)
        check(self.L.h2_dev_widen_u64(small.data_ptr(), n, out.data_ptr(), self.stream if stream is None else stream),
              "h2_dev_widen_u64")
        small.record_stream(self.tstream)
        return out

    def upload_async(self, a):
        """-> (device tensor, event or None): a pinned source is copied by DMA on the copy stream and the event marks
        its arrival; anything else goes through the synchronous path.  `a`: a canonical (n, 4) u64 column, a COMPACT column
        (1-D u64: every value below 2^64 -- 8 bytes per cell cross PCIe and the device widens them), or a device tensor
        (a witness that is already resident: copied, since the prover blinds and converts its columns in place)."""
        if self.torch.is_tensor(a):
            return self.clone(a), None
        a = np.ascontiguousarray(a)
        src = self._pinned.get(a.ctypes.data)
        if src is None or src.numel() != a.size:
            return self.upload(a), None
        torch = self.torch
        with torch.cuda.stream(self.copy_stream):
            t = src.to(self.dev, non_blocking=True)
            if a.ndim == 1:                    # widened on the copy stream too: behind its own arrival, ahead of nothing
                small, t = t, torch.empty((a.shape[0], 4), dtype=torch.int64, device=self.dev)
                check(self.L.h2_dev_widen_u64(small.data_ptr(), a.shape[0], t.data_ptr(), _vp(self.copy_stream.cuda_stream)),
                      "h2_dev_widen_u64")
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        t.record_stream(self.tstream)          # allocated under the copy stream, consumed on the compute stream
        return t, ev

    def download(self, t):
        with self.torch.cuda.stream(self.tstream):
            return t.cpu().numpy().view(np.uint64)

    def clone(self, t):
        with self.torch.cuda.stream(self.tstream):
            return t.clone()

    def set_rows(self, t, start, values):
        """t[start : start + len(values)] <- canonical integers (stored Montgomery)"""
        if not len(values):                      # (an instance column without public inputs)
            return
        a = np.array([fr_to_mont_limbs(v) for v in values], dtype=np.uint64)
        with self.torch.cuda.stream(self.tstream):
            t[start:start + len(values)] = self.torch.from_numpy(a.view(np.int64)).to(self.dev)

    def set_rows_many(self, items):
        """[(t, start, values)]: set_rows for several vectors with ONE host-to-device copy (the blinding rows of every
        product column of a phase: a copy from pageable memory synchronises, ~50 us each)"""
        items = [it for it in items if len(it[2])]
        if not items:
            return
        flat = np.array([fr_to_mont_limbs(v) for _, _, vals in items for v in vals], dtype=np.uint64)
        with self.torch.cuda.stream(self.tstream):
            blob = self.torch.from_numpy(flat.view(np.int64)).to(self.dev)
            at = 0
            for t, start, vals in items:
                t[start:start + len(vals)] = blob[at:at + len(vals)]
                at += len(vals)

    def set_rows_raw(self, t, start, small):
        """t[start : start + len(small)] <- small non-negative integers < 2^63, limb 0 only (no Montgomery form)"""
        a = np.zeros((len(small), 4), dtype=np.int64)
        a[:, 0] = small
        with self.torch.cuda.stream(self.tstream):
            t[start:start + len(small)] = self.torch.from_numpy(a).to(self.dev)

    def max_scalar_bits(self, t):
        """find_max_scalar_bits (plonk/prover.rs:237-254) of a canonical column resident on the device"""
        torch = self.torch
        sign = -(1 << 63)
        with torch.cuda.stream(self.tstream):
            # unsigned maxima of the four limbs: flip the sign bit so that the signed max orders them as unsigned
            m = ((t ^ sign).amax(dim=0) ^ sign).cpu().tolist()
        for limb in (3, 2, 1, 0):
            v = m[limb] & ((1 << 64) - 1)
            if v:
                return 64 * limb + v.bit_length()
        return 0

    def max_scalar_bits_many(self, cols, n):
        """find_max_scalar_bits of several canonical columns resident on the device: one launch, one synchronisation"""
        count = len(cols)
        if not count:
            return []
        ptrs = (ctypes.c_void_p * count)(*[c.data_ptr() for c in cols])
        out = (ctypes.c_uint32 * count)()
        with self.torch.cuda.stream(self.tstream):
            words = self.torch.empty((count, 8), dtype=self.torch.int32, device=self.dev)
        check(self.L.h2_dev_max_scalar_bits(ptrs, count, n, words.data_ptr(), out, self.stream), "h2_dev_max_scalar_bits")
        return list(out)

    def pinned_columns(self, count, n, compact=False):
        """`count` zeroed (n, 4) u64 numpy columns in page-locked host memory: a witness synthesised into them
        reaches the device by DMA instead of through the driver's staging copies.  `compact`: 1-D columns of n u64 values
        (a quarter of the bytes over PCIe) for columns whose values all fit 64 bits."""
        out = []
        for _ in range(count):
            t = self.torch.zeros((n,) if compact else (n, 4), dtype=self.torch.int64).pin_memory()
            a = t.numpy().view(np.uint64)
            self._pinned[a.ctypes.data] = t
            out.append(a)
        return out

    def get_rows(self, t, start, count):
        with self.torch.cuda.stream(self.tstream):
            a = t[start:start + count].cpu().numpy().view(np.uint64)
        return [fr_from_mont_limbs(r) for r in a]

    def sync(self):
        self.tstream.synchronize()

    # -- vectors that will not change any more (hooks for the host-slice data flow; nothing to do for resident vectors) ----------
    def retain(self, vectors, owner=None):
        """`vectors` are final from here on -- the proving key's coefficient forms (owner = the key), or a proof's advice /
        product / quotient polynomials (until `release_retained`).  On this device they already live in HBM; the literal
        drop-in (host_api.HostApiDevice) registers them with the library (h2_poly_register) so that later calls that read them
        find a device copy instead of uploading them again."""

    def release_retained(self):
        """end of a proof: the per-proof registrations of `retain` go"""

    def scratch(self, nbytes):
        if self._scratch is None or self._scratch.numel() < nbytes:
            self._scratch = None
            with self.torch.cuda.stream(self.tstream):
                self._scratch = self.torch.empty(nbytes, dtype=self.torch.uint8, device=self.dev)
        return self._scratch

    # -- coset sharding of the extended domain (one proof over several ranks) --------------------------
    def coset_plan(self, dom):
        """None = the quotient is evaluated on the whole extended domain; else (c, shards, owned): it is evaluated coset
        by coset.  The extended domain is the union of 2^(extended_k - k) cosets g_j H (g_j = zeta extended_omega^j, H =
        the n-th roots of unity; extended index 2^(extended_k - k) i + j is point i of coset j), but the quotient has
        only quotient_poly_degree = degree - 1 pieces h_t of n coefficients, and on coset j it reads
        P_j(X) = sum_t gamma_j^t h_t(X) (gamma_j = g_j^n): its values on the FIRST c = quotient_poly_degree cosets
        determine it (a c x c Vandermonde system per coefficient).  This rank evaluates the cosets in `owned`.

        Used when one proof is spread over several ranks.  On a single device the fused extended-domain transforms win
        (2^(extended_k - k) is the next power of two above degree - 1, so at best 5 of 8 cosets are saved, and the
        per-coset launches cost more than that below k = 22: profiles/r2_coset_path_vs_extended.txt); `force_cosets` /
        H2_COSETS=1 runs the coset route there -- the same proof bytes -- for tests."""
        from .parallel import coset_plan

        c = dom.quotient_poly_degree
        if self.group_size <= 1 and not self.force_cosets and os.environ.get("H2_COSETS") != "1":
            return None
        shards, owned = coset_plan(c, self.group_size, self.group_rank)
        return c, shards, owned

    def residency(self, cs, dom, instances=1):
        """('extended', None) or ('cosets', keep): how the extended-domain phase of this circuit runs on ONE device under
        the memory budget.  `keep` = cosets' worth of proving-key tables retained between uses.  `instances`: circuit
        instances in one proof (decided at keygen for one; create_proof_ext asks again when it is handed several)."""
        c = dom.quotient_poly_degree
        if self.eval_cache is not None:
            return "cosets", max(0, min(c, self.eval_cache))
        budget = self.mem_budget
        if budget is None:
            # what the device has left NOW -- behind the SRS and its shifted-base tables (96 GiB at k = 26), other keys --
            # plus what torch's allocator holds but does not use; a fifth of it stays free for library tables and slack
            torch = self.torch
            free, _ = torch.cuda.mem_get_info(self.dev)
            held = self._scratch.numel() if self._scratch is not None else 0     # commitment scratch: reused, and counted
            budget = int(0.8 * (free + torch.cuda.memory_reserved(self.dev) - torch.cuda.memory_allocated(self.dev) + held))
            if footprint(cs, dom, None, instances)["extended"] <= budget:
                return "extended", None
        elif footprint(cs, dom, None, instances)["extended"] <= budget:
            return "extended", None
        for keep in range(c, 0, -1):
            if footprint(cs, dom, keep, instances)["cosets"] <= budget:
                return "cosets", keep
        if footprint(cs, dom, 1, instances)["cosets"] > budget * 1.5:
            raise MemoryError("this circuit needs ~%.1f GiB of device memory even coset by coset; the budget is %.1f GiB"
                              % (footprint(cs, dom, 1, instances)["cosets"] / 2**30, budget / 2**30))
        return "cosets", 0

    def coeff_to_coset(self, poly, dom, j):
        """values of a coefficient vector (n entries) on coset j: a[t] *= g_j^t, then the n-point NTT -- coeff_to_extended
        (poly/domain.rs:270-287) restricted to the extended indices c i + j.  One fused transform (h2_dev_coset_ntt: the
        powers of g_j are applied on the first pass's load): `poly` is read, not copied"""
        out = self.empty(dom.n)
        g = dom.g_coset * pow(dom.extended_omega, j, R_MOD) % R_MOD
        tmp = self.empty(dom.n)
        check(self.L.h2_dev_coset_ntt(poly.data_ptr(), out.data_ptr(), tmp.data_ptr(), dom.k, _fr(g), _fr(dom.omega),
                                      self.stream), "h2_dev_coset_ntt")
        return out

    @staticmethod
    def _batch_width(points, cap_bytes=1 << 30):
        """vectors per batched-transform call: up to 16, fewer when their scratch (one vector each) would pass `cap_bytes` --
        large transforms fill the chip on their own, and the scratch must stay small next to a memory budget"""
        return max(1, min(16, cap_bytes // (32 * points)))

    def coeffs_to_coset(self, polys, dom, j):
        """coeff_to_coset of several coefficient vectors: up to 16 transforms per launch (h2_dev_coset_ntt_batch) -- the
        tiles of one 2^20-point pass do not fill the chip for long enough to hide their own latencies, those of a
        dozen vectors do"""
        count = len(polys)
        if count == 0:
            return []
        if count == 1:
            return [self.coeff_to_coset(polys[0], dom, j)]
        outs = [self.empty(dom.n) for _ in polys]
        width = self._batch_width(dom.n)
        tmp = self.empty(min(count, width) * dom.n)
        g = dom.g_coset * pow(dom.extended_omega, j, R_MOD) % R_MOD
        for at in range(0, count, width):
            part = min(width, count - at)
            src = (_vp * part)(*[p.data_ptr() for p in polys[at:at + part]])
            dst = (_vp * part)(*[o.data_ptr() for o in outs[at:at + part]])
            check(self.L.h2_dev_coset_ntt_batch(src, dst, part, tmp.data_ptr(), dom.k, _fr(g), _fr(dom.omega), self.stream),
                  "h2_dev_coset_ntt_batch")
        return outs

    def intt_many(self, ts, dom):
        """lagrange_to_coeff in place on several vectors, up to 16 per launch (h2_dev_intt_batch)"""
        count = len(ts)
        if count == 0:
            return ts
        if count == 1 or not hasattr(self.L, "h2_dev_intt_batch"):
            return [self.intt(t, dom) for t in ts]
        width = self._batch_width(dom.n)
        tmp = self.empty(min(count, width) * dom.n)
        for at in range(0, count, width):
            part = min(width, count - at)
            ptrs = (_vp * part)(*[t.data_ptr() for t in ts[at:at + part]])
            check(self.L.h2_dev_intt_batch(ptrs, part, tmp.data_ptr(), _fr(dom.omega_inv), _fr(dom.ifft_divisor), dom.k,
                                           self.stream), "h2_dev_intt_batch")
        return ts

    def column_owner(self, i):
        return i % self.group_size

    def coset_rank_group(self, c):
        """(group, members G, this rank's member index) of the ranks that share this rank's coset when the P ranks are a
        multiple G >= 2 of the c cosets (rank r evaluates coset r mod c; its group is {j, j + c, j + 2 c, ...}), else None.
        Creating the groups is collective: every rank creates all c of them, in order, the first time a circuit with c
        cosets is proved (cached)."""
        P = self.group_size
        if P <= 1 or c < 1 or P % c or P // c < 2 or self.group is not None or os.environ.get("H2_COSET_GROUPS") == "0":
            return None
        import torch.distributed as dist

        G = P // c
        # (per process, not per Device: the same c groups serve every Device of this world size)
        groups = _process_group_cache(("coset", P, c), lambda: [dist.new_group(ranks=[j + c * m for m in range(G)]) for j in range(c)])
        sub = (groups[self.group_rank % c], G, self.group_rank // c)
        self.__dict__.setdefault("_coset_groups", {})[c] = sub      # (what this Device used: the multi-rank tests look)
        return sub

    def coeffs_to_coset_rows(self, polys, dom, j, sub, halo):
        """the values of `polys` on coset j where THIS member of the coset's rank group needs them: member g transforms
        every G-th column (batched coset transforms) and the members exchange row slices (parallel.exchange_row_slices);
        returns full-size vectors valid on this member's rows and their halo"""
        from .parallel import exchange_row_slices

        group, G, g = sub
        count = len(polys)
        if count == 0:
            return []
        owners = [i % G for i in range(count)]
        done = self.coeffs_to_coset([p for p, o in zip(polys, owners) if o == g], dom, j)
        it = iter(done)
        cols = [next(it) if o == g else self.empty(dom.n) for o in owners]
        exchange_row_slices(cols, owners, dom.n, G, g, halo[0], halo[1], group=group, stream=self.tstream)
        return cols

    def intt_columns_begin(self, cols, dom, complete=True, keep=False):
        """lagrange_to_coeff of whole columns dealt round-robin over the ranks -- north_star's per-column NTT sharding
        (plonk/prover.rs:643-646 runs them as a par_iter) -- instead of every rank transforming every column: rank
        i mod P transforms column i and the coefficient vectors travel to the other ranks on the side stream
        (parallel.broadcast_columns_begin).  `complete` False: every rank holds only its row range of the columns (they
        were computed by ranges): the rows are gathered to the owner, not to everybody.  `keep`: the columns' Lagrange
        values stay as they are and the result is a new list.  Returns (coefficient columns, arrival): `arrival.wait()`
        before the compute stream reads them.  One device: plain transforms, arrival None."""
        n = dom.n
        if self.group_size <= 1:
            out = [self.clone(t) for t in cols] if keep else list(cols)
            return self.intt_many(out, dom), None
        from .parallel import broadcast_columns_begin, gather_rows_to

        owners = [self.column_owner(i) for i in range(len(cols))]
        lo, hi = self.row_range(n)
        if not complete and (lo, hi) != (0, n):
            for t, owner in zip(cols, owners):
                gather_rows_to(t, lo, hi, owner, group=self.group, stream=self.tstream)
        if keep:
            out = [self.clone(t) if owner == self.group_rank else self.empty(n) for t, owner in zip(cols, owners)]
        else:
            out = list(cols)
        self.intt_many([t for t, owner in zip(out, owners) if owner == self.group_rank], dom)
        arrival = broadcast_columns_begin(out, owners, group=self.bulk_group, stream=self.tstream, side=self.copy_stream)
        return out, arrival

    def coset_to_coeff(self, vals, dom, j):
        """in place: the polynomial of degree < n that takes the values `vals` on coset j (the inverse transform with
        1 / n and the powers of 1 / g_j fused into its last pass)"""
        tmp = self.empty(dom.n)
        g_inv = _inv(dom.g_coset * pow(dom.extended_omega, j, R_MOD) % R_MOD)
        check(self.L.h2_dev_coset_intt(vals.data_ptr(), tmp.data_ptr(), dom.k, _fr(g_inv), _fr(dom.omega_inv),
                                       _fr(dom.ifft_divisor), self.stream), "h2_dev_coset_intt")
        return vals

    # -- transforms -------------------------------------------------------------------------------------
    def intt(self, t, dom):
        """lagrange_to_coeff in place (poly/domain.rs:233-266)"""
        tmp = self.empty(dom.n)
        check(self.L.h2_dev_intt(t.data_ptr(), tmp.data_ptr(), _fr(dom.omega_inv), _fr(dom.ifft_divisor), dom.k,
                                 self.stream), "h2_dev_intt")
        return t

    def _side_stream(self):
        """the process's side stream of this GPU (see the streams' note in __init__)"""
        if "side" not in self._streams:
            self._streams["side"] = self.torch.cuda.Stream(device=self.dev, priority=self._prio[1])
        return self._streams["side"]

    def intt_on_side_stream(self, cols, dom, extend=False):
        """coefficient forms of `cols` (left untouched) -- and with `extend` their values on the extended domain --
        computed on the side stream, behind what is queued on the compute stream now; returns (copies, extended or None,
        event): the compute stream must wait for the event before it reads them.  The transforms then fill the issue
        slots that the latency-bound tails of the commitments in between leave."""
        if getattr(self, "_side", None) is None:
            import concurrent.futures

            self._side = (self._side_stream(), concurrent.futures.ThreadPoolExecutor(max_workers=1))
        side = self._side[0]
        ready = self.torch.cuda.Event()
        ready.record(self.tstream)
        side.wait_event(ready)
        out = []
        with self.torch.cuda.stream(side):
            sptr = _vp(side.cuda_stream)
            out = [t.clone() for t in cols]
            if len(out) >= 2 and hasattr(self.L, "h2_dev_intt_batch"):
                width = self._batch_width(dom.n)
                tmp = self.torch.empty((min(len(out), width) * dom.n, 4), dtype=self.torch.int64, device=self.dev)
                for at in range(0, len(out), width):
                    part = out[at:at + width]
                    ptrs = (_vp * len(part))(*[c.data_ptr() for c in part])
                    check(self.L.h2_dev_intt_batch(ptrs, len(part), tmp.data_ptr(), _fr(dom.omega_inv), _fr(dom.ifft_divisor),
                                                   dom.k, sptr), "h2_dev_intt_batch")
                tmp.record_stream(self.tstream)
            else:
                for c in out:
                    tmp = self.torch.empty_like(c)
                    check(self.L.h2_dev_intt(c.data_ptr(), tmp.data_ptr(), _fr(dom.omega_inv), _fr(dom.ifft_divisor), dom.k,
                                             sptr), "h2_dev_intt")
            for c in out:
                c.record_stream(self.tstream)
            ext = None
            if extend:
                ext = self.coeffs_to_extended(out, dom, stream=sptr)
                for e in ext:
                    e.record_stream(self.tstream)
            done = self.torch.cuda.Event()
            done.record(side)
        return out, ext, done

    def coeff_to_extended(self, t, dom, out=None, stream=None):
        """`stream`: a caller's side stream (a raw handle; the caller is inside its torch.cuda.stream context, so the buffers
        made here belong to it) -- the launch must be on the stream the input was produced on, not on the compute stream"""
        if stream is None:
            out = out if out is not None else self.empty(dom.extended_n)
            tmp = self.empty(dom.extended_n)
        else:
            out = out if out is not None else self.torch.empty((dom.extended_n, 4), dtype=self.torch.int64, device=self.dev)
            tmp = self.torch.empty((dom.extended_n, 4), dtype=self.torch.int64, device=self.dev)
        check(self.L.h2_dev_coeff_to_extended(t.data_ptr(), out.data_ptr(), tmp.data_ptr(), dom.k, dom.extended_k,
                                              _fr(dom.g_coset), _fr(dom.g_coset_inv), _fr(dom.extended_omega),
                                              self.stream if stream is None else stream), "h2_dev_coeff_to_extended")
        return out

    def coeffs_to_extended(self, ts, dom, stream=None):
        """coeff_to_extended of several coefficient vectors: up to 16 per launch while the extended domain is small enough
        for that to matter (<= 2^23 points: a pass over one vector does not keep the chip busy; the scratch is 16 extended
        vectors), one by one above"""
        count = len(ts)
        if count < 2 or dom.extended_k > 23 or not hasattr(self.L, "h2_dev_coeff_to_extended_batch"):
            return [self.coeff_to_extended(t, dom, stream=stream) for t in ts]
        import contextlib

        torch = self.torch
        # (allocations belong to the stream they are made under: the caller's side stream, or the compute stream)
        width = self._batch_width(dom.extended_n, 2 << 30)
        with (torch.cuda.stream(self.tstream) if stream is None else contextlib.nullcontext()):
            outs = [torch.empty((dom.extended_n, 4), dtype=torch.int64, device=self.dev) for _ in ts]
            tmp = torch.empty((min(count, width) * dom.extended_n, 4), dtype=torch.int64, device=self.dev)
        for at in range(0, count, width):
            part = min(width, count - at)
            src = (_vp * part)(*[t.data_ptr() for t in ts[at:at + part]])
            dst = (_vp * part)(*[o.data_ptr() for o in outs[at:at + part]])
            check(self.L.h2_dev_coeff_to_extended_batch(src, dst, part, tmp.data_ptr(), dom.k, dom.extended_k, _fr(dom.g_coset),
                                                        _fr(dom.g_coset_inv), _fr(dom.extended_omega),
                                                        self.stream if stream is None else stream), "h2_dev_coeff_to_extended_batch")
        return outs

    def extended_to_coeff(self, t, dom):
        tmp = self.empty(dom.extended_n)
        check(self.L.h2_dev_extended_to_coeff(t.data_ptr(), tmp.data_ptr(), dom.extended_k, _fr(dom.g_coset),
                                              _fr(dom.g_coset_inv), _fr(dom.extended_omega_inv),
                                              _fr(dom.extended_ifft_divisor), self.stream), "h2_dev_extended_to_coeff")
        return t

    # -- commitments ------------------------------------------------------------------------------------
    def msm(self, scalars, bases, n, max_bits=254):
        """[Params::commit / commit_lagrange(_with_bound)] over device-resident bases -> affine point"""
        return self.msm_batch([scalars], bases, n, max_bits)[0]

    def msm_async(self, scalars, bases, n, max_bits=254):
        """One commitment on a SIDE stream, driven by a helper thread (the C call blocks until its result is back): the
        caller's stream goes on with other work and collects the point with `.result()`.  Only what is already queued
        on the compute stream is waited for.  One device only: a process group keeps its collectives in program order."""
        import concurrent.futures

        if self.group_size > 1 or self.force_collective:
            point = self.msm(scalars, bases, n, max_bits)
            fut = concurrent.futures.Future()
            fut.set_result(point)
            return fut
        if getattr(self, "_side", None) is None:
            self._side = (self._side_stream(), concurrent.futures.ThreadPoolExecutor(max_workers=1))
        side, pool = self._side
        ready = self.torch.cuda.Event()
        ready.record(self.tstream)
        side.wait_event(ready)
        nbytes = self.L.h2_msm_scratch_bytes(n, max_bits)
        if getattr(self, "_side_scratch", None) is None or self._side_scratch.numel() < nbytes:
            self._side_scratch = self.torch.empty(nbytes, dtype=self.torch.uint8, device=self.dev)
        scratch = self._side_scratch

        def run():
            self.torch.cuda.set_device(self.dev)
            out = np.zeros(12, dtype=np.uint64)
            check(self.L.h2_dev_msm(scalars.data_ptr(), bases.data_ptr(), n, max_bits, scratch.data_ptr(), nbytes,
                                    out.ctypes.data, _vp(side.cuda_stream)), "h2_dev_msm")
            return jacobian_to_affine(out)

        return pool.submit(run)

    def msm_batch(self, columns, bases, n, max_bits=254, also=None):
        """one MSM per column over the same bases (max_bits: one bound or one per column), pipelined inside the
        library; `also` = (scalars, other bases) is one more MSM over a different table in the same pipeline.  With a process group (one process per GPU, every rank holding the same
        polynomials) each MSM is split into contiguous ranges over the ranks -- gpu_multiexp_bound's split
        (arithmetic.rs:413-440) -- and the partial points are all-gathered and folded (parallel.py)."""
        if not columns and not also:
            return []
        lo, hi = 0, n
        collective = self.group_size > 1 or self.force_collective
        if collective:
            from .parallel import allgather_fold_many, msm_split_range

            lo, hi = msm_split_range(n, self.group_size, self.group_rank)
        out = self.msm_partial(columns, bases, lo, hi, max_bits, also)
        if collective:
            out = allgather_fold_many(out, group=self.group, device=self.dev, stream=self.tstream)
        return jacobians_to_affine(out)

    def msm_partial(self, columns, bases, lo, hi, max_bits=254, also=None):
        """the MSMs restricted to the index range [lo, hi): raw Jacobian results, (count (+1), 12) u64.
        max_bits: one bound for every column or a list with one bound per column."""
        items = [(c, bases, b) for c, b in zip(columns, max_bits if isinstance(max_bits, (list, tuple)) else
                                               [max_bits] * len(columns))]
        if also:
            items.append((also[0], also[1], 254))
        count, m = len(items), hi - lo
        per = max((self.L.h2_msm_scratch_bytes(m, b) + 255) // 256 * 256 for b in {b for _, _, b in items})
        groups = {}
        for _, bs, b in items:                      # columns over one table with one bound can be fused by the library
            groups[(bs.data_ptr(), b)] = groups.get((bs.data_ptr(), b), 0) + 1
        nbytes = max([2 * per] + [self.L.h2_msm_batch_scratch_bytes(m, b, cnt) for (_, b), cnt in groups.items()])
        scratch = self.scratch(nbytes)
        out = np.zeros((count, 12), dtype=np.uint64)
        if count == 1:
            c, bs, b = items[0]
            check(self.L.h2_dev_msm(c.data_ptr() + 32 * lo, bs.data_ptr() + 64 * lo, m, b, scratch.data_ptr(), per,
                                    out.ctypes.data, self.stream), "h2_dev_msm")
        elif count > 1:
            sp = (_vp * count)(*[c.data_ptr() + 32 * lo for c, _, _ in items])
            bp = (_vp * count)(*[bs.data_ptr() + 64 * lo for _, bs, _ in items])
            bits = (ctypes.c_uint32 * count)(*[b for _, _, b in items])
            check(self.L.h2_dev_msm_batch_ex(sp, bp, bits, count, m, scratch.data_ptr(), nbytes, out.ctypes.data,
                                             self.stream), "h2_dev_msm_batch_ex")
        return out

    # -- elementwise / scans ----------------------------------------------------------------------------
    def eval_op(self, op, res, l=None, r=None, c=None, size=None):
        size = size if size is not None else res.shape[0]
        check(self.L.h2_dev_eval_op(op, res.data_ptr(), l.data_ptr() if l is not None else None,
                                    r.data_ptr() if r is not None else None, 0, 0, size,
                                    _fr(c) if c is not None else None, self.stream), "h2_dev_eval_op")
        return res

    def eval_polynomial(self, t, n, x):
        out = _FR()
        check(self.L.h2_dev_eval_polynomial(t.data_ptr(), n, _fr(x), out, self.stream), "h2_dev_eval_polynomial")
        return fr_from_mont_limbs(out)

    def eval_polynomial_batch(self, polys, n, points):
        """[poly_j(point_j)]: enqueued back to back, one read-back (the par_iter of plonk/prover.rs:731-737)"""
        count = len(polys)
        if count == 0:
            return []
        ptrs = (_vp * count)(*[p.data_ptr() for p in polys])
        pts = np.array([fr_to_mont_limbs(x % R_MOD) for x in points], dtype=np.uint64)
        out = np.zeros((count, 4), dtype=np.uint64)
        check(self.L.h2_dev_eval_polynomial_batch(ptrs, count, n, pts.ctypes.data, out.ctypes.data, self.stream),
              "h2_dev_eval_polynomial_batch")
        return [fr_from_mont_limbs(r) for r in out]

    def lincomb(self, res, polys, coeffs, size):
        ptrs = (_vp * len(polys))(*[p.data_ptr() for p in polys])
        cf = np.array([fr_to_mont_limbs(c % R_MOD) for c in coeffs], dtype=np.uint64)
        check(self.L.h2_dev_lincomb(res.data_ptr(), ptrs, cf.ctypes.data, len(polys), size, self.stream),
              "h2_dev_lincomb")
        return res

    def kate_division(self, a, n, b, out):
        """out[0 : n-1] = a / (X - b); out[n-1] = 0  (arithmetic.rs:754-773, resized as shplonk/prover.rs:112)"""
        check(self.L.h2_dev_kate_division(a.data_ptr(), n, _fr(b), out.data_ptr(), self.stream), "h2_dev_kate_division")
        with self.torch.cuda.stream(self.tstream):
            out[n - 1:n] = 0
        return out

    def sub_low(self, t, low):
        """t[i] -= low[i] for the first few coefficients"""
        if not low:
            return
        cur = self.get_rows(t, 0, len(low))
        self.set_rows(t, 0, [(c - l) % R_MOD for c, l in zip(cur, low)])

    # -- index-range sharding of the O(n) passes (one proof over several ranks) ------------------
    def row_range(self, n):
        """[lo, hi): the rows / coefficients of an n-vector this rank works on -- the contiguous range its share of every
        range-split MSM consumes (arithmetic.rs:425-435) -- or the whole vector on one device.  Vectors stay full-size
        allocations; a rank only ever computes and reads its own range of the ones that are sharded."""
        if self.group_size > 1 and n % self.group_size == 0 and n // self.group_size >= 8:
            from .parallel import msm_split_range

            return msm_split_range(n, self.group_size, self.group_rank)
        return 0, n

    def _exchange(self, values):
        from .parallel import allgather_scalars

        return allgather_scalars(values, group=self.group, device=self.dev)

    def prefix_scan(self, f, n, init, product, probe):
        """z[0] = init, z[i] = z[i-1] * f[i-1] (product) / + f[i-1]: the grand-product / grand-sum columns
        (permutation/prover.rs:151-160, logup/prover.rs:353-367, shuffle/prover.rs).  Returns (z, z[probe]).
        Sharded: `f` is this rank's rows [lo, hi) of the factors / terms (hi - lo elements); the local scan starts from the
        neutral element, the totals of the ranges cross in ONE all-gather of a field element per rank and the range is
        corrected by the value entering it.  z is a full-size vector valid on [lo, hi) (`gather_rows` completes it)."""
        lo, hi = self.row_range(n)
        kernel = self.L.h2_dev_prefix_product if product else self.L.h2_dev_prefix_sum
        z = self.empty(n)
        if (lo, hi) == (0, n):
            check(kernel(f.data_ptr(), n, _fr(init), z.data_ptr(), self.stream), "h2_dev_prefix_scan")
            return z, self.get_rows(z, probe, 1)[0]
        from .parallel import scan_carries

        m = hi - lo
        tmp = self.empty(m + 1)                      # tmp[t] = f[lo] .. f[lo + t - 1] folded; tmp[m] = this range's total
        check(kernel(f.data_ptr(), m + 1, _fr(1 if product else 0), tmp.data_ptr(), self.stream), "h2_dev_prefix_scan")
        total = self.get_rows(tmp, m, 1)[0]
        at = self.get_rows(tmp, probe - lo, 1)[0] if lo <= probe < hi else 0
        gathered = self._exchange([total, at])
        carries = scan_carries([g[0] for g in gathered], init, product)
        self.eval_op(0 if product else 1, z[lo:hi], tmp[:m], c=carries[self.group_rank], size=m)   # H2_OP_MUL_C / SUM_C
        owner = probe // m
        value = carries[owner] * gathered[owner][1] % R_MOD if product else (carries[owner] + gathered[owner][1]) % R_MOD
        return z, value

    def gather_rows(self, t, n):
        """complete a vector of which every rank computed its own range (all-gather over xGMI)"""
        lo, hi = self.row_range(n)
        if (lo, hi) != (0, n):
            from .parallel import allgather_rows

            allgather_rows(t[:n], lo, hi, group=self.group, stream=self.tstream)
        return t

    def eval_polynomial_ranges(self, polys, n, points):
        """[poly_j(point_j)] with every polynomial cut into the ranks' coefficient ranges: p(x) = sum_r x^(lo_r) p_r(x);
        the partial values cross in one all-gather of len(polys) field elements per rank.  `polys` need only be valid on
        this rank's range."""
        lo, hi = self.row_range(n)
        if (lo, hi) == (0, n):
            return self.eval_polynomial_batch(polys, n, points)
        from .parallel import combine_range_evals

        m = hi - lo
        parts = self.eval_polynomial_batch([p[lo:hi] for p in polys], m, points)
        gathered = self._exchange(parts)
        return [combine_range_evals([g[j] for g in gathered], points[j] % R_MOD, m) for j in range(len(polys))]

    def lincomb_range(self, res, polys, coeffs, n):
        """res[lo:hi) = sum_j coeffs[j] * polys[j][lo:hi) over this rank's range (the whole vector on one device)"""
        lo, hi = self.row_range(n)
        self.lincomb(res[lo:hi], [p[lo:hi] for p in polys], coeffs, hi - lo)
        return res

    def sub_low_range(self, t, low, n):
        """t[i] -= low[i] for the first few coefficients: they live in the first rank's range"""
        lo, hi = self.row_range(n)
        assert len(low) <= hi - lo
        if lo == 0:
            self.sub_low(t, low)

    def kate_division_ranges(self, a, n, b, out):
        """out = a / (X - b) (arithmetic.rs:754-773; out[n-1] = 0), `a` valid on this rank's range, `out` filled on it:
        out[j] = sum_{t > j} a[t] b^(t - j - 1).  The part of the sum inside the range is the same kernel on the range; the
        part above it is C b^(hi - 1 - j) with C = the recurrence value entering the range from above -- the local Horner
        values of the ranges at b cross in one all-gather, C follows on the host (parallel.kate_carries)."""
        lo, hi = self.row_range(n)
        if (lo, hi) == (0, n):
            return self.kate_division(a, n, b, out)
        from .parallel import kate_carries

        m = hi - lo
        self.kate_division(a[lo:hi], m, b, out[lo:hi])               # the last coefficient of the range is zeroed
        part = self.eval_polynomial(a[lo:hi], m, b)
        carry = kate_carries([g[0] for g in self._exchange([part])], b % R_MOD, m)[self.group_rank]
        if carry:
            b %= R_MOD
            if b == 0:                                                # only out[hi - 1] = C
                self.set_rows(out, hi - 1, [carry])
            else:
                corr = self.eval_op(8, self.empty(m), c=carry * pow(b, m - 1, R_MOD) % R_MOD)      # H2_OP_CONSTANT
                check(self.L.h2_dev_distribute_powers(corr.data_ptr(), m, _fr(_inv(b)), self.stream), "h2_dev_distribute_powers")
                self.eval_op(2, out[lo:hi], out[lo:hi], corr, size=m)                              # H2_OP_SUM
        return out


def sharding_description(device):
    """how one proof is spread over the ranks of `device`'s process group (bench.py reports it)"""
    if device.group_size <= 1 and not device.force_collective:
        return "one proof on one device"
    return ("one proof over %d rank(s): every MSM range-split over the ranks (partial points: one all-gather per batch, folded "
            "on the device); extended-domain phase (coset NTTs, evaluate_h, vanishing division, inverse transform) split by "
            "coset of the n-th roots of unity, the per-coset quotients scattered as coefficient ranges; the O(n) passes of the "
            "permutation / lookup / shuffle products (terms, batch inversion, prefix scans), the evaluations and the multiopen "
            "argument (linear combinations, Kate divisions) on the rank's row / coefficient range, one field element per rank "
            "exchanged per scan / division / evaluation batch; witness-dependent inverse transforms dealt by column (rows to "
            "the owner, coefficient vectors broadcast on a second communicator and a side stream); with more ranks than cosets "
            "the ranks of a coset deal its columns for the coset transforms, exchange row slices and evaluate the quotient by "
            "row range; lookup inputs compressed by rows, multiplicities all-reduced as integer counts" % device.group_size)


def _forget_tables(L, ptrs):
    for ptr in ptrs:
        L.h2_dev_bases_forget(ptr)


class Params:
    """poly/commitment.rs:23-29: k, n, g, g_lagrange -- both tables resident on the device"""

    def __init__(self, device, k, g, g_lagrange, tables=None):
        self.k, self.n = k, 1 << k
        self.g = g if not isinstance(g, np.ndarray) else device.upload(g)
        self.g_lagrange = g_lagrange if not isinstance(g_lagrange, np.ndarray) else device.upload(g_lagrange)
        assert self.g.shape[0] == self.n and self.g_lagrange.shape[0] == self.n
        self.table_bytes = 0
        if tables is None:
            tables = os.environ.get("H2_MSM_TABLES", "1") != "0"
        if tables:
            self.precompute_tables(device)

    def precompute_tables(self, device, digits=0):
        """Shifted-base tables of both point sets (h2_dev_bases_precompute, include/halo2_hip.h): every commitment of
        every proof made with these parameters adds all digits of a scalar into one bucket set -- 10-35 % off each MSM
        for digits x n x 64 B of HBM per table (12 GiB at k = 24) and ~0.25 s of doublings, once.  Skipped below 2^15
        rows (no gain) and when the tables would take more than half of the free device memory.  The tables live in
        library memory keyed by the tensors' addresses; they are dropped when this object is collected."""
        import weakref

        L = device.L
        # one proof over several ranks: this rank only ever commits its own contiguous range of the bases (the range
        # split of every MSM), so the tables cover that range only -- and their digit count is chosen for its length
        lo, hi = 0, self.n
        if device.group_size > 1:
            from .parallel import msm_split_range

            lo, hi = msm_split_range(self.n, device.group_size, device.group_rank)
        rows = hi - lo
        if rows < (1 << 15) or self.table_bytes:
            return False
        one = L.h2_dev_bases_precompute_bytes(rows, digits)
        free, _ = device.torch.cuda.mem_get_info(device.dev)
        # what the tables may take: half of the free memory, and under a memory budget (H2_DEVICE_MEM_BUDGET) a third of
        # it.  Each base set is optional on its own: g_lagrange first (the advice / product / multiplicity columns of a
        # wide circuit are committed against it; g only takes the h pieces, the random polynomial and the openings).
        room = free // 2 if device.mem_budget is None else min(free // 2, device.mem_budget // 3)
        which = [self.g_lagrange, self.g][:max(0, min(2, room // one))] if one else []
        if not which:
            return False
        device.sync()
        ptrs = [t.data_ptr() + 64 * lo for t in which]
        for ptr in ptrs:
            check(L.h2_dev_bases_precompute(ptr, rows, digits, device.stream), "h2_dev_bases_precompute")
        self.table_bytes = one * len(ptrs)
        weakref.finalize(self, _forget_tables, L, ptrs).atexit = False   # at interpreter exit the process frees them
        return True

    @staticmethod
    def unsafe_setup(device, k, s):
        """Params::unsafe_setup (poly/commitment.rs:56-124) with the toxic scalar `s` supplied by the caller instead of
        OsRng -- tests and benchmarks only, MUST NOT be used in production (as the reference says).
        g[i] = [s^i] G (:67-83), g_lagrange[i] = [(s^n - 1)/n * w^i / (s - w^i)] G (:85-112), all on the device."""
        from .transcript import Q_MOD

        D, L = device, device.L
        n = 1 << k
        s %= R_MOD
        omega = pow(ROOT_OF_UNITY, 1 << (S - k), R_MOD)
        # table of [2^j] G, j < 254 (affine; host big integers, 254 doublings)
        pts, P = [], (1, 2)
        for _ in range(254):
            pts.append(P)
            lam = 3 * P[0] * P[0] * pow(2 * P[1], -1, Q_MOD) % Q_MOD
            x3 = (lam * lam - 2 * P[0]) % Q_MOD
            P = (x3, (lam * (P[0] - x3) - P[1]) % Q_MOD)
        mq = lambda v: [((v << 256) % Q_MOD >> (64 * i)) & ((1 << 64) - 1) for i in range(4)]  # noqa: E731
        table = D.upload(np.array([mq(x) + mq(y) for x, y in pts], dtype=np.uint64))

        def powers(base):                       # [base^i]: the running product of a constant column
            f = D.eval_op(8, D.empty(n), c=base)                                # H2_OP_CONSTANT
            out = D.empty(n)
            check(L.h2_dev_prefix_product(f.data_ptr(), n, _fr(1), out.data_ptr(), D.stream), "h2_dev_prefix_product")
            return out

        def fixed_base(scalars):
            with D.torch.cuda.stream(D.tstream):
                out = D.torch.empty((n, 8), dtype=D.torch.int64, device=D.dev)
            check(L.h2_dev_fixed_base_mul(scalars.data_ptr(), table.data_ptr(), n, out.data_ptr(), D.stream),
                  "h2_dev_fixed_base_mul")
            return out

        g = fixed_base(powers(s))
        w = powers(omega)
        t = D.eval_op(1, D.empty(n), w, c=-s)                                   # w^i - s
        check(L.h2_dev_batch_invert(t.data_ptr(), D.empty(n).data_ptr(), n, D.stream), "h2_dev_batch_invert")
        D.eval_op(3, t, t, w)                                                   # w^i / (w^i - s)
        multiplier = (pow(s, n, R_MOD) - 1) * pow(n, -1, R_MOD) % R_MOD
        D.eval_op(0, t, t, c=-multiplier)                                       # multiplier * w^i / (s - w^i)
        g_lagrange = fixed_base(t)
        D.sync()
        return Params(D, k, g, g_lagrange)

    @staticmethod
    def synthetic(device, k, seed=0x48414C4F32):
        """Timing-only parameters: two tables of valid curve points with no common trapdoor, so proofs made
        with them exercise exactly the same work but cannot verify."""
        n = 1 << k
        tabs = []
        for i in range(2):
            with device.torch.cuda.stream(device.tstream):
                t = device.torch.empty((n, 8), dtype=device.torch.int64, device=device.dev)
            check(device.L.h2_dev_random_points(seed + i, n, t.data_ptr(), device.stream), "h2_dev_random_points")
            tabs.append(t)
        device.sync()
        return Params(device, k, tabs[0], tabs[1])


def max_scalar_bits(col):
    """find_max_scalar_bits (plonk/prover.rs:237-254) on a canonical (n, 4) u64 column"""
    for limb in (3, 2, 1, 0):
        m = int(col[:, limb].max())
        if m:
            return 64 * limb + m.bit_length()
    return 0


def permutation_mapping(ncols, n, copies):
    """`copies`: (m, 4) integers (left column position, left row, right column position, right row).
    Returns (map_col, map_row) u32 arrays of shape (ncols, n): every cycle sorted by (column, row), each cell
    pointing at its successor (plonk/permutation/keygen.rs:112-143)."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components

    ids = np.arange(ncols * n, dtype=np.int64)
    nxt = ids.copy()
    copies = np.asarray(copies, dtype=np.int64).reshape(-1, 4)
    if len(copies):
        l = copies[:, 0] * n + copies[:, 1]
        r = copies[:, 2] * n + copies[:, 3]
        nodes, inv = np.unique(np.concatenate([l, r]), return_inverse=True)
        m = len(nodes)
        graph = coo_matrix((np.ones(len(l), dtype=np.int8), (inv[:len(l)], inv[len(l):])), shape=(m, m))
        _, labels = connected_components(graph, directed=False)
        order = np.lexsort((nodes, labels))          # by cycle, then by (column, row)
        sl, sn = labels[order], nodes[order]
        succ = np.roll(sn, -1)
        ends = np.flatnonzero(sl != np.roll(sl, -1)) if m > 1 else np.array([0])
        starts = np.concatenate([[0], ends[:-1] + 1])
        succ[ends] = sn[starts]
        nxt[sn] = succ
    return (nxt // n).astype(np.uint32).reshape(ncols, n), (nxt % n).astype(np.uint32).reshape(ncols, n)


def vk_digest(cs, dom, fixed_commitments, perm_commitments):
    """VerifyingKey::hash_into (plonk.rs:91-109): Blake2b-512 ("Halo2-Verify-Key") over a u64 length and the pinned
    verifying key, reduced by from_bytes_wide.  The reference pins `format!("{:?}", vk.pinned())` -- the Debug text of
    the domain, the whole constraint system and the commitments -- which cannot be reproduced without the Rust binary
    (parity unpinned); this build hashes the same CONTENT in a canonical binary form: domain (k, extended_k, omega),
    both field moduli, the write_cs serialisation of the constraint system (gates, queries, permutation columns, lookups,
    shuffles, instance columns: formats.cs_store) and the fixed / permutation commitments.  Two circuits that differ in
    any gate, lookup or query therefore get different transcripts.  `keygen(..., transcript_repr=...)` overrides the
    value with one dumped from the Rust side (tools/ref_dump) once that can be pinned."""
    from .formats import cs_store
    from .transcript import Q_MOD

    cs_bytes = cs_store(cs)
    body = [b"halo2-hip-vk-v2", dom.k.to_bytes(4, "little"), dom.extended_k.to_bytes(4, "little"),
            dom.omega.to_bytes(32, "little"), R_MOD.to_bytes(32, "little"), Q_MOD.to_bytes(32, "little"),
            len(cs_bytes).to_bytes(4, "little"), cs_bytes]
    for group in (fixed_commitments, perm_commitments):
        body.append(len(group).to_bytes(4, "little"))
        body += [point_to_bytes(p) for p in group]
    body = b"".join(body)
    h = hashlib.blake2b(digest_size=64, person=b"Halo2-Verify-Key")
    h.update(len(body).to_bytes(8, "little"))
    h.update(body)
    return int.from_bytes(h.digest(), "little") % R_MOD


class ProvingKey:
    pass


def program_descriptor(cs, k, extended_k, graph=None, value_parts=None, lookup_calcs=None, shuffle_calcs=None):
    """the evaluate_h descriptor of a circuit's PROGRAM alone -- constants, rotations, calculations, value parts, lookup /
    shuffle calculations, the permutation argument's shape; every column pointer null -- as h2_evalh_prepare /
    h2_evalh_compile / h2_evalh_source take it"""
    if graph is None:
        graph, value_parts, lookup_calcs, shuffle_calcs = compile_evaluator(cs)
    ncols, chunk = len(cs.perm_columns), cs.degree() - 2
    nsets = (ncols + chunk - 1) // chunk if ncols else 0
    zero = fr_to_mont_limbs(0)
    nz = [len(sets) for _, _, sets in cs.lookups]
    return ev.Builder().build(
        k=k, extended_k=extended_k, blinding_factors=cs.blinding_factors(), chunk_len=chunk,
        constants=np.array([fr_to_mont_limbs(c) for c in graph.constants], dtype=np.uint64), rotations=graph.rotations,
        calculations=graph.calculations, value_parts=value_parts, lookups=lookup_calcs, shuffles=shuffle_calcs,
        fixed=[0] * cs.num_fixed, advice=[0] * cs.num_advice, instance=[0] * cs.num_instance,
        perm_z=[0] * nsets, perm_columns=[(_ANY[kd], i) for kd, i in cs.perm_columns], perm_sigma=[0] * ncols,
        lookup_z=[0] * sum(nz), lookup_m=[0] * len(nz), shuffle_z=[0] * len(shuffle_calcs),
        y=zero, beta=zero, gamma=zero, theta=zero, delta=zero, zeta=zero, extended_omega=zero)


def keygen(device, params, cs, fixed, copies, mapping=None, fixed_montgomery=False, transcript_repr=None):
    """keygen_vk + keygen_pk.  fixed: list of canonical (n, 4) u64 columns; copies: see permutation_mapping.
    `mapping` = (map_col, map_row) replaces `copies` and `fixed_montgomery` marks columns already in the in-memory
    representation: the two things a CircuitData file holds (keygen_pk_from_info, plonk/keygen.rs:458-553)."""
    D, L = device, device.L
    dom = Domain(params.k, cs.degree())
    n, bf = dom.n, cs.blinding_factors()
    assert n >= cs.minimum_rows()
    pk = ProvingKey()
    pk.cs, pk.domain = cs, dom
    plan = D.coset_plan(dom)
    # one device under a memory budget: when the extended cosets do not fit, the proving key keeps coefficient forms only
    # and the extended-domain phase runs coset by coset (all quotient_poly_degree of them, tables built on demand).
    # Decided before anything of this key is allocated: the estimate is compared with the memory that is free NOW.
    pk.residency, keep = ("cosets", None) if plan is not None else D.residency(cs, dom)
    if plan is None and pk.residency == "cosets":
        plan = (dom.quotient_poly_degree, 1, list(range(dom.quotient_poly_degree)))
    # fixed columns: values, coefficient form, extended cosets
    pk.fixed_values = []
    for col in fixed:
        t = D.upload(col)
        if not fixed_montgomery:
            check(L.h2_dev_batch_mont(t.data_ptr(), n, D.stream), "h2_dev_batch_mont")
        pk.fixed_values.append(t)
    pk.fixed_commitments = D.msm_batch(pk.fixed_values, params.g_lagrange, n, 254)
    pk.fixed_polys = [D.intt(D.clone(t), dom) for t in pk.fixed_values]
    pk.fixed_cosets = [D.coeff_to_extended(t, dom) for t in pk.fixed_polys] if plan is None else None
    # permutation: sigma columns (Lagrange), polys, cosets
    ncols = len(cs.perm_columns)
    map_col, map_row = mapping if mapping is not None else permutation_mapping(ncols, n, copies)
    assert len(map_col) == ncols and all(len(c) == n for c in map_col)
    pk.mapping = (map_col, map_row)
    pk.sigma_values = []
    for i in range(ncols):
        out = D.empty(n)
        with D.torch.cuda.stream(D.tstream):
            mc = D.torch.from_numpy(np.ascontiguousarray(map_col[i], dtype=np.uint32).view(np.int32)).to(D.dev)
            mr = D.torch.from_numpy(np.ascontiguousarray(map_row[i], dtype=np.uint32).view(np.int32)).to(D.dev)
        check(L.h2_dev_permutation_sigma(out.data_ptr(), mc.data_ptr(), mr.data_ptr(), n, _fr(DELTA), _fr(dom.omega),
                                         D.stream), "h2_dev_permutation_sigma")
        D.sync()
        pk.sigma_values.append(out)
    pk.perm_commitments = D.msm_batch(pk.sigma_values, params.g_lagrange, n, 254)
    pk.sigma_polys = [D.intt(D.clone(t), dom) for t in pk.sigma_values]
    pk.sigma_cosets = [D.coeff_to_extended(t, dom) for t in pk.sigma_polys] if plan is None else None
    # l0, l_last, l_active_row = 1 - (l_last + l_blind) on the extended coset (keygen.rs:395-425)
    def lagrange_poly(rows):
        t = D.zeros(n)
        D.set_rows(t, rows[0], [1] * len(rows))
        return D.intt(t, dom)

    l0_poly, l_last_poly = lagrange_poly([0]), lagrange_poly([n - bf - 1])
    l_blind_poly = lagrange_poly(list(range(n - bf, n)))

    def active_row(l_last, l_blind, size):
        tmp = D.eval_op(2, D.empty(size), l_last, l_blind)                     # H2_OP_SUM
        one = D.eval_op(8, D.empty(size), c=1)                                 # H2_OP_CONSTANT
        return D.eval_op(4, one, one, tmp)                                     # H2_OP_SUB

    def coset_tables(j):
        nf, ns = len(pk.fixed_polys), len(pk.sigma_polys)
        vals = D.coeffs_to_coset(list(pk.fixed_polys) + list(pk.sigma_polys) + [l0_poly, l_last_poly, l_blind_poly], dom, j)
        return {
            "fixed": vals[:nf], "sigma": vals[nf:nf + ns], "l0": vals[nf + ns], "l_last": vals[nf + ns + 1],
            "l_active_row": active_row(vals[nf + ns + 1], vals[nf + ns + 2], n),
        }

    # kept with the key (three n-vectors behind the closure): a proof of SEVERAL circuit instances may not fit the
    # residency decided here for one and then runs coset by coset from tables built on demand (create_proof_ext)
    pk.coset_builder = coset_tables
    pk.l0_poly, pk.l_last_poly = l0_poly, l_last_poly     # (the cuda-shaped evaluator takes l0 / l_last as coefficient forms)
    if plan is None:
        pk.l0, pk.l_last = D.coeff_to_extended(l0_poly, dom), D.coeff_to_extended(l_last_poly, dom)
        pk.l_active_row = active_row(pk.l_last, D.coeff_to_extended(l_blind_poly, dom), dom.extended_n)
        pk.coset = None
    else:
        # one proof over several ranks: only the cosets this rank evaluates, each an n-point table
        # (under a memory budget on one device: every coset, built on demand and retained up to `keep` of them)
        pk.l0 = pk.l_last = pk.l_active_row = None
        pk.coset = CosetTables(coset_tables, plan[2], keep)
        if keep is None:                       # one proof over several ranks: this rank's cosets stay resident
            for j in plan[2]:
                pk.coset[j]
    # read by every proof, never written (sigma_values: the permutation argument's denominators, permutation/prover.rs:89-128;
    # l_active_row: the extended values the evaluator takes as they are)
    D.retain(list(pk.fixed_polys) + list(pk.sigma_polys) + [l0_poly, l_last_poly] + list(pk.sigma_values) +
             ([pk.l_active_row] if pk.l_active_row is not None else []), owner=pk)
    pk.t_evaluations = D.upload(np.array([fr_to_mont_limbs(v) for v in dom.t_evaluations], dtype=np.uint64))
    # Evaluator::new: the gate program with the lookup / shuffle result calculations, and the compression programs
    # (evaluate_with_theta) of every lookup / shuffle expression list
    pk.graph, pk.value_parts, pk.lookup_calcs, pk.shuffle_calcs = compile_evaluator(cs)
    pk.lookup_programs = [(compile_compress(table), [[compile_compress(inputs) for inputs in st] for st in sets])
                          for _, table, sets in cs.lookups]
    pk.shuffle_programs = [[(compile_compress(inp), compile_compress(shf)) for _, inp, shf in group]
                           for group in cs.shuffles]
    # The library generates and compiles the program's kernels itself the first time a descriptor carries it
    # (csrc/evalh_gen.cpp, hipRTC); doing that here moves the cost from the first proof to keygen and reports what was built.
    # None = the interpreter kernels run (H2_EVALH_JIT=0, or no hipRTC on this machine).
    pk.evalh_stats = None
    if os.environ.get("H2_EVALH_JIT", "1") != "0":
        try:
            pk.evalh_stats = ev.prepare(program_descriptor(cs, dom.k, dom.extended_k, pk.graph, pk.value_parts, pk.lookup_calcs,
                                                           pk.shuffle_calcs))
        except H2Error as e:
            import warnings

            warnings.warn("evaluate_h keeps the interpreter kernels: %s" % e)
    pk.transcript_repr = (transcript_repr if transcript_repr is not None else
                          vk_digest(cs, dom, pk.fixed_commitments, pk.perm_commitments))
    D.sync()
    return pk


def keygen_from_info(device, params, info):
    """CircuitData::into_proving_key (plonk.rs:196-198): `info` = formats.circuit_data_read(path).  The commitments
    are recomputed from the columns and must equal the ones the file carries."""
    if info["k"] != params.k:
        raise ValueError("circuit data for k = %d under params of k = %d" % (info["k"], params.k))
    pk = keygen(device, params, info["cs"], info["fixed"], None, mapping=info["mapping"], fixed_montgomery=True)
    have = [point_to_bytes(P) for P in list(pk.fixed_commitments) + list(pk.perm_commitments)]
    if have != list(info["fixed_commitments"]) + list(info["perm_commitments"]):
        raise ValueError("circuit data: the verifying key's commitments do not match its columns under these params")
    return pk


def _intermediate_sets(queries):
    """construct_intermediate_sets (poly/multiopen/shplonk.rs:58-135) on (key, rotation, point, eval) tuples:
    BTreeMap / BTreeSet iteration orders become sorted()."""
    rot_point = {}
    for _, rot, point, _ in queries:
        assert rot_point.setdefault(rot, point) == point
    super_point_set = [rot_point[r] for r in sorted(rot_point)]
    order, rotsets = [], {}
    for key, rot, _, _ in queries:
        if key not in rotsets:
            rotsets[key] = set()
            order.append(key)
        rotsets[key].add(rot)
    groups = {}
    for key in order:
        groups.setdefault(tuple(sorted(rotsets[key])), []).append(key)
    evals = {(key, rot): e for key, rot, _, e in queries}
    sets = [{"points": [rot_point[r] for r in rots],
             "commitments": [(key, [evals[(key, r)] for r in rots]) for key in groups[rots]]}
            for rots in sorted(groups)]
    return sets, super_point_set


def _lagrange_interpolate(points, evals):
    """arithmetic.rs:849-903 on host integers (at most a handful of points)"""
    n = len(points)
    out = [0] * n
    for j in range(n):
        num, den = [1], 1
        for m in range(n):
            if m != j:
                num = [((num[i - 1] if i else 0) - points[m] * (num[i] if i < len(num) else 0)) % R_MOD
                       for i in range(len(num) + 1)]
                den = den * (points[j] - points[m]) % R_MOD
        c = evals[j] * _inv(den) % R_MOD
        for i in range(n):
            out[i] = (out[i] + c * num[i]) % R_MOD
    return out


def _horner(coeffs, x):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % R_MOD
    return acc


def _vanishing(roots, z):
    acc = 1
    for r in roots:
        acc = acc * (z - r) % R_MOD
    return acc


def create_proof(device, params, pk, advice, rng, timings=None, instances=()):
    """plonk/prover.rs:877-893: the GWC multiopen, as the reference's `create_proof`"""
    return create_proof_ext(device, params, pk, advice, rng, True, timings, instances)


def create_proof_with_shplonk(device, params, pk, advice, rng, timings=None, instances=()):
    """plonk/prover.rs:856-871"""
    return create_proof_ext(device, params, pk, advice, rng, False, timings, instances)


_ANY = {"advice": ev.ANY_ADVICE, "fixed": ev.ANY_FIXED, "instance": ev.ANY_INSTANCE}


def _compress(D, dom, program, theta, fixed, advice, instance, rows=None):
    """evaluate_with_theta (plonk/evaluation.rs:2330-2398): the theta-compression of an expression list over the
    n-point Lagrange domain = the evaluator program with y := theta and extended_k := k.  The descriptor of a program
    is built once per device and re-bound to the columns / theta of each call (building it costs ~0.1 ms of host time,
    a k = 18 proof compresses eight expression lists).  `rows` = (first, count): only these rows are computed (one rank's
    share of a proof dealt by rows); the result is a full-size vector valid there."""
    g, parts = program
    # the pure-column fast path of the reference (plonk/evaluation.rs:2266-2276): ONE expression that is a plain query at the
    # current rotation compresses to the column itself -- no kernel, no copy (the callers only read the result; an advice column
    # keeps its Lagrange values until the quotient phase turns it into coefficients, after every lookup pass has consumed them)
    if len(parts) == 1 and not g.calculations and parts[0].kind in (ev.VS_FIXED, ev.VS_ADVICE, ev.VS_INSTANCE) and \
            g.rotations[parts[0].rot] == 0 and os.environ.get("H2_COMPRESS_PURE", "1") != "0":
        return {ev.VS_FIXED: fixed, ev.VS_ADVICE: advice, ev.VS_INSTANCE: instance}[parts[0].kind][parts[0].index]
    cache = D.__dict__.setdefault("_compress_descs", {})
    pointers = dict(fixed=[t.data_ptr() for t in fixed], advice=[t.data_ptr() for t in advice],
                    instance=[t.data_ptr() for t in instance])
    hit = cache.get((id(program), dom.k))
    if hit is not None and hit[0] is program:
        b = hit[1].rebind(y=fr_to_mont_limbs(theta), theta=fr_to_mont_limbs(theta), **pointers)
        b.desc.row_begin, b.desc.row_count = rows if rows is not None else (0, 0)
    else:
        zero = fr_to_mont_limbs(0)
        b = ev.Builder().build(
            k=dom.k, extended_k=dom.k, blinding_factors=0, chunk_len=1,
            constants=np.array([fr_to_mont_limbs(c) for c in g.constants], dtype=np.uint64), rotations=g.rotations,
            calculations=g.calculations, value_parts=parts,
            y=fr_to_mont_limbs(theta), beta=zero, gamma=zero, theta=fr_to_mont_limbs(theta),
            delta=fr_to_mont_limbs(DELTA), zeta=fr_to_mont_limbs(ZETA), extended_omega=fr_to_mont_limbs(dom.omega),
            row_begin=rows[0] if rows is not None else 0, row_count=rows[1] if rows is not None else 0, **pointers)
        cache[(id(program), dom.k)] = (program, b)
    out = D.empty(dom.n)
    check(D.L.h2_dev_evaluate_h(ctypes.byref(b.desc), out.data_ptr(), D.stream), "h2_dev_evaluate_h (compress)")
    return out


def range_check_assigner(vmin, vmax, step):
    """RangeCheckRelAssigner (plonk/range_check.rs:40-63): vmin, vmin + step, ... capped at vmax, then vmax itself"""
    out, cur = [], vmin
    while True:
        value = cur
        if value < vmax:
            cur = min(value + step, vmax)
            out.append(value)
        elif cur == vmax:
            cur += step
            out.append(value)
        else:
            return out


def complete_range_check_witness(cs, n, advice, first_unassigned=None):
    """What `create_proof` does to the witness of every `advice_column_range` after synthesis (plonk/prover.rs:1699-1783):
    every value of the range is planted in the unused cells of the range-checked column from the last usable row upwards
    (so that its sorted copy starts at min, ends at max and has no gap wider than step), and the companion column
    becomes the counting sort of the usable rows (`sort`, prover.rs:164-200).  In place on canonical (n, 4) u64 host
    columns, like the reference; `first_unassigned[column]` (optional) is checked as the reference asserts it."""
    usable = n - (cs.blinding_factors() + 1)
    last_active = usable - 1
    for origin, sort, vmin, vmax, step in cs.range_checks:
        col, companion = advice[origin], advice[sort]
        if not isinstance(col, np.ndarray) or not isinstance(companion, np.ndarray):
            raise TypeError("range check: the range-checked column and its companion must be host columns")
        low = lambda c: c if c.ndim == 1 else c[:, 0]            # noqa: E731  (compact columns hold limb 0 only)
        values = np.array(range_check_assigner(vmin, vmax, step), dtype=np.uint64)
        lo = last_active + 1 - len(values)
        # the reference asserts first_unassigned_offset <= (the offset below the last planted cell) = lo - 1 (prover.rs:1731)
        if lo < 1 or (first_unassigned is not None and first_unassigned.get(origin, 0) >= lo):
            raise ValueError("range check: the range does not fit the unused cells of its column")
        if first_unassigned is None:
            # synthesis did not say which cells it assigned: the cells about to be planted (and the spare one below them)
            # must be untouched -- zero -- or already hold exactly the planted values (the same host columns proved again);
            # a witness that uses them would otherwise be silently overwritten and a different statement proved
            target = low(col)[lo:last_active + 1]
            wide_clear = col.ndim == 1 or not col[lo - 1:last_active + 1, 1:].any()
            planted = np.array_equal(target, values[::-1])
            if not wide_clear or low(col)[lo - 1] != 0 or not (planted or not target.any()):
                raise ValueError("range check: the witness already uses the cells the range is planted in")
        low(col)[lo:last_active + 1] = values[::-1]
        if col.ndim == 2:
            col[lo:last_active + 1, 1:] = 0
        body = low(col)[:usable]
        if (col.ndim == 2 and col[:usable, 1:].any()) or int(body.max()) > vmax or int(body.min()) < vmin:
            raise ValueError("range check: a value of the column lies outside its range")   # the reference's HashMap lookup panics
        if vmax - vmin < (1 << 24):         # the reference's counting sort (`sort`, prover.rs:164-200): O(n + range)
            counts = np.bincount((body - np.uint64(vmin)).astype(np.int64), minlength=vmax - vmin + 1)
            low(companion)[:usable] = np.repeat(np.arange(vmin, vmax + 1, dtype=np.uint64), counts)
        else:
            low(companion)[:usable] = np.sort(body, kind="stable")
        if companion.ndim == 2:
            companion[:usable, 1:] = 0
    return advice


def create_proof_from_witness(device, params, pk, witness, rng, use_gwc=True, timings=None, instances=()):
    """plonk/prover.rs:916-1500: the advice columns come from a witness file (formats.witness_fetch), i.e. in the
    in-memory Montgomery representation"""
    return create_proof_ext(device, params, pk, witness, rng, use_gwc, timings, instances, montgomery=True)


def create_proof_ext(device, params, pk, advice, rng, use_gwc, timings=None, instances=(), montgomery=False,
                     first_unassigned=None):
    """plonk/prover.rs:206-850.  advice: list of (n, 4) u64 columns, canonical integers (or Montgomery residues with
    montgomery=True); rows past the usable range are overwritten with blinding values; instances: one list of
    canonical integers per instance column; rng: a rng.ProverRng.  Returns the proof bytes.
    A column may also be COMPACT -- a 1-D u64 array of n values below 2^64 (booleans, bytes, limbs: 8 bytes per cell over
    PCIe instead of 32, widened on the device) -- or a device tensor of canonical scalars (a witness already resident).

    Several circuit instances in one proof (`circuits: &[ConcreteCircuit]`, prover.rs:206-232): pass `advice` as a list
    of such column lists and `instances` as the matching list of instance-column lists.  Every phase then runs circuit
    by circuit in the reference's order (instance commitments, advice commitments, theta, lookup multiplicities, beta /
    gamma, permutation / lookup / shuffle products, y, ONE quotient over all circuits, x, evaluations, openings).

    `first_unassigned`: {advice column index: first row synthesis left unassigned} (one dict, or one per circuit instance) for
    the range-checked columns -- what the reference's assignment tracking records (prover.rs:1706-1731); without it the
    cells the range is planted in must be zero."""
    import time

    D, L = device, device.L
    cs, dom = pk.cs, pk.domain
    n, bf, ek = dom.n, cs.blinding_factors(), dom.extended_k
    en = dom.extended_n
    last_rot = -(bf + 1)
    usable = n - (bf + 1)
    marks = [("start", time.perf_counter())]

    if timings is not None and D.group_size > 1:
        from . import parallel as _par

        _par.comm_trace_begin()              # this (untimed) proof records what its collectives cost, phase by phase

    _host_trace = [] if os.environ.get("H2_PROVER_HOST_TRACE") else None

    def htrace(name):                         # host-side timestamps without any synchronisation (H2_PROVER_HOST_TRACE=1)
        if _host_trace is not None:
            _host_trace.append((name, time.perf_counter()))

    def mark(name):
        htrace("mark " + name)
        if timings is not None:
            D.sync()
            marks.append((name, time.perf_counter()))
            if D.group_size > 1:
                _par.comm_trace_phase(name)
        if TRACE_TRANSCRIPT:                 # where two runs (or two ranks) of one proof part ways: the transcript after a phase
            import sys

            sys.stderr.write("h2 trace: rank %d after %s: %s (%d bytes)\n" % (
                D.group_rank, name, hashlib.sha256(bytes(transcript.writer)).hexdigest()[:12], len(transcript.writer)))

    transcript = Blake2bWrite()
    transcript.common_scalar(pk.transcript_repr)
    D.release_retained()                     # (a proof that raised half way left its registrations behind)
    if D.group_size > 1:
        rng = rng.shared(D.group)            # every rank of one proof draws the same blinding values

    def commit_lagrange_with_tail(cols_, bits_, split_tail):
        """commit_lagrange of columns whose USABLE rows are bounded by bits_[i] and whose bf + 1 blinding rows are 16-bit
        values (advice columns, the lookups' multiplicities).  `split_tail`: a column whose usable rows are narrower than the
        blinding rows (and large enough for the narrow shapes of the MSM to matter) is committed as two sums -- the usable
        rows under THEIR bound, the blinding rows as one more (fused, few-point) MSM -- and the two points are added; the
        others are committed whole, under the bound of the whole column."""
        narrow_ = [i for i, b in enumerate(bits_) if split_tail and b <= 12 and n >= (1 << 20)]
        whole_ = [i for i in range(len(cols_)) if i not in narrow_]
        points_ = [None] * len(cols_)
        if whole_:
            wb = [max(bits_[i], 16) if split_tail else bits_[i] for i in whole_]
            for i, P in zip(whole_, D.msm_batch([cols_[i] for i in whole_], params.g_lagrange, n, wb)):
                points_[i] = P
        if narrow_:
            main_ = D.msm_batch([cols_[i] for i in narrow_], params.g_lagrange, usable, [bits_[i] for i in narrow_])
            tail_ = D.msm_batch([cols_[i][usable:] for i in narrow_], params.g_lagrange[usable:], n - usable, 16)
            for i, a_, b_ in zip(narrow_, main_, tail_):
                points_[i] = g1_add_affine(a_, b_)
        return points_

    multi = len(advice) > 0 and isinstance(advice[0], (list, tuple))
    advice_sets = [list(a) for a in advice] if multi else [list(advice)]
    instance_sets = [list(i) for i in instances] if multi else [list(instances)]
    if len(instance_sets) != len(advice_sets):
        raise ValueError("InvalidInstances")
    ncirc, nadv = len(advice_sets), len(advice_sets[0])
    if any(len(a) != nadv for a in advice_sets):
        raise ValueError("every circuit instance needs the same advice columns")
    if cs.range_checks:
        if montgomery:
            raise ValueError("range-check witness completion needs canonical advice columns")
        fu = first_unassigned if isinstance(first_unassigned, (list, tuple)) else [first_unassigned] * len(advice_sets)
        for a, f in zip(advice_sets, fu):                     # prover.rs:1699-1783: plant the range, sort the companion
            complete_range_check_witness(cs, n, a, f)
    advice = [col for a in advice_sets for col in a]          # circuit-major: the order every phase walks them in
    # The residency of the key was decided at keygen for ONE circuit instance; advice, product and lookup polynomials scale
    # with the number of instances.  A key judged 'extended' whose multi-instance proof does not fit runs this proof by the
    # coset route, from tables built on demand out of the key's coefficient forms (same bytes).
    coset_tabs = pk.coset
    if ncirc > 1 and coset_tabs is None and D.group_size <= 1 and getattr(pk, "coset_builder", None) is not None:
        mode, keep_ = D.residency(cs, dom, ncirc)
        if mode == "cosets":
            hit = pk.__dict__.get("_multi_coset")
            if hit is None or hit.keep != keep_:
                hit = pk._multi_coset = CosetTables(pk.coset_builder, range(dom.quotient_poly_degree), keep_)
            coset_tabs = hit

    # ---- instance columns (prover.rs:85-162): zero-padded, committed, hashed but not written -------------------
    instance_dev_sets = []
    for inst in instance_sets:
        if len(inst) != cs.num_instance:
            raise ValueError("InvalidInstances")
        cols_i = []
        for vals in inst:
            if len(vals) > usable:
                raise ValueError("InstanceTooLarge")
            t = D.zeros(n)
            D.set_rows(t, 0, list(vals))
            cols_i.append(t)
        instance_dev_sets.append(cols_i)
    for P in D.msm_batch([t for cols_i in instance_dev_sets for t in cols_i], params.g_lagrange, n, 254):
        transcript.common_point(P)
    instance_polys_sets = [[D.intt(D.clone(t), dom) for t in cols_i] for cols_i in instance_dev_sets]

    # ---- advice columns: blinding rows, bounded commitments (prover.rs:255-312) ----------------------------
    # Every column is queued for upload on the copy stream first (DMA when it lives in pinned memory); the columns are
    # then blinded, measured (per-column max_bits, as the reference) and committed in small groups -- one pipelined
    # batch per group -- while the later groups are still in flight.
    # One proof over several ranks (n divisible by the group size): a rank uploads only ITS rows [lo, hi) of every
    # column over PCIe -- exactly the range its share of the commitment needs -- and the ranks complete each other's
    # columns over xGMI afterwards (parallel.allgather_rows): 1 / P of the witness per PCIe link instead of all of it.
    sharded_upload = D.group_size > 1 and n % D.group_size == 0 and n // D.group_size > bf + 1
    # ... every rank needs every row of an advice column only where something reads whole Lagrange columns: the
    # theta-compressions of lookups and shuffles (replicated).  Without them (mini-PLONK) a rank keeps its own rows -- all the
    # permutation terms of its range read -- and a column's rows follow their OWNER for the inverse transform
    # (Device.intt_columns_begin with complete = False): 1 / P of the all-gather's traffic.
    whole_advice_rows = bool(cs.lookups or cs.shuffles) or not sharded_upload
    lo_r, hi_r = 0, n
    if sharded_upload:
        from .parallel import allgather_rows, allreduce_max, msm_split_range

        lo_r, hi_r = msm_split_range(n, D.group_size, D.group_rank)
        uploads = []
        for col in advice:
            t = D.empty(n)
            with D.torch.cuda.stream(D.tstream):
                t[lo_r:hi_r] = col[lo_r:hi_r] if D.torch.is_tensor(col) else D.upload(np.ascontiguousarray(col[lo_r:hi_r]))
            uploads.append((t, None))
    else:
        # queued a few commitment groups ahead of the group being committed (queue_uploads below), not all at once: with
        # every column of a wide witness in the copy queue the FIRST group's commitment returned only when the LAST column
        # had crossed PCIe (k = 22, 64 compact columns: the GPU idle for 34 of the phase's 125 ms; tools/experiments/busy.sh)
        uploads = [None] * len(advice)
    uploads_queued = [len(uploads) if sharded_upload else 0]

    def queue_uploads(upto):
        while uploads_queued[0] < min(upto, len(uploads)):
            uploads[uploads_queued[0]] = D.upload_async(advice[uploads_queued[0]])
            uploads_queued[0] += 1

    # columns are blinded, measured and committed in groups while later uploads are still in flight; small witnesses
    # (<= 256 MiB: already on the device by the time the random polynomial is committed) go as one group -- one
    # synchronisation and one pipelined / fused batch instead of several
    # A WIDE witness (dozens of narrow columns) wants large groups: the columns of a group that share a bound are committed
    # as one fused MSM -- one sort, finish and reduce for all of them, and those latency-bound tails cost a narrow column
    # more than its accumulation -- so a group takes as many columns as cross PCIe in ~5 ms (256 MiB: 8 columns of 32-byte
    # cells at 2^20 rows, 32 compact ones), while the later groups are still in flight.
    group = len(uploads) if len(uploads) * n * 32 <= (256 << 20) else max(1, min(4, len(uploads) // 3))
    if len(uploads) >= 12:
        cell = max((8 if (not D.torch.is_tensor(c) and c.ndim == 1) else 32) for c in advice)
        group = max(group, min(len(uploads), (256 << 20) // (cell * n)))
    if os.environ.get("H2_ADVICE_GROUP"):
        group = int(os.environ["H2_ADVICE_GROUP"])
    group = max(group, 1)
    ahead = max(1, int(os.environ.get("H2_ADVICE_AHEAD", "2"))) * group
    queue_uploads(len(uploads) if len(uploads) <= group else group + ahead)
    # The vanishing argument's random polynomial (vanishing/prover.rs:40-67) and its commitment depend on nothing the
    # transcript has hashed: generated and committed NOW, while the witness columns cross PCIe on the copy stream (k = 24:
    # a 22 ms MSM under a 29 ms transfer) -- on a side stream, so that the columns that have already arrived are
    # blinded and committed next to it instead of behind it (k = 22: advice phase 10.7 -> 9.8 ms).  The commitment is
    # written where the protocol puts it, after the z's.
    # (Small witnesses too: folding it into the advice columns' batch instead was measured slower, k = 18 lookup circuit
    # 28.6 -> 30.2 ms -- the early MSM runs under the host's preparation of the blinding rows.)
    htrace("uploads queued")
    random_poly = D.empty(n)
    check(L.h2_dev_random_fr(rng.random_poly_key(), n, random_poly.data_ptr(), D.stream), "h2_dev_random_fr")
    random_commitment = D.msm_async(random_poly, params.g, n)   # collected where the transcript needs it
    D.retain([random_poly])
    # the blinding rows of every column (drawn column by column, as the reference does) go up in one copy
    blind = np.zeros((max(len(uploads), 1), n - usable, 4), dtype=np.int64)
    for ci in range(len(uploads)):
        blind[ci, :, 0] = [rng.u16() for _ in range(usable, n)]
    with D.torch.cuda.stream(D.tstream):
        blind_dev = D.torch.from_numpy(blind).to(D.dev)
    advice_dev = []
    # A witness that crosses PCIe in SEVERAL groups (the wide circuit at k = 22: 64 columns of 32-byte cells, 8 GiB, 157 ms on the
    # link against ~80 ms of narrow commitments) leaves the GPU idle half of this phase: the columns of a group are final once they
    # are blinded, so their coefficient forms and extended cosets -- needed from the quotient on -- are computed on the SIDE stream
    # group by group, under the later groups' transfers, instead of after the whole phase (H2_SIDE_GROUPS=0: as before).  Measured
    # (profiles/r6_side_groups_ab.txt): wide k = 22 447 -> 352 ms (344 from a compact witness, was 369), wide k = 20 122 -> 113,
    # mini-PLONK k = 24 185 -> 177, k = 22 51.5 -> 50.6.  A witness that fits ONE group keeps the whole-phase form further down.
    side_groups = (os.environ.get("H2_SIDE_GROUPS", "1") != "0" and os.environ.get("H2_SIDE_INTT", "1") != "0" and
                   not sharded_upload and D.group_size <= 1 and not D.force_collective and len(uploads) > group and
                   hasattr(D, "intt_on_side_stream"))
    side_parts = []
    for g0 in range(0, len(uploads), group):
        queue_uploads(g0 + group + ahead)
        cols_ = []
        for ci, (t, arrived) in enumerate(uploads[g0:g0 + group], start=g0):
            if arrived is not None:
                D.tstream.wait_event(arrived)
            if montgomery:                                       # find_max_scalar_bits needs the canonical values
                check(L.h2_dev_batch_unmont(t[lo_r:hi_r].data_ptr(), hi_r - lo_r, D.stream), "h2_dev_batch_unmont")
            if hi_r > usable:                                    # the blinding rows live in the last rank's range
                with D.torch.cuda.stream(D.tstream):
                    t[usable:] = blind_dev[ci]
            cols_.append(t)
        # The blinding rows are 16-bit values whatever the column holds (prover.rs:281-289), so the bound the reference
        # computes over the whole column is never below 16 bits -- a column of booleans or of a few tiny values then runs
        # as ONE window of 2^16 buckets with everything in its first partition.  A commitment is a sum: the usable rows
        # are committed under THEIR bound (the narrow-column shapes of the MSM) and the bf + 1 blinding rows as one more
        # (fused, few-point) MSM per group; the two points are added.
        split_tail = not sharded_upload and not (D.group_size > 1 or D.force_collective) and usable >= (1 << 12)
        m_rows = usable if split_tail else hi_r - lo_r
        htrace("advice group %d queued" % g0)
        bits_ = D.max_scalar_bits_many([t[lo_r:lo_r + m_rows] for t in cols_], m_rows)
        htrace("advice group %d bits" % g0)
        if sharded_upload:
            bits_ = allreduce_max(bits_, group=D.group, device=D.dev)       # find_max_scalar_bits over the whole column
        bits_ = [max(b, 1) for b in bits_]
        for t in cols_:
            check(L.h2_dev_batch_mont(t[lo_r:hi_r].data_ptr(), hi_r - lo_r, D.stream), "h2_dev_batch_mont")
        # ... for the columns whose usable rows are narrower than the blinding rows (and large enough for the narrow shapes
        # to matter): the others are committed whole, under the bound of the whole column
        for P in commit_lagrange_with_tail(cols_, bits_, split_tail):
            transcript.write_point(P)
        htrace("advice group %d committed" % g0)
        if sharded_upload and whole_advice_rows:
            for t in cols_:
                allgather_rows(t, lo_r, hi_r, group=D.group, stream=D.tstream)
        if side_groups:
            side_parts.append(D.intt_on_side_stream(cols_, dom, extend=D.coset_plan(dom) is None and coset_tabs is None))
        advice_dev += cols_
    del blind_dev
    del uploads
    mark("advice commit")
    theta = transcript.squeeze_challenge_scalar()
    # per-circuit state: every later phase walks `circuits` in order
    circuits = [{"advice": advice_dev[ci * nadv:(ci + 1) * nadv], "instance": instance_dev_sets[ci],
                 "instance_polys": instance_polys_sets[ci]} for ci in range(ncirc)]
    # the advice columns are final: their coefficient forms (needed from the quotient on) are computed on the side stream
    # while the lookup / permutation phases run on the compute stream -- up to k = 20, where those phases are chains of
    # small latency-bound kernels (k = 18 lookup circuit 28.9 -> 27.6 ms); at k = 22 / 24 they fill the chip themselves and
    # the transforms only take their time away (60.1 vs 60.2 ms, 210 vs 211)
    side_intt = None
    if side_parts:
        # (group by group above; the events of one stream are ordered: the last one covers them all)
        exts_ = [e for _, ext_g, _ in side_parts for e in (ext_g or [])]
        side_intt = ([p_ for polys_g, _, _ in side_parts for p_ in polys_g],
                     exts_ if all(ext_g is not None for _, ext_g, _ in side_parts) else None, side_parts[-1][2])
    elif (os.environ.get("H2_SIDE_INTT", "1") != "0" and dom.k <= int(os.environ.get("H2_SIDE_INTT_MAX_K", "20"))
            and D.group_size <= 1 and not D.force_collective):
        side_intt = D.intt_on_side_stream(advice_dev, dom, extend=D.coset_plan(dom) is None and coset_tabs is None)
    # one proof over several ranks: the advice columns' inverse transforms are dealt by column now (a rank transforms every
    # P-th column) and the coefficient vectors cross xGMI under the lookup / permutation phases that follow
    advice_arrival = None
    if D.group_size > 1:
        advice_coeffs, advice_arrival = D.intt_columns_begin(advice_dev, dom, complete=whole_advice_rows, keep=True)

    # ---- lookups: theta-compressed inputs / table, multiplicities (logup/prover.rs:63-240) ---------------------
    # One proof over several ranks with the rows dealt (Device.row_range): the compressed INPUT expressions are needed on this
    # rank's rows only (the grand sums below read them there), and the multiplicities are integer counts -- an RCCL reduction:
    # every rank counts the hits of its own input rows in the (whole, replicated) compressed table, the counters are summed by
    # ONE all-reduce per lookup and become field elements afterwards.  The shuffles' expressions likewise: rows only.
    lo_c, hi_c = D.row_range(n)
    rows_c = None if (lo_c, hi_c) == (0, n) else (lo_c, hi_c - lo_c)
    for C in circuits:
        def compress(program, C=C, rows=None):
            return _compress(D, dom, program, theta, pk.fixed_values, C["advice"], C["instance"], rows)

        C["lookups"] = []
        for table_prog, set_progs in pk.lookup_programs:
            st = {"table": compress(table_prog), "inputs": [[compress(pr, rows=rows_c) for pr in progs] for progs in set_progs]}
            flat = [c for cols_ in st["inputs"] for c in cols_]
            m = D.empty(n)
            nbytes = L.h2_logup_scratch_bytes(n)
            ptrs = (_vp * len(flat))(*[c.data_ptr() for c in flat])
            m_usable_bits = None
            if rows_c is None and hasattr(L, "h2_dev_logup_multiplicity_bits") and os.environ.get("H2_M_BITS", "1") != "0":
                # ... and the width of the largest multiplicity: a range lookup's counts are a few bits wide, nowhere near
                # the log2(rows x inputs) their sum allows -- m's commitment then takes the narrow-column shapes of the MSM
                got_bits = ctypes.c_uint32(0)
                check(L.h2_dev_logup_multiplicity_bits(st["table"].data_ptr(), ptrs, len(flat), usable, n, m.data_ptr(),
                                                       D.scratch(nbytes).data_ptr(), nbytes, ctypes.byref(got_bits), D.stream),
                      "h2_dev_logup_multiplicity_bits")
                m_usable_bits = max(int(got_bits.value), 1)
            elif rows_c is None:
                check(L.h2_dev_logup_multiplicity(st["table"].data_ptr(), ptrs, len(flat), usable, n, m.data_ptr(),
                                                  D.scratch(nbytes).data_ptr(), nbytes, D.stream), "h2_dev_logup_multiplicity")
            else:
                from .parallel import allreduce_counts

                with D.torch.cuda.stream(D.tstream):
                    counts = D.torch.empty(n + 1, dtype=D.torch.int32, device=D.dev)
                check(L.h2_dev_logup_counts(st["table"].data_ptr(), ptrs, len(flat), usable, n, lo_c, hi_c, counts.data_ptr(),
                                            D.scratch(nbytes).data_ptr(), nbytes, D.stream), "h2_dev_logup_counts")
                missing = allreduce_counts(counts, group=D.group, stream=D.tstream)
                if missing:
                    raise ValueError("logup: %d input value(s) are missing from the table" % missing)
                check(L.h2_dev_logup_emit(counts.data_ptr(), usable, n, m.data_ptr(), D.stream), "h2_dev_logup_emit")
            D.set_rows(m, usable, [rng.u16() for _ in range(usable, n)])
            st["m"], st["m_bits"], st["m_usable_bits"] = m, max(16, (usable * len(flat)).bit_length()), m_usable_bits
            C["lookups"].append(st)
        # ---- shuffles: compressed expressions (shuffle/prover.rs:40-80) ------------------------------------------
        C["shuffles"] = [[(compress(ip, rows=rows_c), compress(sp, rows=rows_c)) for ip, sp in group] for group in pk.shuffle_programs]
    all_lookups = [st for C in circuits for st in C["lookups"]]
    if all_lookups and all(st["m_usable_bits"] is not None for st in all_lookups):
        split_m = not (D.group_size > 1 or D.force_collective) and usable >= (1 << 12)
        # (committed whole, a column's bound covers its 16-bit blinding rows too)
        m_bits_ = [st["m_usable_bits"] if split_m else max(st["m_usable_bits"], 16) for st in all_lookups]
        for P in commit_lagrange_with_tail([st["m"] for st in all_lookups], m_bits_, split_m):
            transcript.write_point(P)
    elif all_lookups:
        for P in D.msm_batch([st["m"] for st in all_lookups], params.g_lagrange, n, max(st["m_bits"] for st in all_lookups)):
            transcript.write_point(P)
    mark("lookups compress")
    beta = transcript.squeeze_challenge_scalar()
    gamma = transcript.squeeze_challenge_scalar()

    # ---- permutation grand products (permutation/prover.rs:47-165) -----------------------------------
    chunk = cs.degree() - 2
    cols = cs.perm_columns
    nsets = (len(cols) + chunk - 1) // chunk
    # Everything that has to be inverted before the grand products / sums can run -- the permutation denominators of
    # every set, (beta + f) of every lookup input and table, the shuffle products -- depends only on beta and gamma: it
    # is laid out in ONE buffer (per circuit instance) and inverted by ONE batch inversion, so the inversion's serial
    # a^(r-2) chain (~0.3 ms of pure latency per call) is paid once per proof instead of once per column.
    # One proof over several ranks: every pass of this phase -- the numerator / denominator terms, the batch inversion, the
    # products, the scans -- runs on this rank's rows [lo_s, hi_s) only (a prefix scan exchanges one field element per rank,
    # Device.prefix_scan); the range is exactly what the rank's share of the z commitments consumes, and the ranks complete
    # each other's z columns over xGMI before the inverse transforms (Device.gather_rows).
    lo_s, hi_s = D.row_range(n)
    m_s = hi_s - lo_s
    rows = lambda t: t[lo_s:hi_s]  # noqa: E731
    omega_lo = pow(dom.omega, lo_s, R_MOD)
    fused_perm = getattr(D, "permutation_product", None) if (lo_s, hi_s) == (0, n) else None
    fused_lookup = getattr(D, "logup_grand_sum", None) if (lo_s, hi_s) == (0, n) else None
    for C in circuits:
        lookups, shuffles = C["lookups"], C["shuffles"]
        colvals = {"advice": C["advice"], "fixed": pk.fixed_values, "instance": C["instance"]}
        # a device whose vectors live on the HOST makes one call per set instead (h2_permutation_product: terms, inversion,
        # product and scan without num / den crossing PCIe around every step): its sets take no slot here
        pslots = 0 if fused_perm else nsets
        lslots = 0 if fused_lookup else sum(len(cols_in) for st in lookups for cols_in in st["inputs"]) + len(lookups)
        slots = pslots + lslots + len(shuffles)
        nums = D.empty(max(pslots, 1) * m_s)
        inv = D.empty(max(slots, 1) * m_s)
        slot = lambda i, inv=inv: inv[i * m_s:(i + 1) * m_s]  # noqa: E731
        for k_, si in enumerate(range(0, len(cols) if pslots else 0, chunk)):
            for ci in range(si, min(si + chunk, len(cols))):
                values = colvals[cols[ci][0]][cols[ci][1]]
                check(L.h2_dev_permutation_terms(nums[k_ * m_s:].data_ptr(), slot(k_).data_ptr(), rows(values).data_ptr(),
                                                 rows(pk.sigma_values[ci]).data_ptr(), m_s, _fr(beta), _fr(gamma),
                                                 _fr(pow(DELTA, ci, R_MOD) * omega_lo), _fr(dom.omega), 1 if ci == si else 0,
                                                 D.stream), "h2_dev_permutation_terms")
        at = pslots
        for st in lookups:
            if fused_lookup:            # (host vectors: h2_logup_grand_sum inverts on the device, set by set, further down)
                continue
            st["inv_inputs"] = []
            for cols_in in st["inputs"]:
                st["inv_inputs"].append([])
                for col in cols_in:                                     # beta + f_i
                    st["inv_inputs"][-1].append(D.eval_op(1, slot(at), rows(col), c=beta, size=m_s))          # H2_OP_SUM_C
                    at += 1
            st["inv_table"] = D.eval_op(1, slot(at), rows(st["table"]), c=beta, size=m_s)  # beta + t
            at += 1
        C["shuffle_inv"] = []
        for group in shuffles:                                          # prod_i (beta^(i+1) + shuffle_i)
            dst = slot(at)
            for i, (_, shf) in enumerate(group):
                if i == 0:
                    D.eval_op(1, dst, rows(shf), c=beta, size=m_s)
                else:
                    D.eval_op(6, dst, rows(shf), dst, c=pow(beta, i + 1, R_MOD), size=m_s)       # H2_OP_LCBETA: (l + c) * r
            C["shuffle_inv"].append(dst)
            at += 1
        assert at == slots
        if slots:
            check(L.h2_dev_batch_invert(inv.data_ptr(), D.empty(slots * m_s).data_ptr(), slots * m_s, D.stream), "h2_dev_batch_invert")
        if pslots:
            D.eval_op(3, nums, nums, inv[:nsets * m_s])                                     # H2_OP_MUL over all sets
        C["nums"], C["inv"] = nums, inv

    # ---- permutation grand products (permutation/prover.rs:89-165), circuit by circuit -------------------------
    blinding = []         # (z, first blinding row, values): drawn in the reference's order, written in one copy below
    for C in circuits:
        C["z"], last_z = [], 1
        colvals = {"advice": C["advice"], "fixed": pk.fixed_values, "instance": C["instance"]}
        for k_ in range(nsets):
            if fused_perm:
                cis = range(k_ * chunk, min((k_ + 1) * chunk, len(cols)))
                z, last_z = fused_perm([colvals[cols[ci][0]][cols[ci][1]] for ci in cis], [pk.sigma_values[ci] for ci in cis], n,
                                       beta, gamma, pow(DELTA, k_ * chunk, R_MOD), dom.omega, last_z, usable)
            else:
                z, last_z = D.prefix_scan(C["nums"][k_ * m_s:(k_ + 1) * m_s], n, last_z, True, usable)
            blinding.append((z, n - bf, [rng.fr() for _ in range(bf)]))
            C["z"].append(z)
        del C["nums"]
    num = D.empty(m_s)
    # ---- lookup grand sums (logup/prover.rs:243-415; blinding prover.rs:446-465) -------------------------------
    for C in circuits:
        for st in C["lookups"]:
            st["z"] = []
            last = 0
            for si, cols_in in enumerate(st["inputs"] if fused_lookup else ()):
                z, last = fused_lookup(cols_in, st["table"] if si == 0 else None, st["m"] if si == 0 else None, n, beta, last, usable)
                blinding.append((z, n - bf, [rng.fr() for _ in range(bf)]))
                st["z"].append(z)
            for si, inverted in enumerate(() if fused_lookup else st["inv_inputs"]):
                src = inverted[0]                                       # sum_i 1 / (beta + f_i)
                for other in inverted[1:]:
                    src = D.eval_op(2, num, src, other, size=m_s)       # H2_OP_SUM
                if si == 0:                                             # - m / (beta + t)
                    D.eval_op(3, st["inv_table"], st["inv_table"], rows(st["m"]), size=m_s)
                    src = D.eval_op(4, num, src, st["inv_table"], size=m_s)       # H2_OP_SUB
                z, last = D.prefix_scan(src, n, last, False, usable)
                blinding.append((z, n - bf, [rng.fr() for _ in range(bf)]))
                st["z"].append(z)
            if last != 0:
                raise ValueError("lookup grand sum does not return to zero")   # sanity-checks feature of the reference
            if not fused_lookup:
                del st["inv_inputs"], st["inv_table"]
    # ---- shuffle products (shuffle/prover.rs:82-150; blinding prover.rs:512-530) -------------------------------
    for C in circuits:
        C["shuffle_z"] = []
        for group, inverted in zip(C["shuffles"], C["shuffle_inv"]):
            for i, (inp, _) in enumerate(group):
                D.eval_op(6, inverted, rows(inp), inverted, c=pow(beta, i + 1, R_MOD), size=m_s)
            z, closing = D.prefix_scan(inverted, n, 1, True, usable)
            if closing != 1:
                raise ValueError("shuffle product does not return to one")
            blinding.append((z, n - bf, [rng.fr() for _ in range(bf)]))
            C["shuffle_z"].append(z)
        del C["inv"], C["shuffle_inv"]
    del num
    D.set_rows_many(blinding)
    # commit_lagrange_and_ifft (poly/commitment.rs:144-197) for every z, in the transcript's order: the permutation
    # products of every circuit, then the lookup sums of every circuit, then the shuffle products (prover.rs:595-625)
    all_z = ([z for C in circuits for z in C["z"]] + [z for C in circuits for st in C["lookups"] for z in st["z"]] +
             [z for C in circuits for z in C["shuffle_z"]])
    # (a device whose vectors live on the HOST does both in one call per column, sharing the one upload:
    # gpu_multiexp_bound_and_fft, arithmetic.rs:375-410)
    commit_ifft = getattr(D, "commit_lagrange_and_ifft", None) if (lo_s, hi_s) == (0, n) and not D.force_collective else None
    z_commitments = commit_ifft(all_z, params.g_lagrange, dom) if commit_ifft else D.msm_batch(all_z, params.g_lagrange, n, 254)
    for P in z_commitments:
        transcript.write_point(P)
    # (one proof over several ranks: every rank computed its own rows of the product columns; a column's rows go to the
    # rank that transforms it, and the coefficient vectors travel while the advice columns are taken to their cosets)
    z_arrival = None
    if not commit_ifft:
        _, z_arrival = D.intt_columns_begin(all_z, dom, complete=False)
    _, m_arrival = D.intt_columns_begin([st["m"] for C in circuits for st in C["lookups"]], dom, complete=True)
    for C in circuits:
        C["z_polys"] = C["z"]                                       # (transformed in place, sixteen to a launch)
        for st in C["lookups"]:
            st["z_polys"], st["m_poly"] = st["z"], st["m"]
            del st["table"], st["inputs"]
        C["shuffle_polys"] = C["shuffle_z"]
        del C["shuffles"]
    mark("permutation")
    htrace("before random_commitment.result")
    transcript.write_point(random_commitment.result())
    y = transcript.squeeze_challenge_scalar()
    htrace("y squeezed")

    # ---- h(X): advice to coefficient form, extended cosets, the fused evaluator --------------------------
    if side_intt is not None:
        polys_, ext_, done_ = side_intt
        D.tstream.wait_event(done_)
        for ci, C in enumerate(circuits):
            C["advice_polys"] = polys_[ci * nadv:(ci + 1) * nadv]
            C["advice_extended"] = ext_[ci * nadv:(ci + 1) * nadv] if ext_ is not None else None
            C["advice"] = None                                       # the Lagrange values are not needed again
        del advice_dev
    elif advice_arrival is not None:
        for ci, C in enumerate(circuits):
            C["advice_polys"] = advice_coeffs[ci * nadv:(ci + 1) * nadv]
            C["advice"] = None
        del advice_dev
    else:
        for C in circuits:
            C["advice_polys"] = D.intt_many(C["advice"], dom)           # in place: the Lagrange values are not needed again
    # the coefficient forms of the witness and of the product columns are final: evaluator, evaluations and openings read them
    D.retain([t for C in circuits for t in list(C["advice_polys"]) + list(C["instance_polys"]) + list(C["z_polys"]) +
              [z_ for st in C["lookups"] for z_ in st["z_polys"]] + [st["m_poly"] for st in C["lookups"]] + list(C["shuffle_polys"])])
    g = pk.graph
    plan = D.coset_plan(dom)
    if D.group_size <= 1:                      # on one device the proving key decides which tables exist
        if coset_tabs is None:
            plan = None
        else:
            plan = (dom.quotient_poly_degree, 1, sorted(coset_tabs))
    # Several circuits share one quotient: the reference keeps folding `value = value * y + term` from one circuit into
    # the next (plonk/evaluation.rs:839-1100), i.e. h = sum_i y^(T (N - 1 - i)) h_i with T terms per circuit and h_i the
    # fold of circuit i alone -- each circuit runs through the evaluator on its own and the results are combined.
    terms_per_circuit = (len(pk.value_parts) + (2 * nsets + 1 if nsets else 0) +
                         sum(2 * len(st["z_polys"]) + 1 for st in circuits[0]["lookups"]) + 3 * len(circuits[0]["shuffle_polys"]))
    y_step = pow(y, terms_per_circuit, R_MOD)

    def evaluate_quotient(points_of, tables, k_domain, zeta_, omega_, size, rows=None, fused=None):
        """`rows` = (first, count): only these rows of the domain are evaluated (and valid in the result).  `fused` (host
        vectors, one circuit instance): the device's h2_quotient_poly_coeff -- the result is h(X) in COEFFICIENT form"""
        total = None
        lo_, cnt_ = rows if rows is not None else (0, size)
        for C in circuits:
            h_c = evaluate_quotient_of(C, points_of, tables, k_domain, zeta_, omega_, size, rows, fused)
            if total is None:
                total = h_c
            else:                                                                           # H2_OP_LCTHETA: l * c + r
                D.eval_op(5, total[lo_:lo_ + cnt_], total[lo_:lo_ + cnt_], h_c[lo_:lo_ + cnt_], c=y_step, size=cnt_)
        return total

    def evaluate_quotient_of(C, points_of, tables, k_domain, zeta_, omega_, size, rows=None, fused=None):
        """the fused evaluator over one evaluation domain: `points_of` maps a list of coefficient vectors to their values there"""
        lookups = C["lookups"]
        if points_of is None:
            # the cuda shape of Evaluator::evaluate_h (plonk/evaluation.rs:1229-1241): COEFFICIENT forms in, the extended
            # values of the numerator out, one h2_evaluate_h_coeff call (host slices: halo2-gpu-specific_amd/host_api.py)
            b = ev.Builder().build(
                k=dom.k, extended_k=k_domain, blinding_factors=bf, chunk_len=chunk,
                constants=np.array([fr_to_mont_limbs(c) for c in g.constants], dtype=np.uint64), rotations=g.rotations,
                calculations=g.calculations, value_parts=pk.value_parts, lookups=pk.lookup_calcs, shuffles=pk.shuffle_calcs,
                fixed=[t.data_ptr() for t in pk.fixed_polys], advice=[t.data_ptr() for t in C["advice_polys"]],
                instance=[t.data_ptr() for t in C["instance_polys"]],
                l0=pk.l0_poly.data_ptr(), l_last=pk.l_last_poly.data_ptr(), l_active_row=tables["l_active_row"].data_ptr(),
                perm_z=[t.data_ptr() for t in C["z_polys"]], perm_columns=[(_ANY[kd], i) for kd, i in cols],
                perm_sigma=[t.data_ptr() for t in pk.sigma_polys],
                lookup_z=[t.data_ptr() for st in lookups for t in st["z_polys"]],
                lookup_m=[st["m_poly"].data_ptr() for st in lookups],
                shuffle_z=[t.data_ptr() for t in C["shuffle_polys"]],
                y=fr_to_mont_limbs(y), beta=fr_to_mont_limbs(beta), gamma=fr_to_mont_limbs(gamma), theta=fr_to_mont_limbs(theta),
                delta=fr_to_mont_limbs(DELTA), zeta=fr_to_mont_limbs(zeta_), extended_omega=fr_to_mont_limbs(omega_),
                flags=0 if pk.evalh_stats else ev.EVALH_INTERPRET)
            if fused:           # ... divided by the vanishing polynomial and taken back to coefficients in the same call
                out = fused(b.desc, dom, pk.t_evaluations)
                mark("evaluate_h")
                return out
            out = D.empty(size)
            check(L.h2_evaluate_h_coeff(ctypes.byref(b.desc), out.data_ptr()), "h2_evaluate_h_coeff")
            mark("evaluate_h")
            return out
        pre = C.get("advice_extended") if size == en else None    # already extended on the side stream (small proofs)
        # every coefficient vector of this circuit instance that the evaluator reads, taken to the evaluation domain as ONE
        # list (the coset route transforms them sixteen to a launch)
        groups = [[] if pre is not None else list(C["advice_polys"]), list(C["instance_polys"]), list(C["z_polys"]),
                  [t for st in lookups for t in st["z_polys"]], [st["m_poly"] for st in lookups], list(C["shuffle_polys"])]
        if advice_arrival is not None:
            # the advice columns' coefficients have been travelling since the commit phase; the product columns' are still on
            # their way: the advice columns go to the evaluation domain first
            advice_arrival.wait()
            first = points_of(groups[0])
            for arrival in (z_arrival, m_arrival):
                if arrival is not None:
                    arrival.wait()
            flat = first + points_of([t for grp in groups[1:] for t in grp])
        else:
            flat = points_of([t for grp in groups for t in grp])
        cut, at = [], 0
        for grp in groups:
            cut.append(flat[at:at + len(grp)])
            at += len(grp)
        advice_cosets = pre if pre is not None else cut[0]
        instance_cosets, z_cosets, lookup_z_cosets, lookup_m_cosets, shuffle_cosets = cut[1:]
        del flat, cut
        htrace("points_of done (launched)")
        mark("cosets")
        b = ev.Builder().build(
            k=dom.k, extended_k=k_domain, blinding_factors=bf, chunk_len=chunk,
            constants=np.array([fr_to_mont_limbs(c) for c in g.constants], dtype=np.uint64), rotations=g.rotations,
            calculations=g.calculations, value_parts=pk.value_parts, lookups=pk.lookup_calcs, shuffles=pk.shuffle_calcs,
            fixed=[t.data_ptr() for t in tables["fixed"]], advice=[t.data_ptr() for t in advice_cosets],
            instance=[t.data_ptr() for t in instance_cosets],
            l0=tables["l0"].data_ptr(), l_last=tables["l_last"].data_ptr(), l_active_row=tables["l_active_row"].data_ptr(),
            perm_z=[t.data_ptr() for t in z_cosets], perm_columns=[(_ANY[kd], i) for kd, i in cols],
            perm_sigma=[t.data_ptr() for t in tables["sigma"]],
            lookup_z=[t.data_ptr() for t in lookup_z_cosets], lookup_m=[t.data_ptr() for t in lookup_m_cosets],
            shuffle_z=[t.data_ptr() for t in shuffle_cosets],
            y=fr_to_mont_limbs(y), beta=fr_to_mont_limbs(beta), gamma=fr_to_mont_limbs(gamma), theta=fr_to_mont_limbs(theta),
            delta=fr_to_mont_limbs(DELTA), zeta=fr_to_mont_limbs(zeta_), extended_omega=fr_to_mont_limbs(omega_),
            flags=0 if pk.evalh_stats else ev.EVALH_INTERPRET,
            row_begin=rows[0] if rows is not None else 0, row_count=rows[1] if rows is not None else 0)
        htrace("descriptor built")
        out = D.empty(size)
        htrace("output allocated")
        check(L.h2_dev_evaluate_h(ctypes.byref(b.desc), out.data_ptr(), D.stream), "h2_dev_evaluate_h")
        htrace("h2_dev_evaluate_h returned")
        mark("evaluate_h")
        return out

    if plan is None:
        # ---- one device: the whole extended domain at once ------------------------------------------------------
        tables = {"fixed": pk.fixed_cosets, "sigma": pk.sigma_cosets, "l0": pk.l0, "l_last": pk.l_last,
                  "l_active_row": pk.l_active_row}
        from_coeffs = getattr(D, "quotient_from_coeffs", False)
        # (host vectors, one circuit instance: the three steps in one call, the 2^extended_k values never leave the device)
        fused = getattr(D, "quotient_poly_coeff", None) if from_coeffs and len(circuits) == 1 else None
        h = evaluate_quotient(None if from_coeffs else (lambda ts: D.coeffs_to_extended(ts, dom)), tables, ek, ZETA,
                              dom.extended_omega, en, fused=fused)
        if not fused:
            # vanishing construct: divide, back to coefficients (vanishing/prover.rs:69-112)
            check(L.h2_dev_divide_by_vanishing_poly(h.data_ptr(), en, pk.t_evaluations.data_ptr(), len(dom.t_evaluations),
                                                    D.stream), "h2_dev_divide_by_vanishing_poly")
            D.extended_to_coeff(h, dom)
        pieces = [h[i * n:(i + 1) * n] for i in range(dom.quotient_poly_degree)]
    else:
        # ---- one proof over several ranks: the extended domain by coset.  On coset j (points
        # g_j w^i, extended indices c i + j) every rotation stays inside the coset, the vanishing polynomial is the
        # constant gamma_j - 1 (gamma_j = g_j^n) and h(X) = sum_m X^(n m) h_m(X) reads P_j(X) = sum_m gamma_j^m h_m(X): the
        # evaluator runs on n points per coset with zeta := g_j, extended_omega := omega, extended_k := k; the inverse
        # coset transform gives P_j; one n-vector per coset is exchanged; the pieces are h_m = sum_j Vinv[m][j] P_j.
        from .parallel import coset_unmix_matrix, exchange_cosets, scatter_cosets

        c, shards, owned = plan
        mine = {}
        # More ranks than cosets (a multiple G of them): the G ranks of a coset share its work instead of repeating it --
        # every G-th column's coset transform each, row slices exchanged, the evaluator on n / G rows each, the quotient's
        # rows all-gathered inside the group (parallel.exchange_row_slices)
        sub = D.coset_rank_group(c)
        used_rots = list(g.rotations) + ([0, 1, last_rot] if (nsets or circuits[0]["lookups"] or circuits[0]["shuffle_polys"]) else [0])
        halo = (max(0, -min(used_rots)), max(0, max(used_rots)))
        if sub is not None and (n % sub[1] or n // sub[1] <= halo[0] + halo[1] + 1):
            sub = None
        for j in owned:
            g_j = ZETA * pow(dom.extended_omega, j, R_MOD) % R_MOD
            if sub is None:
                h_j = evaluate_quotient(lambda ts, j=j: D.coeffs_to_coset(ts, dom, j), coset_tabs[j], dom.k, g_j, dom.omega, n)
                D.eval_op(0, h_j, h_j, c=dom.t_evaluations[j % len(dom.t_evaluations)])      # H2_OP_MUL_C: / (gamma_j - 1)
            else:
                from .parallel import allgather_rows

                m_g = n // sub[1]
                lo_g = sub[2] * m_g
                h_j = evaluate_quotient(lambda ts, j=j: D.coeffs_to_coset_rows(ts, dom, j, sub, halo), coset_tabs[j], dom.k, g_j,
                                        dom.omega, n, rows=(lo_g, m_g))
                D.eval_op(0, h_j[lo_g:lo_g + m_g], h_j[lo_g:lo_g + m_g], c=dom.t_evaluations[j % len(dom.t_evaluations)], size=m_g)
                allgather_rows(h_j, lo_g, lo_g + m_g, group=sub[0], stream=D.tstream)
            mine[j] = D.coset_to_coeff(h_j, dom, j)
        # Everything after the quotient -- the un-mixing, the h pieces' commitments, the evaluations, the multiopen argument --
        # works on coefficient RANGES (Device.row_range): a rank needs only its own n / P coefficients of every coset
        # polynomial, so their owners scatter slices (c x n / P x 32 B per rank) instead of broadcasting whole vectors.
        if D.group_size > 1 and D.row_range(n) != (0, n):
            polys_j = scatter_cosets(mine, c, shards, *D.row_range(n), group=D.group, stream=D.tstream)
        elif D.group_size > 1:
            polys_j = exchange_cosets(mine, c, shards, group=D.group, stream=D.tstream)
        else:
            polys_j = [mine[j] for j in range(c)]
        gammas = [pow(ZETA * pow(dom.extended_omega, j, R_MOD) % R_MOD, n, R_MOD) for j in range(c)]
        unmix = coset_unmix_matrix(gammas, dom.quotient_poly_degree)
        pieces = [D.lincomb_range(D.empty(n), polys_j, row, n) for row in unmix]
        del polys_j, mine
        coset_tabs.trim()
    mark("vanishing transforms")
    for P in D.msm_batch(pieces, params.g, n, 254):
        transcript.write_point(P)
    x = transcript.squeeze_challenge_scalar()
    xn = pow(x, n, R_MOD)
    mark("vanishing construct")

    # ---- evaluations (prover.rs:700-790): every (polynomial, point) pair of the proof in one batched launch ------
    # h(X) = sum_i x^(n i) piece_i (vanishing/prover.rs:120-124)
    D.retain(pieces)
    h_poly = D.lincomb_range(D.empty(n), pieces, [pow(xn, i, R_MOD) for i in range(len(pieces))], n)
    D.retain([h_poly])
    wanted, written = [], []            # (key, poly, rotation); the subset the transcript receives, in its order

    def want(key, poly, rot, write=True):
        if (key, rot) not in [(k_, r_) for k_, _, r_ in wanted]:
            wanted.append((key, poly, rot))
        if write:
            written.append((key, rot))

    def want_set_evals(name, polys_):
        for i, p in enumerate(polys_):
            want((name, i), p, 0)
            want((name, i), p, 1)
            if i + 1 < len(polys_):
                want((name, i), p, last_rot)

    # evaluations in the transcript's order (prover.rs:704-790): instance columns of every circuit, advice columns of
    # every circuit, fixed, the random polynomial, sigma, then per circuit the permutation / lookup / shuffle products
    for ci, C in enumerate(circuits):
        for c, rot in cs.instance_queries:
            want(("instance", ci, c), C["instance_polys"][c], rot)
    for ci, C in enumerate(circuits):
        for c, rot in cs.advice_queries:
            want(("advice", ci, c), C["advice_polys"][c], rot)
    for c, rot in cs.fixed_queries:
        want(("fixed", c), pk.fixed_polys[c], rot)
    want(("random",), random_poly, 0)
    for i, p in enumerate(pk.sigma_polys):
        want(("sigma", i), p, 0)
    for ci, C in enumerate(circuits):
        want_set_evals("z%d" % ci, C["z_polys"])
    for ci, C in enumerate(circuits):
        for li, st in enumerate(C["lookups"]):                         # logup/prover.rs:419-446
            want(("lookup_m", ci, li), st["m_poly"], 0)
            want_set_evals("lookup_z%d_%d" % (ci, li), st["z_polys"])
    for ci, C in enumerate(circuits):
        for i, p in enumerate(C["shuffle_polys"]):                     # shuffle/prover.rs:196-212
            want(("shuffle_z", ci, i), p, 0)
            want(("shuffle_z", ci, i), p, 1)
    want(("h",), h_poly, 0, write=False)                               # opened, not written (vanishing/prover.rs:140-155)
    values = D.eval_polynomial_ranges([p for _, p, _ in wanted], n, [dom.rotate_omega(x, r) for _, _, r in wanted])
    evals = {(key, rot): v for (key, _, rot), v in zip(wanted, values)}
    for key, rot in written:
        transcript.write_scalar(evals[(key, rot)])
    mark("evaluations")

    def evaluate(key, poly, rot):
        return evals[(key, rot)]

    # ---- multiopen query list in the reference's order (prover.rs:792-840): per circuit its instance, advice,
    # permutation, lookup and shuffle openings; then fixed, sigma, h and the random polynomial -----------------------
    polys, queries = {}, []

    def query(key, poly, rot):
        polys[key] = poly
        queries.append((key, rot, dom.rotate_omega(x, rot), evaluate(key, poly, rot)))

    def open_sets(name, polys_):
        for i, p in enumerate(polys_):
            query((name, i), p, 0)
            query((name, i), p, 1)
        for i in reversed(range(len(polys_) - 1)):
            query((name, i), polys_[i], last_rot)

    for ci, C in enumerate(circuits):
        for c, rot in cs.instance_queries:
            query(("instance", ci, c), C["instance_polys"][c], rot)
        for c, rot in cs.advice_queries:
            query(("advice", ci, c), C["advice_polys"][c], rot)
        open_sets("z%d" % ci, C["z_polys"])
        for li, st in enumerate(C["lookups"]):
            query(("lookup_m", ci, li), st["m_poly"], 0)
            open_sets("lookup_z%d_%d" % (ci, li), st["z_polys"])
        for i, p in enumerate(C["shuffle_polys"]):
            query(("shuffle_z", ci, i), p, 0)
            query(("shuffle_z", ci, i), p, 1)
    for c, rot in cs.fixed_queries:
        query(("fixed", c), pk.fixed_polys[c], rot)
    for i, p in enumerate(pk.sigma_polys):
        query(("sigma", i), p, 0)
    query(("h",), h_poly, 0)
    query(("random",), random_poly, 0)
    (_gwc if use_gwc else _shplonk)(D, params, transcript, queries, polys, n)
    D.release_retained()
    mark("multiopen")
    if _host_trace:
        import sys

        base = _host_trace[0][1]
        sys.stderr.write("host trace (ms): " + ", ".join("%s %.2f" % (nm, (t - base) * 1e3) for nm, t in _host_trace) + "\n")
    if timings is not None:
        for (_, t0), (name, t1) in zip(marks, marks[1:]):
            timings[name] = timings.get(name, 0.0) + (t1 - t0)
        if D.group_size > 1:
            # per phase: seconds / bytes / calls of this rank's collectives (parallel.COMM_TRACE), next to `timings`
            D.last_comm = _par.comm_trace_end()
    return transcript.finalize()


def _gwc(D, params, transcript, queries, polys, n):
    """poly/multiopen/gwc/prover.rs:20-175: per opening point, batch = sum_i v^(m-1-i) p_i, witness =
    (batch - batch(z)) / (X - z).  The reference's cuda branch (:57-151) uploads every p_i again for its eval_mul_c /
    eval_sum pair; here they never left the device and one lincomb forms the batch.  One proof over several ranks: every
    vector pass runs on the rank's coefficient range (Device.*_range(s)); the commitments are range-split anyway."""
    v = transcript.squeeze_challenge_scalar()
    quotient_sum = getattr(D, "quotient_sum", None) if D.row_range(n) == (0, n) else None
    groups = {}
    for qu in queries:
        groups.setdefault(qu[1], []).append(qu)          # BTreeMap<Rotation, Vec<Q>> (gwc.rs:40-49)
    witnesses = []
    for rot in sorted(groups):
        group = groups[rot]
        z, m = group[0][2], len(group)
        vpow = [pow(v, m - 1 - i, R_MOD) for i in range(m)]
        at_z = sum(c * e for c, (_, _, _, e) in zip(vpow, group)) % R_MOD                       # = batch(z)
        if quotient_sum:                   # (host vectors: the fold, the subtraction and the division in one call)
            witnesses.append(quotient_sum(n, [([polys[key] for key, _, _, _ in group], vpow, [at_z], [z])])[0])
            continue
        batch = D.lincomb_range(D.empty(n), [polys[key] for key, _, _, _ in group], vpow, n)
        D.sub_low_range(batch, [at_z], n)
        witnesses.append(D.kate_division_ranges(batch, n, z, D.empty(n)))
    for P in D.msm_batch(witnesses, params.g, n, 254):
        transcript.write_point(P)


def _shplonk(D, params, transcript, queries, polys, n):
    """poly/multiopen/shplonk/prover.rs:89-225.  Every fold `acc * c + p` of the reference is a linear combination
    with powers of the challenge; the device computes each one in a single pass (h2_dev_lincomb) and the host
    adjusts the <= 3 low coefficients the low-degree equivalents r_i(X) touch.  One proof over several ranks: every
    vector pass runs on the rank's coefficient range; a Kate division exchanges one field element per rank."""
    y = transcript.squeeze_challenge_scalar()
    sets, super_points = _intermediate_sets(queries)
    for rs in sets:
        rs["low"] = [_lagrange_interpolate(rs["points"], e) for _, e in rs["commitments"]]
    v = transcript.squeeze_challenge_scalar()
    R = len(sets)
    vpow = [pow(v, R - 1 - r, R_MOD) for r in range(R)]
    # a device whose vectors live on the HOST computes the whole sum in one call (h2_quotient_sum: the combinations, the
    # subtractions and the synthetic divisions stay on the device, h(X) crosses PCIe once)
    quotient_sum = getattr(D, "quotient_sum", None) if D.row_range(n) == (0, n) else None
    # quotient contribution of every rotation set: (sum_i y^(m-1-i) (p_i - r_i)) / prod (X - point)
    quotients, fused_sets = [], []
    ping, pong = D.empty(n), D.empty(n)
    for r, rs in enumerate(sets):
        m = len(rs["commitments"])
        ypow = [pow(y, m - 1 - i, R_MOD) for i in range(m)]
        width = len(rs["points"])
        low = [sum(ypow[i] * rs["low"][i][j] for i in range(m)) % R_MOD for j in range(width)]
        if quotient_sum:        # v^(R-1-r) goes into the set's coefficients: division is linear, the field elements are the same
            fused_sets.append(([polys[key] for key, _ in rs["commitments"]], [vpow[r] * c % R_MOD for c in ypow],
                               [vpow[r] * c % R_MOD for c in low], rs["points"]))
            continue
        n_x = D.lincomb_range(D.empty(n), [polys[key] for key, _ in rs["commitments"]], ypow, n)
        D.sub_low_range(n_x, low, n)
        cur = n_x
        for pt in rs["points"]:
            nxt = ping if cur is not ping else pong
            D.kate_division_ranges(cur, n, pt, nxt)
            cur = nxt
        quotients.append(D.clone(cur))
    if quotient_sum:
        h_x, _ = quotient_sum(n, fused_sets)
    else:
        h_x = D.lincomb_range(D.empty(n), quotients, vpow, n)
    del quotients
    transcript.write_point(D.msm(h_x, params.g, n))
    u = transcript.squeeze_challenge_scalar()
    zt_eval = _vanishing(super_points, u)
    # linearisation: l(X) = sum_r v^(R-1-r) z_r sum_i y^(m-1-i) (p_i - r_i(u)) - zt(u) h(X), then / (X - u) / z_0
    z_diffs = [_vanishing([p for p in super_points if p not in rs["points"]], u) for rs in sets]
    scale = _inv(z_diffs[0])
    lin_polys, lin_coeffs, const = [], [], 0
    for r, rs in enumerate(sets):
        m = len(rs["commitments"])
        for i, (key, _) in enumerate(rs["commitments"]):
            c = vpow[r] * z_diffs[r] % R_MOD * pow(y, m - 1 - i, R_MOD) % R_MOD * scale % R_MOD
            lin_polys.append(polys[key])
            lin_coeffs.append(c)
            const = (const + c * _horner(rs["low"][i], u)) % R_MOD
    lin_polys.append(h_x)
    lin_coeffs.append((-zt_eval * scale) % R_MOD)
    if quotient_sum:
        pong, rem = quotient_sum(n, [(lin_polys, lin_coeffs, [const], [u])], remainders=True)
        if rem[0] != 0:
            raise AssertionError("shplonk: l(u) != 0")
    else:
        l_x = D.lincomb_range(ping, lin_polys, lin_coeffs, n)
        D.sub_low_range(l_x, [const], n)
        if D.eval_polynomial_ranges([l_x], n, [u])[0] != 0:
            raise AssertionError("shplonk: l(u) != 0")   # the reference's must_be_zero (prover.rs:213-214)
        D.kate_division_ranges(l_x, n, u, pong)
    transcript.write_point(D.msm(pong, params.g, n))
