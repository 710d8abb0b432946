"""Multi-GPU sharding of the hot path: one process per GPU (torch.distributed; backend "nccl" = RCCL
over xGMI on the GPU box, "gloo" in the CPU tests).

The path shards over independent units; DESIGN.md section 6 has the accounting:
  * every MSM of a proof is split into contiguous ceil(n / P) chunks (gpu_multiexp_bound,
    arithmetic.rs:425-435); each rank holds one partial G1 point (96 B) per MSM.  EC addition is not an
    RCCL reduction operator, so the exchange is an all-gather of P x count x 96 B followed by a fold -- on the
    device (h2_dev_g1_fold) under RCCL, on the host under gloo -- latency-bound, link bandwidth is irrelevant;
  * the extended-domain phase of a proof (coset NTTs, evaluate_h, division by the vanishing polynomial, inverse
    transform) is split by COSET of the n-th roots of unity: rank r evaluates the quotient on the cosets
    j = r mod shards from replicated coefficient vectors -- only the c = degree - 1 cosets that determine the quotient's
    c pieces, not all 2^(extended_k - k) of the extended domain -- no exchange until the per-coset polynomials
    P_j = sum_m gamma_j^m h_m are broadcast (one n-vector each) and un-mixed into the pieces h_m by the inverse
    Vandermonde matrix (`coset_unmix_matrix`);
  * whole columns can also be dealt round-robin (`shard_columns`) when a caller has independent polynomials
    (bench.py's NTT leg: one polynomial per GPU, no collective).
"""
import ctypes

import numpy as np

from ._lib import check, lib

R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


def shard_columns(num_columns, world, rank):
    """Round-robin ownership of whole columns/polynomials."""
    return list(range(rank, num_columns, world))


def msm_split_range(n, world, rank):
    """Contiguous chunk of rank `rank`: part_len = ceil(n / world) (arithmetic.rs:426)."""
    if n == 0:
        return 0, 0
    part_len = (n + world - 1) // world
    lo = min(rank * part_len, n)
    return lo, min(lo + part_len, n)


def g1_sum(points):
    """Host-side fold of Jacobian points, shape (count, 12) uint64 (arithmetic.rs:433-435)."""
    pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 12)
    out = np.zeros(12, dtype=np.uint64)
    check(lib().h2_g1_sum(pts.ctypes.data_as(ctypes.c_void_p), len(pts), out.ctypes.data_as(ctypes.c_void_p)), "h2_g1_sum")
    return out


def _empty_like(t):
    """an uninitialised vector of which only a part will be written (poisoned under H2_POISON_EMPTY=1, prover.POISON_EMPTY)"""
    import os

    import torch

    out = torch.empty_like(t)
    if os.environ.get("H2_POISON_EMPTY") == "1":
        out.fill_(-1)
    return out


def _backend(group):
    import torch.distributed as dist

    return dist.get_backend(group)


class _on_stream:
    """Run the enclosed torch work -- copies, RCCL collectives (they make the CURRENT stream wait for the communicator's
    stream) -- on the caller's compute stream, so that the kernels queued there before and after are ordered with it.
    `stream` = None keeps torch's current stream and brackets the block with device synchronisations instead."""

    def __init__(self, stream):
        self.stream, self.ctx = stream, None

    def __enter__(self):
        import torch

        if self.stream is None:
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            return None
        self.ctx = torch.cuda.stream(self.stream)
        self.ctx.__enter__()
        return ctypes.c_void_p(self.stream.cuda_stream)

    def __exit__(self, *exc):
        import torch

        if self.ctx is not None:
            return self.ctx.__exit__(*exc)
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        return False


# ---- what the collectives of one proof cost, per phase (bench.py --gpus N: "communication next to compute") ------------
# While a TRACED proof runs (prover.create_proof_ext with `timings`: an extra, untimed proof that already synchronises at
# every phase boundary) each collective below is bracketed by stream synchronisations and its wall time, payload and name
# are recorded; the asynchronous broadcasts are then waited for at once, i.e. measured SERIALISED -- an upper bound of what
# the timed proofs, which overlap them with compute, expose.  Off (None) otherwise: no synchronisation is added.
COMM_TRACE = None


def comm_trace_begin():
    global COMM_TRACE
    COMM_TRACE = []


def comm_trace_phase(name):
    """the phase that just ended: every entry recorded since the previous boundary belongs to it"""
    if COMM_TRACE is not None:
        for e in COMM_TRACE:
            if e[0] is None:
                e[0] = name


def comm_trace_end():
    """-> {phase: {"seconds", "bytes", "calls", "by_collective": {name: seconds}}}"""
    global COMM_TRACE
    trace, COMM_TRACE = COMM_TRACE or [], None
    out = {}
    for phase, name, sec, nbytes in trace:
        d = out.setdefault(phase or "after the last phase", {"seconds": 0.0, "bytes": 0, "calls": 0, "by_collective": {}})
        d["seconds"] += sec
        d["bytes"] += int(nbytes)
        d["calls"] += 1
        d["by_collective"][name] = d["by_collective"].get(name, 0.0) + sec
    return out


class _traced:
    def __init__(self, name, nbytes, stream=None):
        self.name, self.nbytes, self.stream, self.t0 = name, nbytes, stream, None

    def _sync(self):
        import torch

        if torch.cuda.is_available():
            if self.stream is not None:
                self.stream.synchronize()
            else:
                torch.cuda.synchronize()

    def __enter__(self):
        if COMM_TRACE is not None:
            import time

            self._sync()
            self.t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        if self.t0 is not None and COMM_TRACE is not None:
            import time

            self._sync()
            COMM_TRACE.append([None, self.name, time.perf_counter() - self.t0, self.nbytes])
        return False


def allgather_fold(partial_xyz, group=None, device=None):
    """All-gather every rank's partial point and fold; every rank returns the full sum."""
    return allgather_fold_many(np.asarray(partial_xyz, dtype=np.uint64).reshape(1, 12), group, device)[0]


def allgather_fold_many(partials_xyz, group=None, device=None, stream=None):
    """`partials_xyz`: (count, 12) -- this rank's partial point of each of `count` range-split MSMs.  One
    all-gather of world x count x 96 B, then `count` folds in rank order (so every rank derives the same
    Jacobian representation, hence the same transcript).  Returns (count, 12).

    Under RCCL ("nccl") with a device the gathered points stay on the GPU: one upload of this rank's partials (the MSM's
    window Horner finishes on the host), all-gather over xGMI, h2_dev_g1_fold, one read-back of count x 96 B.  Under
    gloo (CPU tests, or two test processes sharing one GPU) the same exchange runs on host tensors."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    mine_np = np.ascontiguousarray(partials_xyz, dtype=np.uint64).reshape(-1, 12)
    count = mine_np.shape[0]
    mine = torch.from_numpy(mine_np.view(np.int64).copy())
    if device is not None and _backend(group) == "nccl":
        with _traced("all_gather (partial points)", world * count * 96, stream), _on_stream(stream) as handle:
            mine = mine.to(device)
            gathered = torch.empty((world, count, 12), dtype=torch.int64, device=device)
            dist.all_gather_into_tensor(gathered.view(-1), mine.view(-1), group=group)
            out = torch.empty((count, 12), dtype=torch.int64, device=device)
            if handle is None:
                torch.cuda.synchronize()       # the fold runs on the library's stream: the gather must have landed
            check(lib().h2_dev_g1_fold(gathered.data_ptr(), world, count, out.data_ptr(), handle), "h2_dev_g1_fold")
            if handle is None:
                check(lib().h2_synchronize(), "h2_synchronize")
            return out.cpu().numpy().view(np.uint64)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    with _traced("all_gather (partial points)", world * count * 96, stream):
        dist.all_gather(gathered, mine, group=group)
    pts = np.stack([t.numpy().view(np.uint64) for t in gathered])      # (world, count, 12)
    return np.stack([g1_sum(pts[:, j, :]) for j in range(count)])


def allreduce_max(values, group=None, device=None):
    """element-wise maximum of a short list of non-negative integers over the ranks (the per-column scalar bounds)"""
    import torch
    import torch.distributed as dist

    t = torch.tensor(list(values), dtype=torch.int64)
    if device is not None and _backend(group) == "nccl":
        t = t.to(device)
    with _traced("all_reduce max (scalar bounds)", t.numel() * 8):
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        out = [int(v) for v in t.cpu().tolist()]
    return out


def allreduce_counts(counts, group=None, stream=None):
    """`counts`: an int32 device tensor of n + 1 counters (h2_dev_logup_counts: credits per table row, then the number of input
    values this rank could not find); summed over the ranks in place -- integer addition IS an RCCL reduction, unlike field
    addition -- and the total of the last counter returned (the same on every rank, so that all of them raise or none)."""
    import torch.distributed as dist

    with _traced("all_reduce sum (logup counts)", counts.numel() * 4, stream), _on_stream(stream):
        if _backend(group) == "nccl":
            dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
        else:
            host = counts.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            counts.copy_(host.to(counts.device))
        return int(counts[-1].item())


def allgather_rows(t, lo, hi, group=None, stream=None):
    """`t`: an (n, 4) int64 device tensor of which this rank holds the rows [lo, hi) (n / world rows, rank order); on
    return every rank holds every row.  RCCL: one in-place all-gather over xGMI; gloo: through host memory."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    assert t.shape[0] % world == 0 and hi - lo == t.shape[0] // world
    with _traced("all_gather (rows)", t.numel() * 8, stream), _on_stream(stream):
        if _backend(group) == "nccl":
            dist.all_gather_into_tensor(t.view(-1), t[lo:hi].reshape(-1), group=group)
            return t
        mine = t[lo:hi].cpu()
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        t.copy_(torch.cat(parts).to(t.device))
    return t


# ---- whole columns dealt over the ranks: the witness-dependent inverse transforms (north_star's per-column NTT sharding) ---
# An NTT does not split by row range, so the inverse transforms of a proof's advice / product / multiplicity columns are
# dealt round-robin BY COLUMN: rank i mod P transforms column i (the rows of a range-computed column are gathered to that
# rank only), and the coefficient vectors reach the other ranks by broadcasts that run on a second communicator and a
# side stream, under whatever the compute stream does next (the lookup / permutation phases after the advice columns, the
# advice columns' coset transforms after the product columns) instead of P copies of every transform.
def gather_rows_to(t, lo, hi, dst, group=None, stream=None):
    """`t`: an (n, 4) int64 device tensor of which every rank holds the rows [lo, hi) (n / world rows, rank order); on
    return rank `dst` of the group holds every row (the other ranks' tensors are unchanged)."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    m = t.shape[0] // world
    assert t.shape[0] % world == 0 and hi - lo == m
    gdst = dist.get_global_rank(group, dst) if group is not None else dst
    with _traced("gather (rows to the transforming rank)", t.numel() * 8, stream), _on_stream(stream):
        if _backend(group) == "nccl":
            parts = [t[r * m:(r + 1) * m] for r in range(world)] if rank == dst else None
            dist.gather(t[lo:hi].clone(), gather_list=parts, dst=gdst, group=group)
            return t
        mine = t[lo:hi].cpu()
        parts = [torch.empty_like(mine) for _ in range(world)] if rank == dst else None
        dist.gather(mine, gather_list=parts, dst=gdst, group=group)
        if rank == dst:
            t.copy_(torch.cat(parts).to(t.device))
    return t


class _Arrival:
    """columns on their way to this rank: `wait()` makes the compute stream wait for them (idempotent)"""

    def __init__(self, works, stream):
        self.works, self.stream = works, stream

    def wait(self):
        import torch

        if self.works:
            with torch.cuda.stream(self.stream):
                for w in self.works:
                    w.wait()                    # the current (compute) stream waits for the communicator's stream
            self.works = []


def broadcast_columns_begin(cols, owners, group=None, stream=None, side=None):
    """cols[i] is complete on rank owners[i] once the work queued on `stream` has run; start completing it on every rank.
    RCCL: asynchronous broadcasts on `group`'s communicator (a second one next to the default group's, so that the small
    all-gathers of the commitments do not queue behind them), issued from the side stream behind an event on `stream`.
    gloo: through host memory, done on return.  Returns an `_Arrival`."""
    import torch
    import torch.distributed as dist

    rank = dist.get_rank(group)
    nbytes = sum(t.numel() * 8 for t in cols)
    if _backend(group) != "nccl":
        with _traced("broadcast (coefficient vectors)", nbytes, stream), _on_stream(stream):
            for t, owner in zip(cols, owners):
                buf = t.cpu()
                dist.broadcast(buf, src=dist.get_global_rank(group, owner) if group is not None else owner, group=group)
                if rank != owner:
                    t.copy_(buf.to(t.device))
        return _Arrival([], stream)
    ready = torch.cuda.Event()
    ready.record(stream)
    works = []
    with torch.cuda.stream(side if side is not None else stream):
        if side is not None:
            side.wait_event(ready)
        for t, owner in zip(cols, owners):
            if side is not None:
                t.record_stream(side)
            works.append(dist.broadcast(t, src=dist.get_global_rank(group, owner) if group is not None else owner,
                                        group=group, async_op=True))
    arrival = _Arrival(works, stream)
    if COMM_TRACE is not None:
        # a traced proof: the transfer alone, serialised (the timed proofs run it under the next phase's compute)
        import time

        stream.synchronize()
        t0 = time.perf_counter()
        arrival.wait()
        stream.synchronize()
        COMM_TRACE.append([None, "broadcast (coefficient vectors; serialised here, overlapped in the timed proofs)",
                           time.perf_counter() - t0, nbytes])
    return arrival


# ---- coset sharding of the extended-domain phase ---------------------------------------------------------------------
def coset_plan(c, world, rank):
    """c cosets (the quotient_poly_degree = degree - 1 cosets that determine the quotient, prover.Device.coset_plan) over
    `world` ranks: shards = min(c, world) groups; rank r works on the cosets j = r mod shards (ranks beyond c replicate
    a shard).  Returns (shards, owned cosets)."""
    shards = min(c, world)
    return shards, [j for j in range(c) if j % shards == rank % shards]


def coset_unmix_matrix(gammas, rows):
    """The per-coset polynomials are P_j = sum_m gamma_j^m h_m (gamma_j = (zeta extended_omega^j)^n, m < c = len(gammas)):
    returns the first `rows` rows of the inverse of the Vandermonde matrix V[j][m] = gamma_j^m modulo r, so that
    h_m = sum_j M[m][j] P_j.  With c = rows = quotient_poly_degree cosets the system is square.  Host integers; c <= 8."""
    c = len(gammas)
    a = [[pow(g, m, R_MOD) for m in range(c)] + [1 if i == j else 0 for i in range(c)] for j, g in enumerate(gammas)]
    for col in range(c):                       # Gauss-Jordan modulo r
        piv = next(r for r in range(col, c) if a[r][col] % R_MOD)
        a[col], a[piv] = a[piv], a[col]
        inv = pow(a[col][col], -1, R_MOD)
        a[col] = [v * inv % R_MOD for v in a[col]]
        for r in range(c):
            if r != col and a[r][col]:
                f = a[r][col]
                a[r] = [(v - f * w) % R_MOD for v, w in zip(a[r], a[col])]
    return [row[c:] for row in a[:rows]]


def exchange_cosets(mine, c, shards, group=None, stream=None):
    """`mine`: {coset j: (n, 4) int64 device tensor} for the cosets this rank evaluated.  Every coset polynomial is
    broadcast from the first rank of its shard (rank j mod shards); returns the list of all c tensors.  One n-vector per
    coset crosses the links (k = 24: 512 MiB each) -- the only bulk exchange of a proof."""
    import torch
    import torch.distributed as dist

    rank = dist.get_rank(group)
    staged = _backend(group) != "nccl"          # gloo: through host memory
    template = next(iter(mine.values()))
    out = []
    with _traced("broadcast (coset polynomials)", c * template.numel() * 8, stream), _on_stream(stream):
        for j in range(c):
            src = j % shards
            t = mine[j] if (j in mine and rank == src) else _empty_like(template)
            if j in mine and rank != src:
                out.append(mine[j])             # a replica of the shard already holds it; still take part in the broadcast
            buf = t.cpu() if staged else t
            dist.broadcast(buf, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
            if not (j in mine and rank != src):
                out.append(buf.to(template.device) if staged else buf)
    return out


# ---- several ranks per coset: columns dealt for the transforms, rows dealt for the evaluator --------------------------------
# With P ranks and c < P cosets (P a multiple of c) the G = P / c ranks that share coset j used to do the same work G times
# over.  Instead member g of the coset's rank group takes EVERY G-th column to the coset (n-point coset transforms), the
# members exchange row slices -- member g receives the rows [g n / G - halo, (g + 1) n / G + halo) of every column, the halo
# being the rotations the evaluator reads -- and each evaluates the quotient on its n / G rows (h2_evalh_desc::row_begin /
# row_count); the rows of the quotient are all-gathered inside the group for the inverse transform.
def slice_rows(n, G, g, halo_lo, halo_hi, device):
    """indices (modulo n) of the rows member g of a G-member coset group evaluates, with the halo the rotations reach"""
    import torch

    m = n // G
    return torch.arange(g * m - halo_lo, (g + 1) * m + halo_hi, device=device, dtype=torch.int64) % n


def exchange_row_slices(columns, owners, n, G, g, halo_lo, halo_hi, group=None, stream=None):
    """`columns[i]`: an (n, 4) device tensor, complete on member owners[i] of the group (any content elsewhere).  On return
    every member holds, in every column, the rows `slice_rows(n, G, g, ...)` of ITS slice (its own columns stay complete).
    RCCL: one all-to-all over xGMI; gloo: one scatter per member through host memory."""
    import torch
    import torch.distributed as dist

    device = columns[0].device
    per = n // G + halo_lo + halo_hi
    mine = [i for i, o in enumerate(owners) if o == g]
    theirs = {m: [i for i, o in enumerate(owners) if o == m] for m in range(G)}
    with _traced("all_to_all (row slices inside a coset group)", (G - 1) * per * len(mine) * 32, stream), _on_stream(stream):
        # (the index vectors are built by kernels too: on THIS stream -- torch's default stream and a stream created with
        # torch.cuda.Stream() do not order each other, and an index_select that overtakes its own indices reads garbage rows)
        rows = {m: slice_rows(n, G, m, halo_lo, halo_hi, device) for m in range(G)}
        send = [torch.cat([columns[i].index_select(0, rows[m]) for i in mine]) if mine and m != g else
                torch.empty((0, 4), dtype=columns[0].dtype, device=device) for m in range(G)]
        recv = [torch.empty((per * len(theirs[m]) if m != g else 0, 4), dtype=columns[0].dtype, device=device) for m in range(G)]
        if _backend(group) == "nccl":
            dist.all_to_all(recv, send, group=group)
        else:
            for src in range(G):
                if not theirs[src]:                  # a member without columns (fewer columns than members) sends nothing
                    continue
                gsrc = dist.get_global_rank(group, src) if group is not None else src
                if src == g:
                    parts = [t.cpu() for t in send]
                    parts[g] = torch.zeros((per * len(mine), 4), dtype=columns[0].dtype)   # (gloo scatters equal sizes)
                    dist.scatter(torch.empty_like(parts[g]), scatter_list=parts, src=gsrc, group=group)
                else:
                    buf = torch.empty((recv[src].shape[0], 4), dtype=columns[0].dtype)
                    dist.scatter(buf, scatter_list=None, src=gsrc, group=group)
                    recv[src] = buf.to(device)
        for m in range(G):
            if m == g:
                continue
            for k_, i in enumerate(theirs[m]):
                columns[i].index_copy_(0, rows[g], recv[m][k_ * per:(k_ + 1) * per])
    return columns


# ---- index-range sharding of the O(n) passes --------------------------------------------------
# Every rank holds the same full-size vectors and works on the rows / coefficients [lo, hi) of its contiguous range (the
# range its share of every range-split MSM consumes).  Elementwise passes and linear combinations need nothing else; the
# three operations with a dependency ACROSS the range -- prefix scans, Kate division, Horner evaluation -- exchange one field
# element per rank (an all-gather of world x 32 B) and finish locally.  The helpers below are that arithmetic on host
# integers; the device work is prover.py's.
MASK64 = (1 << 64) - 1


def allgather_scalars(values, group=None, device=None):
    """every rank contributes len(values) integers below 2^256; returns [rank][i] on every rank.  One small all-gather
    (RCCL: from device memory; gloo: host tensors)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    a = np.array([[(int(v) >> (64 * k)) & MASK64 for k in range(4)] for v in values], dtype=np.uint64).reshape(-1, 4)
    mine = torch.from_numpy(a.view(np.int64).copy())
    if device is not None and _backend(group) == "nccl":
        mine = mine.to(device)
    out = [torch.empty_like(mine) for _ in range(world)]
    with _traced("all_gather (field elements)", world * mine.numel() * 8):
        dist.all_gather(out, mine, group=group)
    res = []
    for t in out:
        rows = t.cpu().numpy().view(np.uint64).reshape(-1, 4)
        res.append([int(r[0]) | (int(r[1]) << 64) | (int(r[2]) << 128) | (int(r[3]) << 192) for r in rows])
    return res


def scan_carries(totals, init, product=True):
    """prefix scan over equal contiguous ranges: `totals[r]` = the product (sum) of rank r's own factors (terms);
    returns the value entering each range, carry[r] = init * prod_{s < r} totals[s]  (init + sum ...)."""
    out, acc = [], init % R_MOD
    for t in totals:
        out.append(acc)
        acc = acc * t % R_MOD if product else (acc + t) % R_MOD
    return out


def kate_carries(partials, b, m):
    """Kate division q(X) = a(X) / (X - b) over ranges of m coefficients: q[j] = sum_{t > j} a[t] b^(t - j - 1).
    `partials[s]` = sum_{t in range s} a[t] b^(t - lo_s) (the local Horner value of range s at b).  Returns, per rank r,
    C_r = sum_{t >= hi_r} a[t] b^(t - hi_r) = sum_{s > r} partials[s] b^((s - r - 1) m): the recurrence value entering range
    r from above (it is also q[hi_r - 1])."""
    world = len(partials)
    bm = pow(b, m, R_MOD)
    out, acc = [0] * world, 0
    for r in range(world - 2, -1, -1):
        acc = (partials[r + 1] + bm * acc) % R_MOD
        out[r] = acc
    return out


def combine_range_evals(parts, x, m):
    """p(x) = sum_r x^(r m) parts[r] with parts[r] = sum_{t in range r} p[t] x^(t - lo_r) (Horner over the ranges)"""
    xm, acc = pow(x, m, R_MOD), 0
    for v in reversed(parts):
        acc = (acc * xm + v) % R_MOD
    return acc


def scatter_cosets(mine, c, shards, lo, hi, group=None, stream=None):
    """`mine`: {coset j: (n, 4) device tensor} for the cosets this rank evaluated; every rank needs only the coefficients
    [lo_r, hi_r) of every coset polynomial (the un-mixing into the quotient's pieces is a linear combination, and everything
    after it works on ranges): the owner of coset j (rank j mod shards) scatters the c x n / world slices instead of
    broadcasting n coefficients to everybody.  Returns a list of c full-size tensors of which only [lo, hi) is valid."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    staged = _backend(group) != "nccl"
    template = next(iter(mine.values())) if mine else None
    assert template is not None, "every rank owns at least one coset (ranks beyond the cosets replicate a shard)"
    n = template.shape[0]
    assert n % world == 0 and hi - lo == n // world
    out = []
    with _traced("scatter (coset polynomial slices)", c * (hi - lo) * 32 * (world - 1), stream), _on_stream(stream):
        for j in range(c):
            src = j % shards
            full = mine[j] if j in mine else _empty_like(template)
            recv = full[lo:hi]
            if rank == src:
                parts = [full[r * (n // world):(r + 1) * (n // world)] for r in range(world)]
                if staged:
                    parts = [p.cpu() for p in parts]
                buf = parts[rank].clone() if staged else torch.empty_like(recv)
                dist.scatter(buf, scatter_list=parts, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
            else:
                buf = torch.empty((hi - lo, 4), dtype=template.dtype) if staged else recv
                dist.scatter(buf, scatter_list=None, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
                if staged:
                    recv.copy_(buf.to(template.device))
            out.append(full)
    return out
