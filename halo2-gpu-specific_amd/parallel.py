"""Multi-GPU sharding of the hot path: one process per GPU (torch.distributed; backend "nccl" = RCCL
over xGMI on the GPU box, "gloo" in the CPU tests).

The path shards over independent units:
  * per-column NTTs / MSMs / MSM+iNTTs are dealt round-robin to ranks -- no data-path collective
    (reference: one rayon task per column, plonk/prover.rs:293-299,477-487,643-646);
  * one MSM may also be split into contiguous ceil(n / P) chunks (gpu_multiexp_bound,
    arithmetic.rs:425-435); each rank then holds one partial G1 point (96 B).  EC addition is not an
    RCCL reduction operator, so the exchange is an all-gather of P x 96 B followed by a local fold
    (h2_g1_sum) -- latency-bound, link bandwidth is irrelevant at this size.
"""
import ctypes

import numpy as np

from ._lib import check, lib


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


def allgather_fold(partial_xyz, group=None, device=None):
    """All-gather every rank's partial point and fold locally; every rank returns the full sum."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    mine = torch.from_numpy(np.ascontiguousarray(partial_xyz, dtype=np.uint64).view(np.int64).copy())
    if device is not None:
        mine = mine.to(device)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine, group=group)
    pts = np.stack([t.cpu().numpy().view(np.uint64) for t in gathered])
    return g1_sum(pts)


def allgather_fold_many(partials_xyz, group=None, device=None):
    """`partials_xyz`: (count, 12) -- this rank's partial point of each of `count` range-split MSMs.  One
    all-gather of world x count x 96 B, then `count` local folds in rank order (so every rank derives the same
    Jacobian representation, hence the same transcript).  Returns (count, 12)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    mine_np = np.ascontiguousarray(partials_xyz, dtype=np.uint64).reshape(-1, 12)
    mine = torch.from_numpy(mine_np.view(np.int64).copy())
    if device is not None:
        mine = mine.to(device)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine, group=group)
    pts = np.stack([t.cpu().numpy().view(np.uint64) for t in gathered])      # (world, count, 12)
    return np.stack([g1_sum(pts[:, j, :]) for j in range(mine_np.shape[0])])
