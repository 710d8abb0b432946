"""N > 1 path on CPU: world_size-2 gloo processes run the split-MSM exchange (all-gather of per-rank
partial points + local fold, parallel.allgather_fold) and the per-column sharding plan.  The per-rank
partial points come from the CPU oracle here (there is no GPU); on the GPU box the same code path is
fed by h2_msm and runs over RCCL."""
import os
import socket
import subprocess
import sys

from h2util import ROOT

WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, torch.distributed as dist
from halo2_gpu_specific_amd import parallel
from h2util import Oracle, arr_to_points
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
oracle = Oracle.get()
n = 1000
s, p = oracle.random_fr(5, n), oracle.random_g1(6, n)
lo, hi = parallel.msm_split_range(n, world, rank)
partial = oracle.best_multiexp(s[lo:hi], p[lo:hi], threads=1)
full = parallel.allgather_fold(partial)
want = arr_to_points(oracle.to_affine(oracle.best_multiexp(s, p, threads=2)))[0]
assert arr_to_points(oracle.to_affine(full))[0] == want, "rank %%d: folded MSM differs" %% rank
# the prover's shape: several range-split MSMs folded by one all-gather (parallel.allgather_fold_many)
cols3 = [oracle.random_fr(20 + j, n) for j in range(3)]
partials = np.stack([oracle.best_multiexp(c[lo:hi], p[lo:hi], threads=1) for c in cols3])
folded = parallel.allgather_fold_many(partials)
for j, c in enumerate(cols3):
    want_j = arr_to_points(oracle.to_affine(oracle.best_multiexp(c, p, threads=2)))[0]
    assert arr_to_points(oracle.to_affine(folded[j]))[0] == want_j, "rank %%d: column %%d differs" %% (rank, j)
# every rank must end with the same representation (the transcript hashes the normalised point, but the fold
# order is fixed anyway)
chk = torch.from_numpy(folded.view(np.int64).copy()); ref = chk.clone(); dist.broadcast(ref, 0)
assert torch.equal(chk, ref)
cols = parallel.shard_columns(7, world, rank)
t = torch.zeros(7, dtype=torch.int64); t[cols] = 1
dist.all_reduce(t)
assert t.tolist() == [1] * 7
# bench-style timing reduction: max over ranks
m = torch.tensor([float(rank + 1)], dtype=torch.float64); dist.all_reduce(m, op=dist.ReduceOp.MAX)
assert m.item() == world
# ---- coset sharding of the extended domain (parallel.coset_plan / exchange_cosets / coset_unmix_matrix): the quotient
# h = h_0 + X^n h_1 (two pieces, degree-3 circuit: c = 2 cosets) is known here; every rank "evaluates" only its coset:
# P_j = coset-iNTT of h on coset j = h_0 + gamma_j h_1.  After the broadcast and the un-mixing every rank holds h_0, h_1.
from h2util import R_MOD, from_mont, to_mont, fr_mont
k, j_deg = 6, 3
d, _ = oracle.domain(j_deg, k)
nn, c = 1 << k, 1 << (d.extended_k - k)
shards, owned = parallel.coset_plan(c, world, rank)
assert shards == 2 and owned == [rank]
zeta, wext = from_mont(d.fr("g_coset").reshape(1, 4))[0], from_mont(d.fr("extended_omega").reshape(1, 4))[0]
pieces = [from_mont(oracle.random_fr(70 + m, nn)) for m in range(c)]
hcoef = to_mont([v for m in range(c) for v in pieces[m]])                   # h as one coefficient vector of length c n
ext = from_mont(oracle.best_fft(to_mont([v * pow(zeta, t, R_MOD) %% R_MOD for t, v in enumerate(from_mont(hcoef))]),
                                fr_mont(wext), d.extended_k))              # h on the whole extended coset
mine = {}
for j in owned:
    g_j = zeta * pow(wext, j, R_MOD) %% R_MOD
    vals = to_mont(ext[j::c])                                               # extended indices c i + j
    w_inv = fr_mont(pow(pow(wext, c, R_MOD), -1, R_MOD))
    coeffs = from_mont(oracle.ifft(vals, w_inv, k, fr_mont(pow(nn, -1, R_MOD))))
    g_inv = pow(g_j, -1, R_MOD)
    pj = [v * pow(g_inv, t, R_MOD) %% R_MOD for t, v in enumerate(coeffs)]
    gamma = pow(g_j, nn, R_MOD)
    assert pj == [(pieces[0][t] + gamma * pieces[1][t]) %% R_MOD for t in range(nn)], "P_j != sum gamma_j^m h_m"
    mine[j] = torch.from_numpy(to_mont(pj).view(np.int64).copy())
polys = parallel.exchange_cosets(mine, c, shards)
gammas = [pow(zeta * pow(wext, j, R_MOD) %% R_MOD, nn, R_MOD) for j in range(c)]
unmix = parallel.coset_unmix_matrix(gammas, c)
for m in range(c):
    got = [sum(unmix[m][j] * v for j, v in enumerate(col)) %% R_MOD
           for col in zip(*[from_mont(t.numpy().view(np.uint64)) for t in polys])]
    assert got == pieces[m], "rank %%d: piece %%d differs after the exchange" %% (rank, m)
dist.barrier(); dist.destroy_process_group()
print("OK", rank)
"""


def test_split_msm_allgather_world2(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert "OK %d" % rank in out


RANGE_WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, torch.distributed as dist
from halo2_gpu_specific_amd import parallel
from h2util import Oracle, R_MOD, from_mont, to_mont, fr_mont
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
oracle = Oracle.get()
n = 1 << 10
lo, hi = parallel.msm_split_range(n, world, rank)
m = hi - lo
# ---- Kate division by ranges: the oracle's kate_division (arithmetic.rs:754-773) of the whole vector against the local
# division of every range + the carry from one all-gather of a field element per rank (Device.kate_division_ranges)
a = oracle.random_fr(31, n)
b = from_mont(oracle.random_fr(32, 1))[0]
want = np.zeros((n - 1, 4), dtype=np.uint64)
b_m = fr_mont(b)
oracle.lib.oracle_kate_division(a.ctypes.data, n, b_m.ctypes.data, want.ctypes.data)
want = from_mont(want) + [0]
local = np.zeros((m - 1, 4), dtype=np.uint64)
mine_a = np.ascontiguousarray(a[lo:hi])
oracle.lib.oracle_kate_division(mine_a.ctypes.data, m, b_m.ctypes.data, local.ctypes.data)
local = from_mont(local) + [0]
part = np.zeros(4, dtype=np.uint64)
oracle.lib.oracle_eval_polynomial(mine_a.ctypes.data, m, b_m.ctypes.data, part.ctypes.data)
gathered = parallel.allgather_scalars([from_mont(part.reshape(1, 4))[0]])
carry = parallel.kate_carries([g[0] for g in gathered], b, m)[rank]
got = [(local[t] + carry * pow(b, m - 1 - t, R_MOD)) %% R_MOD for t in range(m)]
assert got == want[lo:hi], "rank %%d: Kate division by ranges differs" %% rank
# ---- evaluation by ranges (Device.eval_polynomial_ranges): p(x) = sum_r x^(lo_r) p_r(x)
x = from_mont(oracle.random_fr(33, 1))[0]
full = np.zeros(4, dtype=np.uint64)
x_m = fr_mont(x)
oracle.lib.oracle_eval_polynomial(a.ctypes.data, n, x_m.ctypes.data, full.ctypes.data)
oracle.lib.oracle_eval_polynomial(mine_a.ctypes.data, m, x_m.ctypes.data, part.ctypes.data)
parts = parallel.allgather_scalars([from_mont(part.reshape(1, 4))[0]])
assert parallel.combine_range_evals([g[0] for g in parts], x, m) == from_mont(full.reshape(1, 4))[0]
# ---- grand product / grand sum by ranges (Device.prefix_scan): z[0] = init, z[i] = z[i-1] * f[i-1]
f = from_mont(a)
init = 12345
for product in (True, False):
    z = [init]
    for i in range(n - 1):
        z.append(z[-1] * f[i] %% R_MOD if product else (z[-1] + f[i]) %% R_MOD)
    tot = 1 if product else 0
    loc = []
    for t in range(m):
        loc.append(tot)
        tot = tot * f[lo + t] %% R_MOD if product else (tot + f[lo + t]) %% R_MOD
    carries = parallel.scan_carries([g[0] for g in parallel.allgather_scalars([tot])], init, product)
    mine = [(carries[rank] * v if product else carries[rank] + v) %% R_MOD for v in loc]
    assert mine == z[lo:hi], "rank %%d: scan by ranges differs" %% rank
# ---- the per-coset quotients scattered as coefficient ranges (parallel.scatter_cosets): owner j %% shards sends rank r its range
c = 3
shards, owned = parallel.coset_plan(c, world, rank)
polys = [oracle.random_fr(50 + j, n) for j in range(c)]
mine = {j: torch.from_numpy(polys[j].view(np.int64).copy()) for j in owned}
out = parallel.scatter_cosets(mine, c, shards, lo, hi)
for j in range(c):
    assert np.array_equal(out[j][lo:hi].numpy().view(np.uint64), polys[j][lo:hi]), (rank, j)
dist.barrier(); dist.destroy_process_group()
print("OK", rank)
"""


def test_range_sharded_scans_division_evaluation_world2(tmp_path):
    """the O(n) passes of one proof on every rank's row / coefficient range -- the carries of
    prefix scans and Kate divisions and the partial Horner values cross in one all-gather of a field element per rank, the
    per-coset quotients are scattered as ranges; here the local pieces come from the CPU oracle, on the GPU box from the
    kernels (tests/test_gpu_plonk.py::test_gloo_ranks_on_one_gpu_prove_the_single_device_bytes)"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(RANGE_WORKER % {"root": ROOT})
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert "OK %d" % rank in out


COLUMN_WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, torch.distributed as dist
from halo2_gpu_specific_amd import parallel
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
gen = torch.Generator(); gen.manual_seed(77)                      # every rank draws the same reference data
n, ncols = 96 * world, 5
ref = [torch.randint(-2**62, 2**62, (n, 4), dtype=torch.int64, generator=gen) for _ in range(ncols)]
lo, hi = parallel.msm_split_range(n, world, rank)
# ---- rows computed by ranges go to the column's owner only (parallel.gather_rows_to)
for i, full in enumerate(ref):
    owner = i %% world
    t = torch.zeros_like(full); t[lo:hi] = full[lo:hi]
    parallel.gather_rows_to(t, lo, hi, owner)
    if rank == owner:
        assert torch.equal(t, full), ("gather_rows_to", i)
    else:
        assert torch.equal(t[lo:hi], full[lo:hi]) and not t[:lo].any() and not t[hi:].any()
# ---- whole columns from their owners to everybody (parallel.broadcast_columns_begin; gloo: complete on return)
cols = [full.clone() if i %% world == rank else torch.zeros_like(full) for i, full in enumerate(ref)]
arrival = parallel.broadcast_columns_begin(cols, [i %% world for i in range(ncols)])
arrival.wait()
assert all(torch.equal(a, b) for a, b in zip(cols, ref)), "broadcast_columns_begin"
# ---- the ranks of one coset: columns dealt for the transforms, row slices (with the rotations' halo, wrapping around the
# domain) exchanged so that every member holds ITS rows of EVERY column (parallel.exchange_row_slices)
G, g = world, rank
for ncol, halo in ((5, (6, 1)), (2, (0, 0)), (7, (3, 2))):        # 2 columns over 3 members: one member owns none
    owners = [i %% G for i in range(ncol)]
    data = [torch.randint(-2**62, 2**62, (n, 4), dtype=torch.int64, generator=gen) for _ in range(ncol)]
    mine = [d.clone() if o == g else torch.full_like(d, -7) for d, o in zip(data, owners)]
    parallel.exchange_row_slices(mine, owners, n, G, g, halo[0], halo[1])
    rows = parallel.slice_rows(n, G, g, halo[0], halo[1], torch.device("cpu"))
    assert rows.shape[0] == n // G + halo[0] + halo[1] and int(rows[0]) == (g * (n // G) - halo[0]) %% n
    for i, (got, want, o) in enumerate(zip(mine, data, owners)):
        assert torch.equal(got[rows], want[rows]), ("exchange_row_slices", ncol, i)
        if o == g:
            assert torch.equal(got, want)
        else:
            other = torch.ones(n, dtype=torch.bool); other[rows] = False
            assert (got[other] == -7).all(), "rows outside the slice were written"
dist.barrier(); dist.destroy_process_group()
print("OK", rank)
"""


def _run_workers(tmp_path, text, world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(text % {"root": ROOT})
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert "OK %d" % rank in out


def test_columns_dealt_over_the_ranks_world2_and_world3(tmp_path):
    """round 4's exchanges: the rows of a range-computed column gathered to its owner, whole coefficient columns broadcast from
    their owners, and -- inside the rank group of one coset -- row slices with the evaluator's halo, including the wrap around
    the domain and a member that owns no column; two and three gloo ranks"""
    _run_workers(tmp_path, COLUMN_WORKER, 2)
    _run_workers(tmp_path, COLUMN_WORKER, 3)
