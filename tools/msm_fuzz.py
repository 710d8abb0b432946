"""Randomised MSM parity: random sizes (1 .. 2^19, ragged), scalar bounds (1 .. 254 bits) and value distributions
(uniform under the bound, a few distinct values, one dominant value, sparse, booleans, P / -P pairs in the bases) through
the C ABI against the CPU oracle, for a time budget.   usage: python tools/msm_fuzz.py [seconds] [seed] [tables|fused]
With `tables` every case also runs over a shifted-base table of its bases (h2_dev_bases_precompute with a random digit
count, tables allowed from one row on), through the device entry point.  With `fused` every case is a GROUP of 2..7 columns
under one bound through h2_dev_msm_batch with the scratch that lets the library commit them as one fused MSM (and, every other
case, with the pipeline's scratch): every column's point against the oracle."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before HIP initialises (halo2-gpu-specific_amd/__init__.py says why)
import torch  # noqa: E402

torch.cuda.init()
import numpy as np  # noqa: E402

from h2util import Oracle, to_mont  # noqa: E402
from halo2_gpu_specific_amd import arithmetic as ar  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tables = len(sys.argv) > 3 and sys.argv[3] == "tables"
fused = len(sys.argv) > 3 and sys.argv[3] == "fused"
if tables:
    os.environ["H2_MSM_TABLE_MIN_N"] = "1"
rng = np.random.default_rng(seed)
oracle = Oracle.get()
R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


def aff(j):
    return tuple(oracle.to_affine(j).tolist())


def rand_ints(count, bits):
    """`count` uniform integers below min(2^bits, r) as Python ints"""
    words = rng.integers(0, 1 << 62, size=(count, 5), dtype=np.uint64)
    out = []
    for row in words:
        v = 0
        for w in row:
            v = (v << 62) | int(w)
        out.append((v & ((1 << bits) - 1)) % R_MOD)
    return out


def over_table(scalars, pts, bits, digits):
    import halo2_gpu_specific_amd as h2

    L = h2.lib()
    d_p = torch.from_numpy(np.ascontiguousarray(pts).view(np.int64)).cuda()
    d_s = torch.from_numpy(np.ascontiguousarray(scalars).view(np.int64)).cuda()
    assert L.h2_dev_bases_precompute(d_p.data_ptr(), len(pts), digits, None) == 0, L.h2_last_error()
    nbytes = L.h2_msm_scratch_bytes(len(pts), bits)
    scratch = torch.empty(max(nbytes, 256), dtype=torch.uint8, device="cuda")
    out = np.zeros(12, dtype=np.uint64)
    rc = L.h2_dev_msm(d_s.data_ptr(), d_p.data_ptr(), len(pts), bits, scratch.data_ptr(), nbytes, out.ctypes.data, None)
    assert rc == 0, L.h2_last_error()
    assert L.h2_dev_bases_forget(d_p.data_ptr()) == 0
    return out


def column(n, bits, kind):
    pool = rand_ints(min(n, 4096), bits)
    idx = rng.integers(0, len(pool), size=n)
    if kind == 0:
        return [pool[i] for i in idx]
    if kind == 1:
        d = int(rng.choice([1, 2, 3, 16, 200]))
        return [pool[i % d] for i in idx]
    if kind == 2:
        keep = rng.random(n) < float(rng.choice([0.05, 0.3, 0.6]))
        return [pool[i] if k else pool[0] for i, k in zip(idx, keep)]
    if kind == 3:
        keep = rng.random(n) < 0.08
        return [pool[i] if k else 0 for i, k in zip(idx, keep)]
    if kind == 4:
        return [int(v) % (1 << bits) for v in rng.integers(0, 3, size=n)]
    edge = [0, 1, (1 << bits) - 1 if bits < 254 else R_MOD - 1, 1]
    return [edge[i % 4] % R_MOD for i in idx]


def fused_cases():
    """groups of columns under one bound through the batch entry point, fused and pipelined"""
    import ctypes

    import halo2_gpu_specific_amd as h2

    L = h2.lib()
    cases = 0
    t_end = time.time() + budget
    all_pts = oracle.random_g1(seed, 1 << 17)
    while time.time() < t_end:
        log_n = int(rng.integers(4, 18))
        n = int(rng.integers(max(8, (1 << log_n) // 2), (1 << log_n) + 1))
        bits = int(rng.choice([1, 2, 5, 8, 12, 13, 16, 17, 20, 32, 33, 48, 64, 100, 128, 200, 254]))
        count = int(rng.integers(2, 8))
        cols = [column(n, bits, int(rng.integers(0, 6))) for _ in range(count)]
        start = int(rng.integers(0, (1 << 17) - n + 1))
        pts = all_pts[start:start + n].copy()
        if rng.random() < 0.2:
            pts[n // 2:] = pts[: n - n // 2]
        sc = [to_mont(c) for c in cols]
        want = [aff(oracle.best_multiexp(c, pts)) for c in sc]
        d_p = torch.from_numpy(np.ascontiguousarray(pts).view(np.int64)).cuda()
        d_s = [torch.from_numpy(np.ascontiguousarray(c).view(np.int64)).cuda() for c in sc]
        ptrs = (ctypes.c_void_p * count)(*[t.data_ptr() for t in d_s])
        fused_bytes = L.h2_msm_batch_scratch_bytes(n, bits, count)
        pipe_bytes = 2 * ((L.h2_msm_scratch_bytes(n, bits) + 255) // 256 * 256)
        for nbytes in ((fused_bytes, pipe_bytes) if cases % 2 == 0 else (fused_bytes,)):
            scratch = torch.empty(max(nbytes, 256), dtype=torch.uint8, device="cuda")
            out = np.zeros((count, 12), dtype=np.uint64)
            rc = L.h2_dev_msm_batch(ptrs, count, d_p.data_ptr(), n, bits, scratch.data_ptr(), nbytes, out.ctypes.data, None)
            assert rc == 0, L.h2_last_error()
            got = [aff(out[j]) for j in range(count)]
            if got != want:
                bad = [j for j in range(count) if got[j] != want[j]]
                print("MISMATCH fused group: n=%d bits=%d columns=%d (wrong: %s) scratch=%d seed=%d case=%d" % (
                    n, bits, count, bad, nbytes, seed, cases))
                sys.exit(1)
        cases += 1
    print("msm_fuzz: %d groups of 2..7 columns in %.0f s, every column equal to the oracle (seed %d), fused and pipelined" % (
        cases, budget, seed))


if fused:
    fused_cases()
    sys.exit(0)

cases = 0
t_end = time.time() + budget
all_pts = oracle.random_g1(seed, 1 << 19)
while time.time() < t_end:
    log_n = int(rng.integers(0, 20))
    n = int(rng.integers(max(1, (1 << log_n) // 2), (1 << log_n) + 1))
    bits = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 12, 13, 16, 17, 20, 31, 32, 33, 48, 64, 65, 100, 127, 128, 129, 200, 253, 254]))
    kind = int(rng.integers(0, 6))
    pool = rand_ints(min(n, 4096), bits)
    idx = rng.integers(0, len(pool), size=n)
    if kind == 0:      # uniform (from a pool of <= 4096 values when n is larger: cheap to build, still spread)
        vals = [pool[i] for i in idx]
    elif kind == 1:    # a few distinct values
        d = int(rng.choice([1, 2, 3, 16, 200]))
        vals = [pool[i % d] for i in idx]
    elif kind == 2:    # one dominant value + uniform remainder
        keep = rng.random(n) < float(rng.choice([0.05, 0.3, 0.6]))
        vals = [pool[i] if k else pool[0] for i, k in zip(idx, keep)]
    elif kind == 3:    # sparse: mostly zero
        keep = rng.random(n) < 0.08
        vals = [pool[i] if k else 0 for i, k in zip(idx, keep)]
    elif kind == 4:    # booleans / tiny values whatever the bound says
        vals = [int(v) for v in rng.integers(0, 3, size=n)]
    else:              # edge values
        edge = [0, 1, (1 << bits) - 1 if bits < 254 else R_MOD - 1, R_MOD - 1 if bits == 254 else 1]
        vals = [edge[i % 4] % R_MOD for i in idx]
    top = max(vals).bit_length()
    use_bits = max(bits, top) if rng.random() < 0.8 else 254
    scalars = to_mont(vals)
    start = int(rng.integers(0, (1 << 19) - n + 1))
    pts = all_pts[start:start + n].copy()
    if rng.random() < 0.2 and n >= 4:  # repeated points: P + P and bucket collisions
        pts[n // 2:] = pts[: n - n // 2]
    want = aff(oracle.best_multiexp(scalars, pts))
    got = aff(ar.gpu_multiexp_single_gpu_with_bound(scalars, pts, use_bits))
    cases += 1
    if got != want:
        print("MISMATCH n=%d bits=%d use_bits=%d kind=%d seed=%d case=%d" % (n, bits, use_bits, kind, seed, cases))
        sys.exit(1)
    if tables:
        digits = int(rng.choice([0, 11, 12, 13, 15, 16, 20, 27, 32]))
        if aff(over_table(scalars, pts, use_bits, digits)) != want:
            print("MISMATCH over a table: n=%d bits=%d use_bits=%d kind=%d digits=%d seed=%d case=%d" % (n, bits, use_bits, kind, digits, seed, cases))
            sys.exit(1)
print("msm_fuzz: %d cases in %.0f s, all equal to the oracle (seed %d)%s" % (cases, budget, seed, ", windowed and over tables" if tables else ""))
