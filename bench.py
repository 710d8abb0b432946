#!/usr/bin/env python3
"""bench.py -- halo2 prover hot path on MI355X.

One step = one forward + one inverse 2^24-point BN254 Fr NTT (BASELINE.json configs[2], the
k = 24 size the metric is quoted on), data resident in HBM.  `value` = NTT Fr-ops/s with
Fr-ops = 3 * (n/2) * log2(n) per transform.  The MSM leg (2^20 random
scalars/points, configs[1]) is timed separately and reported under "msm".

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU (torchrun), every rank runs the same per-column workload on its own
device (independent polynomials: no data-path collective) -> "scaling": "weak"; the timing is
the max over ranks.  torch is used for device memory, streams and the rank barrier only.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
ROOT_OF_UNITY = 0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec
MUL_CEILING = 1.31e11  # 254-bit Montgomery multiplications/s, whole chip (tools/mulbench.hip, profiles/r1_mulbench.txt)


def fr_limbs(v):
    v = v * (1 << 256) % R_MOD  # Montgomery form
    return np.array([(v >> (64 * i)) & ((1 << 64) - 1) for i in range(4)], dtype=np.uint64)


def vp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


Q_MOD = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47


def jac_eq(p, q):
    """projective equality of two Jacobian points (12 x u64, Montgomery limbs): the representation
    depends on the (atomic-ordered) summation order, the group element does not."""
    def limbs(a, i):
        return sum(int(a[4 * i + k]) << (64 * k) for k in range(4))
    x1, y1, z1, x2, y2, z2 = limbs(p, 0), limbs(p, 1), limbs(p, 2), limbs(q, 0), limbs(q, 1), limbs(q, 2)
    if z1 == 0 or z2 == 0:
        return z1 == z2
    return (x1 * z2 * z2 - x2 * z1 * z1) % Q_MOD == 0 and (y1 * z2 ** 3 - y2 * z1 ** 3) % Q_MOD == 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-n", type=int, default=24)
    ap.add_argument("--msm-log-n", type=int, default=20)
    ap.add_argument("--msm-steps", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-msm", action="store_true")
    ap.add_argument("--prove-k", type=int, default=22, help="create_proof leg: mini-PLONK with 2^k rows (0 = skip)")
    ap.add_argument("--prove-steps", type=int, default=3)
    args = ap.parse_args()

    import torch

    import halo2_gpu_specific_amd as h2
    from halo2_gpu_specific_amd._lib import check

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("H2_BENCH_FORCE_DIST"):  # the env knob exercises the RCCL path with one rank
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    L = h2.lib()  # raises loudly if libhalo2_hip.so is missing: no fallback path
    dev = torch.device("cuda", local_rank)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # ---------------------------------------------------------------- NTT leg (primary metric)
    log_n, n = args.log_n, 1 << args.log_n
    omega = pow(ROOT_OF_UNITY, 1 << (28 - log_n), R_MOD)
    w_f, w_i, n_inv = fr_limbs(omega), fr_limbs(pow(omega, -1, R_MOD)), fr_limbs(pow(n, -1, R_MOD))
    g = torch.Generator(device=dev)
    g.manual_seed(0x48414C4F32 + 1 + rank)
    a = torch.randint(-(2**63), 2**63 - 1, (n, 4), dtype=torch.int64, device=dev, generator=g)
    a[:, 3] &= 0x1FFFFFFFFFFFFFFF  # top limb < 2^61 < r's top limb: every row is a valid residue
    tmp = torch.empty_like(a)

    def ntt_step():
        check(L.h2_dev_ntt(a.data_ptr(), tmp.data_ptr(), vp(w_f), log_n, stream), "h2_dev_ntt")
        check(L.h2_dev_intt(a.data_ptr(), tmp.data_ptr(), vp(w_i), vp(n_inv), log_n, stream), "h2_dev_intt")

    ref_head = a[:64].clone()
    for _ in range(args.warmup):
        ntt_step()
    barrier()
    t0 = time.perf_counter()
    check(L.h2_timer_start(stream), "timer")
    for _ in range(args.steps):
        ntt_step()
    ev_ms = ctypes.c_float(0)
    check(L.h2_timer_stop(stream, ctypes.byref(ev_ms)), "timer")
    barrier()
    t1 = time.perf_counter()
    assert torch.equal(a[:64], ref_head), "inverse(forward(x)) != x"  # round-trip property at full size
    elapsed = t1 - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    fr_ops_per_transform = 3 * (n // 2) * log_n
    value = world * args.steps * 2 * fr_ops_per_transform / elapsed

    # roofline of the dominant kernel (k_ntt_pass): algorithmic bytes per SURVEY.md 8(d) =
    # 64 * n * ceil(log_n / 12) per transform, spread over the passes this build launches per transform
    passes = (log_n + 7) // 8 if log_n else 1
    alg_bytes_per_transform = 64 * n * ((log_n + 11) // 12)
    launches = args.steps * 2 * passes
    avg_launch_ms = ev_ms.value / launches
    achieved_gbs = (alg_bytes_per_transform / passes) / (avg_launch_ms * 1e-3) / 1e9

    # HBM bytes per k_ntt_pass launch from the PMC passes committed under profiles/ (separate rocprofv3
    # --pmc FETCH_SIZE / WRITE_SIZE runs, gfx950 correction applied); null when the file is absent
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r1_v10_hbm_traffic.json")) as f:
            if log_n == 24:
                traffic = json.load(f)["kernels"]["h2::k_ntt_pass"]["hbm_bytes_per_launch_corrected"]
    except (OSError, KeyError, ValueError):
        traffic = None

    out = {
        "metric": "NTT Fr-ops/s @ k=24 (forward+inverse 2^24 BN254 Fr NTT; MSM G1-adds/s under 'msm')",
        "value": value,
        "unit": "Fr-ops/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32x8 (254-bit Montgomery, integer)",
        "data": "synthetic",
        "config": {
            "workload": "BASELINE configs[2]: 2^%d-point BN254 Fr forward+inverse NTT per step, resident in HBM" % log_n,
            "log_n": log_n,
            "per_rank": "one independent polynomial per GPU (per-column sharding, no collective)",
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "k_ntt_pass",
            "achieved": achieved_gbs,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved_gbs / HBM_PEAK_GBS,
            "traffic": traffic,
            "algorithmic_bytes_per_launch": alg_bytes_per_transform / passes,
            "avg_launch_ms": avg_launch_ms,
            "launches_per_transform": passes,
            "alu": {
                "bound": "integer multiplier",
                "achieved_mul_per_s": (log_n / 2.0 + passes) * n / (passes * avg_launch_ms * 1e-3),
                "peak_mul_per_s": MUL_CEILING,
                "frac": (log_n / 2.0 + passes) * n / (passes * avg_launch_ms * 1e-3) / MUL_CEILING,
                "note": "multiplications per transform = (n/2) log2 n butterflies + one inter-pass / scaling twiddle "
                "per element per pass (the last pass forms its twiddle from two table entries: one more, not counted)",
            },
            "note": "VALU-bound in practice: ~15n 254-bit Montgomery multiplications per transform against a measured "
            "chip ceiling of 1.31e11 multiplications/s (tools/mulbench.hip, profiles/r1_mulbench.txt); see DESIGN.md",
        },
    }

    # ---------------------------------------------------------------- MSM leg (secondary)
    if not args.no_msm:
        mlog, mn = args.msm_log_n, 1 << args.msm_log_n
        sc = torch.randint(-(2**63), 2**63 - 1, (mn, 4), dtype=torch.int64, device=dev, generator=g)
        sc[:, 3] &= 0x1FFFFFFFFFFFFFFF
        bases = torch.empty((mn, 8), dtype=torch.int64, device=dev)
        check(L.h2_dev_random_points(0x48414C4F32, mn, bases.data_ptr(), stream), "h2_dev_random_points")
        sbytes = L.h2_msm_scratch_bytes(mn, 254)
        scratch = torch.empty(sbytes, dtype=torch.uint8, device=dev)
        res = np.zeros(12, dtype=np.uint64)
        c, W, nb = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
        L.h2_msm_shape(mn, 254, ctypes.byref(c), ctypes.byref(W), ctypes.byref(nb))

        def msm_step():
            check(L.h2_dev_msm(sc.data_ptr(), bases.data_ptr(), mn, 254, scratch.data_ptr(), sbytes, vp(res), stream), "h2_dev_msm")

        msm_step()
        first = res.copy()
        barrier()
        m0 = time.perf_counter()
        for _ in range(args.msm_steps):
            msm_step()
        barrier()
        m1 = time.perf_counter()
        assert jac_eq(first, res), "MSM result changed between runs"
        melapsed = m1 - m0
        # batch leg: BATCH columns committed against the same bases (the prover's shape, plonk/prover.rs:293-299),
        # pipelined by the library on two internal streams
        BATCH = 8
        per = (sbytes + 255) // 256 * 256
        scratch2 = torch.empty(2 * per, dtype=torch.uint8, device=dev)
        ptrs = (ctypes.c_void_p * BATCH)(*([sc.data_ptr()] * BATCH))
        bres = np.zeros((BATCH, 12), dtype=np.uint64)

        def batch_step():
            check(L.h2_dev_msm_batch(ptrs, BATCH, bases.data_ptr(), mn, 254, scratch2.data_ptr(), 2 * per, vp(bres), stream), "h2_dev_msm_batch")

        batch_step()
        assert all(jac_eq(first, bres[i]) for i in range(BATCH)), "batched MSM differs from the single MSM"
        barrier()
        b0 = time.perf_counter()
        for _ in range(args.msm_steps):
            batch_step()
        barrier()
        b1 = time.perf_counter()
        belapsed = b1 - b0
        if dist is not None:
            t = torch.tensor([melapsed, belapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            melapsed, belapsed = float(t[0].item()), float(t[1].item())
        adds = mn * W.value + 2 * nb.value * W.value + W.value * c.value
        out["msm"] = {
            "workload": "BASELINE configs[1]: 2^%d BN254 G1 MSM, uniform 254-bit scalars, resident in HBM" % mlog,
            "g1_adds_per_s": world * args.msm_steps * BATCH * adds / belapsed,
            "pairs_per_s": world * args.msm_steps * BATCH * mn / belapsed,
            "ms_per_msm_batched": belapsed / (args.msm_steps * BATCH) * 1e3,
            "batch": "%d MSMs over shared bases per call (h2_dev_msm_batch, two streams)" % BATCH,
            "single_msm": {
                "ms_per_msm": melapsed / args.msm_steps * 1e3,
                "g1_adds_per_s": world * args.msm_steps * adds / melapsed,
                "pairs_per_s": world * args.msm_steps * mn / melapsed,
            },
            "window_bits": c.value,
            "windows": W.value,
            "buckets_per_window": nb.value,
            "g1_adds_per_msm": adds,
            "g1_adds_formula": "n*W + 2*2^(c-1)*W + W*c (bucket accumulate + running-sum reduce + window doublings)",
            "alu_roofline": {
                "bound": "integer multiplier (not HBM: 96 B per pair)",
                "peak_adds_per_s": world * MUL_CEILING / 10.0,
                "frac": (world * args.msm_steps * BATCH * adds / belapsed) / (world * MUL_CEILING / 10.0),
                "note": "peak = measured chip ceiling of 1.31e11 254-bit Montgomery multiplications/s (profiles/r1_mulbench.txt) "
                "/ 10 multiplications per mixed XYZZ addition; field additions, sorting and the bucket reduction are not credited",
            },
            "steps": args.msm_steps,
        }
        del scratch2
        del sc, bases, scratch

    # ---------------------------------------------------------------- create_proof leg (BASELINE configs[3])
    if args.prove_k:
        try:
            from halo2_gpu_specific_amd import circuits, prover
            from halo2_gpu_specific_amd.rng import ProverRng

            pk_k = args.prove_k
            # N > 1: ONE proof over all ranks -- every rank holds the same polynomials, each MSM is range-split over the
            # ranks and folded after an all-gather of the partial points (config 5's "RCCL final reduce over xGMI")
            D = prover.Device(local_rank, force_collective=dist is not None)
            # Params::unsafe_setup on the device with a fixed toxic scalar: a real (insecure, test-only) SRS, so the timed proofs
            # are valid proofs (tests/test_gpu_plonk.py has the same k = 22 flow accepted by the reference verifier)
            params = prover.Params.unsafe_setup(D, pk_k, 0x1D0C5F0A3B7E91C2A4D6F8091B2C3D4E5F60718293A4B5C6D7E8F9010203)
            adv, fixed, copies = circuits.mini_plonk_synthesize(pk_k, alloc=D.pinned_columns)
            pk = prover.keygen(D, params, circuits.mini_plonk(), fixed, copies)
            proof = prover.create_proof_with_shplonk(D, params, pk, adv, ProverRng(1))  # warm-up (arena growth, plan caches)
            assert proof == prover.create_proof_with_shplonk(D, params, pk, adv, ProverRng(1)), "create_proof is not deterministic"
            phases = {}
            barrier()
            p0 = time.perf_counter()
            for i in range(args.prove_steps):
                prover.create_proof_with_shplonk(D, params, pk, adv, ProverRng(2 + i))
            D.sync()
            barrier()
            pelapsed = time.perf_counter() - p0
            prover.create_proof_with_shplonk(D, params, pk, adv, ProverRng(1), timings=phases)  # per-phase split (adds syncs: untimed)
            if dist is not None:
                t = torch.tensor([pelapsed], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                pelapsed = float(t[0].item())
            out["create_proof"] = {
                "workload": "BASELINE configs[3]: mini-PLONK (examples/simple-example-2.rs) with 2^%d rows, KZG/SHPLONK, witness "
                "in pinned host memory, SRS / proving key resident in HBM" % pk_k,
                "k": pk_k,
                "seconds": pelapsed / args.prove_steps,
                "scaling": "strong" if world > 1 else "n/a",
                "sharding": "one proof over %d rank(s): MSMs range-split + all-gather of partial points, transforms and "
                "elementwise passes replicated" % world,
                "proof_bytes": len(proof),
                "phases_ms": {n: round(v * 1e3, 2) for n, v in phases.items()},
                "steps": args.prove_steps,
                "srs": "Params::unsafe_setup on the device with a fixed trapdoor (g[i] = [s^i]G, g_lagrange[i] = [l_i(s)]G); "
                "tests/test_gpu_plonk.py checks proof bytes against the reference prover and verifies the k = 22 proof",
            }
            del D, params, pk, adv, fixed
        except Exception as e:  # noqa: BLE001 - the primary (NTT) line must still be printed
            out["create_proof"] = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---------------------------------------------------------------- CPU baseline (rank 0, N = 1 only)
    if world == 1 and not args.no_cpu_baseline:
        from h2util import Oracle  # the oracle is only the timed CPU baseline here, never the product path

        oracle = Oracle.get()
        cores = os.cpu_count() or 1
        clog = min(log_n, 22)
        x = oracle.random_fr(7, 1 << clog)
        wc = fr_limbs(pow(ROOT_OF_UNITY, 1 << (28 - clog), R_MOD))
        # libgomp does not always scale to every hardware thread: probe a few team sizes once, then
        # time the best one (`cores` in the report = the threads actually used)
        best_t, best_th = None, cores
        for th in sorted({cores, min(cores, 128), min(cores, 64), min(cores, 32)}, reverse=True):
            c0 = time.perf_counter()
            oracle.best_fft(x, wc, clog, threads=th)
            dt = time.perf_counter() - c0
            if best_t is None or dt < best_t:
                best_t, best_th = dt, th
            if time.perf_counter() - c0 > 20:
                break
        reps, c0 = 0, time.perf_counter()
        while reps < 3 and (time.perf_counter() - c0) < 15.0:
            oracle.best_fft(x, wc, clog, threads=best_th)
            reps += 1
        ct = (time.perf_counter() - c0) / max(reps, 1)
        out["cpu_baseline"] = {
            "value": 3 * ((1 << clog) // 2) * clog / ct,
            "unit": "Fr-ops/s",
            "cores": best_th,
            "host_threads_available": cores,
            "kind": "port",
            "sample": "oracle best_fft (C restatement of arithmetic.rs:556-705, OpenMP tasks ~ rayon) on one forward "
            "2^%d NTT, %d reps, %.3f s each (includes the oracle wrapper's input copy)" % (clog, reps, ct),
        }

        # the MSM leg's CPU twin: best_multiexp (arithmetic.rs:465-492, c = ceil(ln n) per thread chunk) on a bounded sample
        if "msm" in out:
            mlog_c = min(args.msm_log_n, 18)
            ms, mp = oracle.random_fr(11, 1 << mlog_c), oracle.random_g1(12, 1 << mlog_c)
            oracle.best_multiexp(ms, mp, threads=best_th)
            c0 = time.perf_counter()
            oracle.best_multiexp(ms, mp, threads=best_th)
            mt = time.perf_counter() - c0
            out["msm"]["cpu_baseline"] = {
                "pairs_per_s": (1 << mlog_c) / mt,
                "cores": best_th,
                "kind": "port",
                "sample": "oracle best_multiexp on 2^%d uniform pairs, %.3f s" % (mlog_c, mt),
            }

    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
