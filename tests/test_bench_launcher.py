"""`python3 bench.py --gpus N` with NO launcher around it must start N ranks itself (the reference splits over N_GPU
devices inside one call: arithmetic.rs:413-440, plonk/prover.rs:56-74) and print one line with n_gpus == N.

On the GPU box the multi-rank flow is dry-run with two gloo ranks sharing the one GPU (H2_BENCH_BACKEND=gloo): rank start-up,
barriers, the reduction of the timings, ONE proof over the ranks (mini-PLONK and the wide circuit), the relayed JSON line.
Without a GPU the ranks cannot run: the launcher must then exit non-zero (a failed rank is a failed run), promptly."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run_bench(args, env_extra, timeout):
    env = dict(os.environ, **env_extra)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, BENCH] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=timeout, cwd=ROOT)
    return p, time.perf_counter() - t0


@pytest.mark.gpu
def test_bare_gpus_2_starts_two_ranks():
    p, _ = run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--k24", "0", "--prove-k", "12", "--wide-k", "10",
                      "--wide-quads", "2", "--log-n", "16", "--msm-log-n", "14", "--no-cpu-baseline"],
                     {"H2_BENCH_BACKEND": "gloo"}, 600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 2
    assert "2 rank(s)" in line["create_proof"]["sharding"], line["create_proof"]
    assert line["create_proof"]["verified"] is True
    wide = line["create_proof_wide"]
    assert "error" not in wide, wide
    assert "2 rank(s)" in wide["sharded"]["sharding"] and wide["verified"] is True
    # round 5: communication next to compute per phase, the replicas leg, the whole metric inside `config`
    comm = line["create_proof"]["communication"]
    assert comm["comm_ms_total"] > 0 and set(comm["phases"]) == set(line["create_proof"]["phases_ms"])
    assert any(p["calls"] for p in comm["phases"].values()) and wide["sharded"]["communication"]["comm_ms_total"] > 0
    rep = line["create_proof_replicas"]
    assert "error" not in rep and rep["proofs_per_s"] > 0 and rep["proof_bytes"] == line["create_proof"]["proof_bytes"]
    head = line["config"]["headline"]
    assert head["create_proof_k12_seconds"] == line["create_proof"]["seconds"] and head["msm_g1_adds_per_s"] > 0
    assert head["create_proof_wide_k10_seconds"] == wide["sharded"]["seconds"]
    # the same circuits on one rank: the sharded proofs carry the same bytes
    q, _ = run_bench(["--gpus", "1", "--steps", "1", "--warmup", "1", "--k24", "0", "--prove-k", "0", "--wide-k", "10",
                      "--wide-quads", "2", "--log-n", "16", "--no-msm", "--no-cpu-baseline"], {}, 600)
    assert q.returncode == 0, q.stderr[-2000:]
    one = json.loads(q.stdout.strip().splitlines()[-1])
    assert one["n_gpus"] == 1
    assert one["create_proof_wide"]["resident"]["proof_sha256"] == wide["sharded"]["proof_sha256"]
    assert one["create_proof_wide"]["resident_compact_witness"]["same_proof_bytes"] is True


@pytest.mark.gpu
def test_rccl_refuses_more_ranks_than_gpus():
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer GPUs than ranks")
    p, dt = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--k24", "0", "--prove-k", "0", "--wide-k", "0",
                       "--log-n", "12", "--no-msm", "--no-cpu-baseline"], {"H2_BENCH_RANK_GRACE_S": "5"}, 300)
    assert p.returncode != 0
    assert "need 2 GPUs" in p.stderr


def test_launcher_reports_failed_ranks_without_a_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    p, dt = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"H2_BENCH_RANK_GRACE_S": "2"}, 300)
    assert p.returncode != 0
    assert "rank 0 exited" in p.stderr and "rank 1 exited" in p.stderr
    assert p.stdout.strip() == ""       # no result line from a failed run
