"""worker of tests/test_gpu_cpu_prover.py::test_full_size_proof_bytes_equal_cpu_proof_bytes[24]: the CPU side of the k = 24 twin
(keygen + create_proof over the C oracle, tests/oracle_prover.py: ~3 minutes of host cores) in a process of its own, started by
tests/conftest.py when the session begins so that it runs UNDER the other GPU tests instead of in front of them; the test
itself runs last, makes the device proof and compares.  TEST INFRASTRUCTURE: the only GPU work here is the SRS (the same
Params::unsafe_setup the test uses), handed to the oracle device as host arrays; everything after that is the CPU path.

usage: cpu_prover_worker.py <k> <seed> <use_gwc 0|1> <out.json>       (H2_ORACLE_THREADS: the host threads it may use)"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ["H2_MSM_TABLES"] = "0"          # the SRS only: no shifted-base tables in this process


def main():
    k, seed, use_gwc, out = int(sys.argv[1]), int(sys.argv[2]), bool(int(sys.argv[3])), sys.argv[4]
    import torch  # noqa: F401  (first: the library binds to torch's HIP runtime)

    import oracle_prover as op
    from halo2_gpu_specific_amd import circuits, prover
    from halo2_gpu_specific_amd.rng import ProverRng
    from test_plonk_host import S_TRAPDOOR

    D = prover.Device()
    params = prover.Params.unsafe_setup(D, k, S_TRAPDOOR)
    cpu = op.OracleDevice(**({"eval_cache": 0} if k >= 24 else {}))
    cparams = op.params_like(cpu, params)
    del params, D
    torch.cuda.empty_cache()              # the GPU is the other process's from here on
    cs = circuits.mini_plonk()
    adv, fixed, copies = circuits.mini_plonk_synthesize(k)
    t0 = time.perf_counter()
    cpk = op.keygen(cpu, cparams, cs, fixed, copies)
    t_keygen = time.perf_counter() - t0
    t0 = time.perf_counter()
    proof = prover.create_proof_ext(cpu, cparams, cpk, adv, ProverRng(seed), use_gwc)
    t_cpu = time.perf_counter() - t0
    doc = {"k": k, "seed": seed, "use_gwc": use_gwc, "proof": bytes(proof).hex(), "fixed_commitments": repr(cpk.fixed_commitments),
           "perm_commitments": repr(cpk.perm_commitments), "transcript_repr": repr(cpk.transcript_repr), "keygen_seconds": t_keygen,
           "cpu_seconds": t_cpu, "threads": cpu.L.threads}
    with open(out + ".tmp", "w") as f:
        json.dump(doc, f)
    os.replace(out + ".tmp", out)


if __name__ == "__main__":
    main()
