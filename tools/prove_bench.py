"""create_proof wall-clock on one GPU for the mini-PLONK circuit (BASELINE config 4: k = 22), per phase.
usage: python tools/prove_bench.py [k] [reps]      (SRS from the device-side unsafe_setup with a fixed trapdoor)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before HIP initialises (halo2-gpu-specific_amd/__init__.py says why)
import torch  # noqa: E402  (first, so that the library binds to torch's HIP runtime)

torch.cuda.init()

from halo2_gpu_specific_amd import circuits, prover  # noqa: E402
from halo2_gpu_specific_amd.rng import ProverRng  # noqa: E402


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 22
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    D = prover.Device()
    t0 = time.perf_counter()
    params = prover.Params.unsafe_setup(D, k, 0x1D0C5F0A3B7E91C2A4D6F8091B2C3D4E5F60718293A4B5C6D7E8F9010203)
    t1 = time.perf_counter()
    adv, fixed, copies = circuits.mini_plonk_synthesize(k, alloc=D.pinned_columns)
    t2 = time.perf_counter()
    pk = prover.keygen(D, params, circuits.mini_plonk(), fixed, copies)
    t3 = time.perf_counter()
    print("k=%d  srs %.3fs (tables %.1f GiB)  synthesize %.3fs  keygen %.3fs" % (k, t1 - t0, params.table_bytes / 2**30, t2 - t1, t3 - t2))
    for rep in range(reps):
        timings = {}
        ta = time.perf_counter()
        proof = prover.create_proof_with_shplonk(D, params, pk, adv, ProverRng(rep), timings=timings if rep and not os.environ.get("H2_PROVE_BENCH_NO_TIMINGS") else None)
        D.sync()
        tb = time.perf_counter()
        print("rep %d: create_proof %.1f ms  (%d bytes)  %s" % (rep, (tb - ta) * 1e3, len(proof),
                                                                 {n: round(t * 1e3, 1) for n, t in timings.items()}))
    print("peak device memory: %.2f GiB" % (torch.cuda.max_memory_allocated() / 2**30))


if __name__ == "__main__":
    main()
