"""the wide circuit's proof over N gloo ranks sharing one GPU against the single-device proof (hashes): a reproduction harness
usage: wide_ranks_repro.py <ranks> <k> <quads> <seed> <pinned 0|1>   (ranks = 1: the single-device hash)"""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ranks, k, quads, seed, pinned = (int(v) for v in sys.argv[1:6])
    if ranks > 1 and "RANK" not in os.environ:
        import socket

        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                  env=dict(os.environ, RANK=str(r), WORLD_SIZE=str(ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)))
                 for r in range(ranks)]
        sys.exit(max(p.wait() for p in procs))
    import torch

    torch.cuda.set_device(0)
    if ranks > 1:
        import torch.distributed as dist

        dist.init_process_group("gloo")
    from halo2_gpu_specific_amd import circuits, prover
    from halo2_gpu_specific_amd.rng import ProverRng

    trapdoor = 0x1D0C5F0A3B7E91C2A4D6F8091B2C3D4E5F60718293A4B5C6D7E8F9010203
    D = prover.Device(0)
    cs = circuits.wide(quads)
    params = prover.Params.unsafe_setup(D, k, trapdoor)
    adv, fixed, copies = circuits.wide_synthesize(k, quads, alloc=D.pinned_columns if pinned else None)
    pk = prover.keygen(D, params, cs, fixed, copies)
    out = []
    for rep in range(int(os.environ.get("REPS", "2"))):
        ph = {}
        proof = prover.create_proof_with_shplonk(D, params, pk, adv, ProverRng(seed), timings=ph if os.environ.get("TIMINGS") else None)
        out.append(hashlib.sha256(proof).hexdigest()[:16])
    sys.stdout.write("rank %s of %d k %d quads %d seed %d pinned %d: %s\n" % (os.environ.get("RANK", "0"), ranks, k, quads, seed, pinned, " ".join(out)))
    sys.stdout.flush()
    if ranks > 1:
        dist.barrier()
        dist.destroy_process_group()


main()
