import os, sys, time, hashlib
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
torch.cuda.init()
from halo2_gpu_specific_amd import circuits, prover
from halo2_gpu_specific_amd.rng import ProverRng
k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
quads = 16
D = prover.Device(0)
params = prover.Params.unsafe_setup(D, k, 0x1D0C5F0A3B7E91C2A4D6F8091B2C3D4E5F60718293A4B5C6D7E8F9010203)
cs = circuits.wide(quads)
adv, fixed, copies = circuits.wide_synthesize(k, quads, alloc=D.pinned_columns)
cadv = circuits.wide_synthesize(k, quads, alloc=D.pinned_columns, compact=True)[0]
pk = prover.keygen(D, params, cs, fixed, copies)
for name, a in (("wide32", adv), ("compact", cadv)):
    for grp in sys.argv[2:] or ["0"]:
        if grp != "0": os.environ["H2_ADVICE_GROUP"] = grp
        else: os.environ.pop("H2_ADVICE_GROUP", None)
        proof = prover.create_proof_with_shplonk(D, params, pk, a, ProverRng(1))
        D.sync(); t0 = time.perf_counter()
        for i in range(3): prover.create_proof_with_shplonk(D, params, pk, a, ProverRng(2 + i))
        D.sync(); sec = (time.perf_counter() - t0) / 3
        ph = {}
        prover.create_proof_with_shplonk(D, params, pk, a, ProverRng(1), timings=ph)
        print(name, "group", grp, "%.1f ms" % (sec * 1e3), hashlib.sha256(proof).hexdigest()[:12], {n: round(v * 1e3, 1) for n, v in ph.items()}, flush=True)
