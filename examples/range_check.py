"""keygen -> create_proof -> verify on the circuit of the reference's examples/range-check.rs: one advice column
range-checked into 0 ..= 65535 with step 2 by `advice_column_range` (plonk/circuit.rs:1769-1826) -- a companion column that
holds the same values sorted, a degree-4 gate (starts at min, ends at max, neighbours differ by at most step) and a
shuffle between the two -- 65535 random values, k = 18, on one MI355X.  `create_proof` completes the witness as the
reference does (plonk/prover.rs:1699-1783): the range is planted in the unused cells and the companion is sorted.

The verifier is the big-integer one the tests use (tests/ref_plonk.py: test infrastructure, not product code), fed the
verifying key the device keygen produced.

usage: python examples/range_check.py [k >= 17] [proofs]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before HIP initialises (halo2-gpu-specific_amd/__init__.py says why)
import torch  # noqa: E402  (first: the library binds to torch's HIP runtime)

torch.cuda.init()

from halo2_gpu_specific_amd import circuits, prover  # noqa: E402
from halo2_gpu_specific_amd.rng import ProverRng  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 18
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
S = 0x1D0C5F0A3B7E91C2A4D6F8091B2C3D4E5F60718293A4B5C6D7E8F9010203   # Params::unsafe_setup's toxic scalar, fixed here

D = prover.Device()
t0 = time.perf_counter()
params = prover.Params.unsafe_setup(D, k, S)
cs = circuits.range_check()
advice, fixed, copies = circuits.range_check_synthesize(k, alloc=D.pinned_columns)
pk = prover.keygen(D, params, cs, fixed, copies)
D.sync()
print("setup + keygen: %.3f s (degree %d, %d advice / %d fixed columns, %d shuffle group)" % (
    time.perf_counter() - t0, cs.degree(), cs.num_advice, cs.num_fixed, len(cs.shuffles)))
for rep in range(reps):
    timings = {}
    t0 = time.perf_counter()
    proof = prover.create_proof_with_shplonk(D, params, pk, advice, ProverRng(rep), timings=timings if rep else None)
    D.sync()
    print("create_proof: %.1f ms, %d bytes %s" % ((time.perf_counter() - t0) * 1e3, len(proof),
                                                 {a: round(b * 1e3, 1) for a, b in timings.items()}))

import ref_plonk as rp  # noqa: E402

vk = rp.Keys()
vk.cs, vk.dom, vk.s = rp.range_check_class(0, 0xFFFF, 2), rp.Domain(k, cs.degree()), S
vk.fixed_commitments, vk.perm_commitments, vk.transcript_repr = pk.fixed_commitments, pk.perm_commitments, pk.transcript_repr
ok = rp.verify_proof(vk, proof)
print("verify_proof:", ok)
assert ok
