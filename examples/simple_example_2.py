"""keygen -> create_proof -> verify on the mini-PLONK circuit of the reference's examples/simple-example-2.rs:177-288
(3 advice columns a, b, c with equality, 4 fixed columns sm, sa, sb, sc, one gate a*sa + b*sb + a*b*sm - c*sc,
2^(k-4) multiply / add pairs with two copy constraints each, witness a = 5), on one MI355X.

The prover is the product path (halo2-gpu-specific_amd/prover.py over libhalo2_hip.so: there is no CPU fallback); the
verifier is the independent big-integer one the tests use (tests/ref_plonk.py: test infrastructure, not product code).

usage: python examples/simple_example_2.py [k] [gwc|shplonk]"""
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

k = int(sys.argv[1]) if len(sys.argv) > 1 else 8
use_gwc = (sys.argv[2] if len(sys.argv) > 2 else "shplonk") == "gwc"
S = 0x1D0C5F0A3B7E91C2A4D6F8091B2C3D4E5F60718293A4B5C6D7E8F9010203   # Params::unsafe_setup's toxic scalar, fixed here

D = prover.Device()
t0 = time.perf_counter()
params = prover.Params.unsafe_setup(D, k, S)                       # poly/commitment.rs:56-124, on the device
advice, fixed, copies = circuits.mini_plonk_synthesize(k, alloc=D.pinned_columns)
pk = prover.keygen(D, params, circuits.mini_plonk(), fixed, copies)  # keygen_vk + keygen_pk
print("setup + keygen: %.3f s" % (time.perf_counter() - t0))

t0 = time.perf_counter()
proof = prover.create_proof_ext(D, params, pk, advice, ProverRng(), use_gwc)
D.sync()
print("create_proof (%s): %.1f ms, %d bytes" % ("GWC" if use_gwc else "SHPLONK", (time.perf_counter() - t0) * 1e3, len(proof)))

if k <= 12:   # the reference-side check is pure Python: seconds at k = 8, minutes beyond 2^12 rows
    import ref_plonk as rp

    adv_r, fixed_r, copies_r = rp.MiniPlonk.synthesize(k)
    rpk = rp.keygen(rp.MiniPlonk, k, S, fixed_r, copies_r)
    ok = rp.verify_proof(rpk, proof, use_gwc=use_gwc)
    tampered = bytearray(proof)
    tampered[40] ^= 1
    try:
        bad = rp.verify_proof(rpk, bytes(tampered), use_gwc=use_gwc)
    except AssertionError:     # the flipped byte no longer decodes to a curve point / canonical scalar
        bad = False
    print("verify_proof: %s;  tampered proof: %s" % (ok, bad))
    assert ok and not bad
else:
    print("verify_proof: skipped (k > 12); tests/test_gpu_plonk.py verifies k = 16 .. 24 through the trapdoor opening check")
