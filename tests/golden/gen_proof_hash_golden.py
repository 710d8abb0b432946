#!/usr/bin/env python3
"""Generate tests/golden/proof_hash_kat.json: SHA-256 of whole proofs produced by the big-integer reference prover
(tests/ref_plonk.py) at sizes where running it inside the test suite would take minutes (k = 12 .. 18).  The device
prover (tests/test_gpu_plonk.py::test_proof_bytes_match_committed_hashes) has to reproduce the same bytes from the same
circuit, trapdoor and blinding seed -- verifier acceptance alone would not catch a wrong-but-valid blinding or
ordering change.

Run:  python tests/golden/gen_proof_hash_golden.py     (about 6 minutes on one core, pure Python)
"""
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ref_plonk as rp  # noqa: E402
from halo2_gpu_specific_amd.rng import ProverRng  # noqa: E402

TRAPDOOR = 0x1D0C5F0A3B7E91C2A4D6F8091B2C3D4E5F60718293A4B5C6D7E8F9010203
CASES = [("mini-plonk", 12, 12, "shplonk"), ("mini-plonk", 14, 14, "gwc"), ("rot-gate", 12, 21, "gwc"),
         ("lookup-shuffle", 10, 31, "shplonk"), ("mini-plonk", 16, 16, "shplonk"), ("mini-plonk", 18, 18, "shplonk")]
CIRCUITS = {"mini-plonk": rp.MiniPlonk, "rot-gate": rp.RotGate, "lookup-shuffle": rp.LookupShuffle}


def main():
    out = []
    for name, k, seed, scheme in CASES:
        t0 = time.time()
        cs = CIRCUITS[name]
        syn = cs.synthesize(k)
        adv, fixed, copies = syn[:3]
        inst = syn[3] if len(syn) > 3 else []
        pk = rp.keygen(cs, k, TRAPDOOR, fixed, copies)
        proof = rp.create_proof(pk, adv, ProverRng(seed), use_gwc=scheme == "gwc", instances=inst)
        assert rp.verify_proof(pk, proof, use_gwc=scheme == "gwc", instances=inst)
        out.append({"circuit": name, "k": k, "seed": seed, "scheme": scheme, "trapdoor": hex(TRAPDOOR),
                    "vk_digest": hex(pk.transcript_repr), "length": len(proof),
                    "sha256": hashlib.sha256(proof).hexdigest()})
        print(name, k, scheme, "%.1f s" % (time.time() - t0), flush=True)
        with open(os.path.join(HERE, "proof_hash_kat.json"), "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
