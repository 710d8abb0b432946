#!/usr/bin/env python3
"""Generate tests/golden/proof_kat.json: whole proofs produced by the big-integer reference prover
(tests/ref_plonk.py) for fixed circuits, trapdoor and blinding seeds.  The fixtures pin (a) the reference prover
against accidental changes of its conventions and (b) the device prover (tests/test_gpu_plonk.py), which has to
reproduce the same bytes from the same inputs.

Run:  python tests/golden/gen_proof_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ref_plonk as rp  # noqa: E402
from halo2_gpu_specific_amd.rng import ProverRng  # noqa: E402

TRAPDOOR = 0x1D0C5F0A3B7E91C2A4D6F8091B2C3D4E5F60718293A4B5C6D7E8F9010203
CASES = [("mini-plonk", 4, 1, "shplonk"), ("mini-plonk", 5, 2, "gwc"), ("rot-gate", 5, 3, "shplonk"),
         ("lookup-shuffle", 5, 4, "shplonk"), ("lookup-shuffle", 5, 5, "gwc")]
CIRCUITS = {"mini-plonk": rp.MiniPlonk, "rot-gate": rp.RotGate, "lookup-shuffle": rp.LookupShuffle}


def main():
    out = []
    for name, k, seed, scheme in CASES:
        cs = CIRCUITS[name]
        syn = cs.synthesize(k)
        adv, fixed, copies = syn[:3]
        inst = syn[3] if len(syn) > 3 else []
        pk = rp.keygen(cs, k, TRAPDOOR, fixed, copies)
        proof = rp.create_proof(pk, adv, ProverRng(seed), use_gwc=scheme == "gwc", instances=inst)
        assert rp.verify_proof(pk, proof, use_gwc=scheme == "gwc", instances=inst, pairing=True)
        out.append({"circuit": name, "k": k, "seed": seed, "scheme": scheme, "trapdoor": hex(TRAPDOOR),
                    "instances": [[hex(v) for v in col] for col in inst],
                    "vk_digest": hex(pk.transcript_repr), "proof": proof.hex()})
    with open(os.path.join(HERE, "proof_kat.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", len(out), "proofs")


if __name__ == "__main__":
    main()
