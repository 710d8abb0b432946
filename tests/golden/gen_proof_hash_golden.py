#!/usr/bin/env python3
"""Generate tests/golden/proof_hash_kat.json: SHA-256 of whole proofs produced by the big-integer reference prover
(tests/ref_plonk.py) at sizes where running it inside the test suite would take minutes (k = 12 .. 18).  The device
prover (tests/test_gpu_plonk.py::test_proof_bytes_match_committed_hashes) has to reproduce the same bytes from the same
circuit, trapdoor and blinding seed -- verifier acceptance alone would not catch a wrong-but-valid blinding or
ordering change.

Run:  python tests/golden/gen_proof_hash_golden.py           (k <= 18: about 6 minutes on one core, pure Python)
      python tests/golden/gen_proof_hash_golden.py large     (adds k = 20 and k = 22 -- BASELINE configs[3]'s size; ~25 min,
                                                              ~12 GB of Python integers)

The `large` cases run the SAME big-integer prover with one substitution: `ref_plonk.fft` (a recursive Python DFT, 60 % of
its time) is replaced by the C oracle's `best_fft` (oracle/oracle.c, the restatement of arithmetic.rs:556-705 that
tests/test_oracle_golden.py pins to the naive big-integer DFT vectors) -- test infrastructure calling test infrastructure.
Everything else (blinding, grand products, the quotient loop over the extended domain, Horner evaluations, SHPLONK) stays
Python integers; `"fft": "oracle"` marks those entries.  The substitution is checked against the pure-Python run at k = 12
(same bytes) before anything large is hashed.
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


LARGE_CASES = [("mini-plonk", 20, 20, "shplonk"), ("mini-plonk", 22, 22, "shplonk")]


def oracle_fft():
    """ref_plonk.fft with the oracle's best_fft underneath: list of ints in, list of ints out"""
    import numpy as np
    from h2util import Oracle

    oracle = Oracle.get()

    def fft(a, omega):
        n = len(a)
        log_n = n.bit_length() - 1
        assert 1 << log_n == n
        buf = np.frombuffer(bytearray(b"".join(v.to_bytes(32, "little") for v in a)), dtype=np.uint64).reshape(n, 4)
        oracle.lib.oracle_from_repr_batch(buf.ctypes.data, n, 0)
        w = np.frombuffer(bytearray(omega.to_bytes(32, "little")), dtype=np.uint64).reshape(1, 4)
        oracle.lib.oracle_from_repr_batch(w.ctypes.data, 1, 0)
        oracle.lib.oracle_best_fft(buf.ctypes.data, w.ctypes.data, log_n, oracle.threads)
        oracle.lib.oracle_to_repr_batch(buf.ctypes.data, n, 0)
        raw = buf.tobytes()
        return [int.from_bytes(raw[32 * i:32 * i + 32], "little") for i in range(n)]

    return fft


def main():
    large = len(sys.argv) > 1 and sys.argv[1] == "large"
    path = os.path.join(HERE, "proof_hash_kat.json")
    out = []
    cases = [c + ("python",) for c in CASES]
    if large:
        with open(path) as f:
            out = [e for e in json.load(f) if e.get("fft", "python") == "python"]
        python_fft, fast = rp.fft, oracle_fft()
        # the substitution reproduces the pure-Python prover's bytes where both can run
        adv, fixed, copies = rp.MiniPlonk.synthesize(12)
        pk = rp.keygen(rp.MiniPlonk, 12, TRAPDOOR, fixed, copies)
        want = rp.create_proof(pk, adv, ProverRng(12))
        rp.fft = fast
        pk2 = rp.keygen(rp.MiniPlonk, 12, TRAPDOOR, fixed, copies)
        assert pk2.transcript_repr == pk.transcript_repr and rp.create_proof(pk2, adv, ProverRng(12)) == want
        print("oracle-FFT prover == pure-Python prover at k = 12", flush=True)
        cases = [c + ("oracle",) for c in LARGE_CASES]
    for name, k, seed, scheme, fft_kind in cases:
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
                    "sha256": hashlib.sha256(proof).hexdigest(), "fft": fft_kind})
        print(name, k, scheme, "%.1f s" % (time.time() - t0), flush=True)
        with open(path, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
