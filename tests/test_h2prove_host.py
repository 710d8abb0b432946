"""tools/h2prove.cpp -- the mini-PLONK prover over the C ABI alone (plain C++, no Python / torch / HIP headers) -- on a machine
without a GPU: `h2prove --host-check k seed` prints everything that tool computes on the HOST (its own Blake2b, transcript and
challenge reduction, the seeded randomness, the domain scalars, the verifying-key digest, the permutation mapping, the witness,
and the circuit's program and serialisation it carries as constants) and each item is compared here with the Python side the
device prover uses.  The proof itself is tests/test_gpu_h2prove.py."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

from halo2_gpu_specific_amd import circuits, formats, prover
from halo2_gpu_specific_amd.circuit import compile_evaluator
from halo2_gpu_specific_amd.rng import ProverRng
from halo2_gpu_specific_amd.transcript import Blake2bWrite

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "h2prove")


def build_tool():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools"), "h2prove"], stdout=subprocess.DEVNULL)
    return TOOL


def fnv(values):
    acc = 1469598103934665603
    for v in values:
        acc = ((acc ^ int(v)) * 1099511628211) & ((1 << 64) - 1)
    return acc


@pytest.mark.parametrize("k,seed", [(8, 12), (11, 0xFEEDFACECAFE)])
def test_host_side_of_h2prove_equals_the_python_side(k, seed):
    out = subprocess.check_output([build_tool(), "--host-check", str(k), str(seed)], text=True)
    got = {ln.split(" ", 1)[0]: ln.split(" ", 1)[1] for ln in out.strip().splitlines()}
    # Blake2b-512 with a personalisation, a short input and one that crosses block boundaries
    assert got["blake2b_abc"] == hashlib.blake2b(b"abc", digest_size=64, person=b"Halo2-Transcript").hexdigest()
    blob = bytes((i * 7 + 3) & 255 for i in range(1000))
    assert got["blake2b_1000"] == hashlib.blake2b(blob, digest_size=64, person=b"Halo2-Verify-Key").hexdigest()
    # the seeded stream: u16 draws, Fr::random, the random polynomial's key (a stream of its own)
    rng = ProverRng(seed)
    assert got["rng_u16"] == "%d %d" % (rng.u16(), rng.u16())
    assert int(got["rng_fr"], 16) == rng.fr()
    assert got["rng_poly_key"] == rng.random_poly_key().hex()
    # EvaluationDomain::new
    dom = prover.Domain(k, 3)
    f = got["domain"].split()
    assert (int(f[0]), int(f[1])) == (dom.k, dom.extended_k)
    assert [int(f[i], 16) for i in (3, 5, 7, 9)] == [dom.omega, dom.extended_omega, dom.t_evaluations[0], dom.t_evaluations[-1]]
    # transcript: common_scalar, write_point (compressed encoding), two squeezes
    tr = Blake2bWrite()
    tr.common_scalar(12345)
    tr.write_point((1, 2))
    c1, c2 = tr.squeeze_challenge_scalar(), tr.squeeze_challenge_scalar()
    f = got["challenge"].split()
    assert int(f[0], 16) == c1 and f[2] == bytes(tr.writer).hex() and int(got["challenge2"], 16) == c2
    # the verifying key's digest over given commitments
    cs = circuits.mini_plonk()
    assert int(got["vk_digest_of_generators"], 16) == prover.vk_digest(cs, dom, [(1, 2), (1, 2)], [(1, 2)])
    # witness, copy constraints -> permutation mapping
    adv, fixed, copies = circuits.mini_plonk_synthesize(k)
    mc, mr = prover.permutation_mapping(3, 1 << k, copies)
    assert int(got["mapping_fnv"], 16) == fnv(np.stack([mc, mr], axis=-1).reshape(-1))
    assert int(got["witness_fnv"], 16) == fnv(np.concatenate([c.reshape(-1) for c in list(adv) + list(fixed)]))
    # the circuit the tool carries as constants: serialisation and evaluator program
    assert got["cs_store"] == formats.cs_store(cs).hex()
    g, parts, lookups, shuffles = compile_evaluator(cs)
    assert not lookups and not shuffles and list(g.rotations) == [0]
    want = ["c:%x" % c for c in g.constants]
    want += ["calc:%d,%d,%d,%d,%d,%d,%d,%d,%d" % (c.op, c.a.kind, c.a.index, c.a.rot, c.b.kind, c.b.index, c.b.rot, c.challenge, c.power)
             for c in g.calculations]
    want += ["vp:%d,%d,%d" % (p.kind, p.index, p.rot) for p in parts]
    want += ["aq:%d,%d" % q for q in cs.advice_queries] + ["fq:%d,%d" % q for q in cs.fixed_queries]
    assert got["program"].split() == want
    assert (cs.degree(), cs.blinding_factors(), cs.num_advice, cs.num_fixed) == (3, 5, 3, 4) and list(cs.perm_columns) == [("advice", i) for i in range(3)]
    assert got["interpolate"] == "1"


def test_h2prove_links_the_library_and_nothing_of_hip_or_python():
    """the point of the tool: the C ABI alone suffices -- its only non-system dependency is libhalo2_hip.so"""
    out = subprocess.check_output(["readelf", "-d", build_tool()], text=True)
    needed = [ln.split("[")[1].rstrip("]") for ln in out.splitlines() if "(NEEDED)" in ln]
    assert "libhalo2_hip.so" in needed
    assert not [n for n in needed if "hip64" in n or "python" in n or "torch" in n or "hsa" in n], needed
    src = open(os.path.join(ROOT, "tools", "h2prove.cpp")).read()
    assert "hip/hip_runtime" not in src and "#include <hip" not in src
