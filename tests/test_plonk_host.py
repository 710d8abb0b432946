"""CPU tests of the proof-level test infrastructure and of the product's host logic around the device path:
the big-integer big-integer prover is accepted by the (trapdoor) verifier, rejects tampering and a bad witness,
and the product's host-side pieces (transcript, permutation mapping, gate flattening, domain scalars, rng)
agree with the independent restatements in ref_plonk.py."""
import ctypes
import os
import random

import numpy as np
import pytest

import ref_plonk as rp
from halo2_gpu_specific_amd import circuit as hc
from halo2_gpu_specific_amd import circuits, evaluation as ev, prover, transcript
from halo2_gpu_specific_amd.rng import ProverRng

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from product_circuits import lookup_shuffle_cs, rot_gate_cs  # noqa: E402,F401 (re-exported for test_gpu_plonk)

S_TRAPDOOR = 0x1D0C5F0A3B7E91C2A4D6F8091B2C3D4E5F60718293A4B5C6D7E8F9010203


def test_big_integer_prover_with_lookups_shuffles_instances():
    cs, k = rp.LookupShuffle, 5
    adv, fixed, copies, inst = cs.synthesize(k)
    pk = rp.keygen(cs, k, S_TRAPDOOR, fixed, copies)
    for use_gwc in (False, True):
        proof = rp.create_proof(pk, adv, ProverRng(4), use_gwc=use_gwc, instances=inst)
        assert rp.verify_proof(pk, proof, use_gwc=use_gwc, instances=inst)
        assert not rp.verify_proof(pk, proof, use_gwc=use_gwc, instances=[[43, 7]])
    assert rp.verify_proof(pk, proof, use_gwc=True, instances=inst, pairing=True)
    bad = [c[:] for c in adv]
    bad[4][2] = 5                                   # e[2] is not in the table u
    with pytest.raises(KeyError):
        rp.create_proof(pk, bad, ProverRng(4), instances=inst)
    bad = [c[:] for c in adv]
    bad[10][1] += 1                                 # p2 is no longer a permutation of p
    with pytest.raises(AssertionError, match="shuffle"):
        rp.create_proof(pk, bad, ProverRng(4), instances=inst)
    bad = [c[:] for c in adv]
    bad[11][0] = 41                                 # w[0] != the public input
    assert not rp.verify_proof(pk, rp.create_proof(pk, bad, ProverRng(4), instances=inst), instances=inst)


def test_lookup_shuffle_constraint_system_mirrors_reference():
    cs, ref = lookup_shuffle_cs(), rp.LookupShuffle
    assert cs.advice_queries == ref.advice_queries and cs.fixed_queries == ref.fixed_queries
    assert cs.instance_queries == ref.instance_queries and cs.perm_columns == ref.perm_columns
    assert cs.degree() == ref.degree and cs.blinding_factors() == ref.blinding_factors
    g, parts, lookups, shuffles = hc.compile_evaluator(cs)
    assert [len(p) for _, p, _ in lookups] == [2, 1] and len(shuffles) == 1
    # run the program + the argument calculations on random values against the reference's closed formulas
    rnd = random.Random(8)
    theta, beta = rnd.randrange(rp.R), rnd.randrange(rp.R)
    vals = {}
    adv = lambda c, r: vals.setdefault(("a", c, r), rnd.randrange(rp.R))  # noqa: E731
    fix = lambda c, r: vals.setdefault(("f", c, r), rnd.randrange(rp.R))  # noqa: E731
    ins = lambda c, r: vals.setdefault(("i", c, r), rnd.randrange(rp.R))  # noqa: E731
    inter = []

    def get(v):
        if v.kind == ev.VS_CONSTANT:
            return g.constants[v.index]
        if v.kind == ev.VS_INTERMEDIATE:
            return inter[v.index]
        return {ev.VS_FIXED: fix, ev.VS_ADVICE: adv, ev.VS_INSTANCE: ins}[v.kind](v.index, g.rotations[v.rot])

    def run(c):
        a = get(c.a)
        if c.op == ev.CALC_ADD: return (a + get(c.b)) % rp.R
        if c.op == ev.CALC_SUB: return (a - get(c.b)) % rp.R
        if c.op == ev.CALC_MUL: return a * get(c.b) % rp.R
        if c.op == ev.CALC_NEGATE: return -a % rp.R
        if c.op == ev.CALC_LC_THETA: return (a * theta + get(c.b)) % rp.R
        if c.op == ev.CALC_ADD_CHALLENGE: return (a + beta) % rp.R
        if c.op == ev.CALC_LC_CHALLENGE: return (a + pow(beta, c.power, rp.R)) * get(c.b) % rp.R
        return a

    for c in g.calculations:
        inter.append(run(c))
    assert [get(p) for p in parts] == ref.gates(adv, fix, ins)
    for (table, prods, sums), lk in zip(lookups, ref.lookups):
        assert run(table) == (rp._compress(lk["table"](adv, fix, ins), theta) + beta) % rp.R
        for pc, sc, st in zip(prods, sums, lk["input_sets"]):
            phi = [(rp._compress(e(adv, fix, ins), theta) + beta) % rp.R for e in st]
            prod = 1
            for v in phi:
                prod = prod * v % rp.R
            assert run(pc) == prod
            assert run(sc) == sum(prod * rp.inv(v) for v in phi) % rp.R
    (ia, sb), group = shuffles[0], ref.shuffles[0]
    want_a = want_b = 1
    for i, (inp, shf) in enumerate(group):
        want_a = want_a * (rp._compress(inp(adv, fix, ins), theta) + pow(beta, i + 1, rp.R)) % rp.R
        want_b = want_b * (rp._compress(shf(adv, fix, ins), theta) + pow(beta, i + 1, rp.R)) % rp.R
    assert (run(ia), run(sb)) == (want_a, want_b)


@pytest.mark.parametrize("cs,k", [(rp.MiniPlonk, 4), (rp.MiniPlonk, 5), (rp.RotGate, 5)])
def test_big_integer_prover_is_accepted(cs, k):
    adv, fixed, copies = cs.synthesize(k)
    pk = rp.keygen(cs, k, S_TRAPDOOR, fixed, copies)
    proof = rp.create_proof(pk, adv, ProverRng(11))
    assert rp.verify_proof(pk, proof)
    assert proof == rp.create_proof(pk, adv, ProverRng(11)), "same seed, same proof"
    assert proof != rp.create_proof(pk, adv, ProverRng(12))
    gwc = rp.create_proof(pk, adv, ProverRng(11), use_gwc=True)
    assert rp.verify_proof(pk, gwc, use_gwc=True) and gwc[:len(proof) - 64] == proof[:-64] and gwc != proof
    # any flipped scalar / point byte must be rejected (a decoding assert counts as a rejection)
    for pos in (5, 32 * 3 + 1, len(proof) - 40, len(proof) - 3):
        bad = bytearray(proof)
        bad[pos] ^= 4
        try:
            accepted = rp.verify_proof(pk, bytes(bad))
        except AssertionError:                           # (a decoding assert counts as a rejection)
            accepted = False
        assert not accepted, pos


@pytest.mark.parametrize("which", ["lookup-api-set", "shuffle-api", "shuffle-gates"])
def test_remaining_reference_examples_on_the_big_integer_prover(which):
    """examples/lookup_api_set.rs, shuffle_api.rs and shuffle.rs: the product-side descriptions go through the traced
    front end and its chunking passes to the shapes the twins state, the witnesses agree, and the twins' proofs are
    accepted -- and rejected once a witness value breaks the argument"""
    from h2util import arr_to_ints

    if which == "lookup-api-set":
        cs, W, k = circuits.lookup_api_set(), rp.LookupApiSet, 7
        adv, fixed, _ = circuits.lookup_api_set_synthesize(k)
        assert [len(st) for st in cs.lookups[0][2]] == [1, 2, 2, 1] and len(cs.lookups) == 1    # the example's "set 0 .. 3"
    elif which == "shuffle-api":
        cs, W, k = circuits.shuffle_api(), rp.ShuffleApi, 6
        adv, fixed, _ = circuits.shuffle_api_synthesize(k)
        assert [len(g) for g in cs.shuffles] == [1] and len(cs.shuffles[0][0][1]) == 2
    else:
        cs, W, k = circuits.shuffle_gates(), rp.shuffle_gates_class(), 6
        adv, fixed, _ = circuits.shuffle_gates_synthesize(k)
        assert not cs.lookups and not cs.shuffles and len(cs.gates) == 3
    assert cs.degree() == W.degree and cs.blinding_factors() == W.blinding_factors
    assert cs.advice_queries == W.advice_queries and cs.fixed_queries == W.fixed_queries and cs.perm_columns == W.perm_columns
    radv, rfixed, rcopies = W.synthesize(k)
    assert [arr_to_ints(c) for c in adv] == radv and [arr_to_ints(c) for c in fixed] == rfixed
    pk = rp.keygen(W, k, S_TRAPDOOR, rfixed, rcopies)
    for gwc in (False, True):
        assert rp.verify_proof(pk, rp.create_proof(pk, radv, ProverRng(3), use_gwc=gwc), use_gwc=gwc)
    bad = [c[:] for c in radv]
    if which == "lookup-api-set":
        bad[3][1] = 11                                   # 10 * 11 is not in the table 0 .. 99
        with pytest.raises(Exception):
            rp.create_proof(pk, bad, ProverRng(3))
    elif which == "shuffle-api":
        bad[2][0] = 5                                    # no longer a permutation of the inputs
        with pytest.raises(Exception):
            rp.create_proof(pk, bad, ProverRng(3))
    else:
        bad[5][3] = (bad[5][3] + 1) % rp.R              # a shuffled cell changed: the transition gate fails on that row
        assert not rp.verify_proof(pk, rp.create_proof(pk, bad, ProverRng(3)))


def _second_lookup_shuffle_witness(k):
    """the LookupShuffle witness with another public input (w[0] = instance[0][0] = 43)"""
    adv, fixed, copies, inst = rp.LookupShuffle.synthesize(k)
    adv = [c[:] for c in adv]
    adv[11][0] = 43
    return adv, [[43, 7]]


def test_big_integer_prover_with_several_circuit_instances():
    """`circuits: &[ConcreteCircuit]` (plonk/prover.rs:206-232): every phase circuit by circuit, one quotient.  One
    circuit given in the list form is the single-circuit proof; two circuits verify with their own public inputs only."""
    cs, k = rp.MiniPlonk, 4
    adv5, fixed, copies = cs.synthesize(k, a=5)
    adv7, _, _ = cs.synthesize(k, a=7)
    pk = rp.keygen(cs, k, S_TRAPDOOR, fixed, copies)
    assert rp.create_proof(pk, [adv5], ProverRng(3), instances=[()]) == rp.create_proof(pk, adv5, ProverRng(3))
    for use_gwc in (False, True):
        proof = rp.create_proof(pk, [adv5, adv7], ProverRng(3), use_gwc=use_gwc, instances=[(), ()])
        assert rp.verify_proof(pk, proof, use_gwc=use_gwc, instances=[(), ()], circuits=2)
        assert len(proof) > len(rp.create_proof(pk, adv5, ProverRng(3), use_gwc=use_gwc))
        bad = bytearray(proof)
        bad[32 * 4 + 3] ^= 1                                    # an advice commitment of the second circuit
        try:
            accepted = rp.verify_proof(pk, bytes(bad), use_gwc=use_gwc, instances=[(), ()], circuits=2)
        except AssertionError:      # (a point that no longer decodes counts as a rejection)
            accepted = False
        assert not accepted
    bad7 = [c[:] for c in adv7]
    bad7[2][0] += 1                                             # the second circuit's witness breaks its gate
    assert not rp.verify_proof(pk, rp.create_proof(pk, [adv5, bad7], ProverRng(3), instances=[(), ()]), instances=[(), ()], circuits=2)
    # lookups, shuffles and instance columns in both circuits
    cs, k = rp.LookupShuffle, 5
    adv, fixed, copies, inst = cs.synthesize(k)
    adv2, inst2 = _second_lookup_shuffle_witness(k)
    pk = rp.keygen(cs, k, S_TRAPDOOR, fixed, copies)
    proof = rp.create_proof(pk, [adv, adv2], ProverRng(8), instances=[inst, inst2])
    assert rp.verify_proof(pk, proof, instances=[inst, inst2], circuits=2)
    assert not rp.verify_proof(pk, proof, instances=[inst2, inst], circuits=2)
    assert not rp.verify_proof(pk, proof, instances=[inst, inst], circuits=2)


def test_pairing_self_checks():
    import bn254_pairing as bp

    assert bp.g2_on_curve(bp.G2) and bp.g2_mul(bp.G2, bp.R) is None and bp.g2_mul(bp.G2, 2) is not None
    x = [3] + [5] * 11
    assert bp.f12_mul(x, bp.f12_inv(x)) == bp.ONE12
    e = bp.pairing(bp.G2, rp.G1)
    assert e != bp.ONE12 and bp.f12_pow(e, bp.R) == bp.ONE12                      # non-degenerate, order r
    a, b = 0x1234567, 0x7654321
    assert bp.pairing(bp.g2_mul(bp.G2, b), rp.g1_mul(rp.G1, a)) == bp.f12_pow(e, a * b % bp.R)   # bilinear
    assert bp.pairing(bp.G2, None) == bp.ONE12


def test_big_integer_verifier_with_the_real_pairing():
    """N3: the acceptance decision through e(L, [s]G2) = e(R, G2) instead of the trapdoor, both multiopen schemes"""
    adv, fixed, copies = rp.MiniPlonk.synthesize(4)
    pk = rp.keygen(rp.MiniPlonk, 4, S_TRAPDOOR, fixed, copies)
    for use_gwc in (False, True):
        proof = rp.create_proof(pk, adv, ProverRng(21), use_gwc=use_gwc)
        assert rp.verify_proof(pk, proof, use_gwc=use_gwc, pairing=True)
        bad = bytearray(proof)
        bad[-1] ^= 0x40 if use_gwc else 0x01
        try:
            accepted = rp.verify_proof(pk, bytes(bad), use_gwc=use_gwc, pairing=True)
        except AssertionError:      # (a point that no longer decodes counts as a rejection)
            accepted = False
        assert not accepted
    wrong = [c[:] for c in adv]
    wrong[2][0] += 1
    assert not rp.verify_proof(pk, rp.create_proof(pk, wrong, ProverRng(21)), pairing=True)


def test_big_integer_verifier_rejects_bad_witness():
    adv, fixed, copies = rp.MiniPlonk.synthesize(4)
    pk = rp.keygen(rp.MiniPlonk, 4, S_TRAPDOOR, fixed, copies)
    bad = [c[:] for c in adv]
    bad[2][0] += 1                      # a * a != c on row 0
    assert not rp.verify_proof(pk, rp.create_proof(pk, bad, ProverRng(3)))
    bad = [c[:] for c in adv]
    bad[0][1], bad[2][1] = 6, 31        # row 1 still satisfies a + b = c, but a0 = a1 is broken
    assert not rp.verify_proof(pk, rp.create_proof(pk, bad, ProverRng(3)))


def test_transcript_matches_reference_restatement():
    rnd = random.Random(1)
    a, b = transcript.Blake2bWrite(), rp.Transcript()
    for step in range(40):
        kind = rnd.randrange(3)
        if kind == 0:
            assert a.squeeze_challenge_scalar() == b.squeeze()
        elif kind == 1:
            v = rnd.randrange(rp.R)
            a.write_scalar(v)
            b.write_scalar(v)
        else:
            P = rp.g1_mul(rp.G1, rnd.randrange(1, rp.R))
            a.write_point(P)
            b.write_point(P)
    assert a.finalize() == bytes(b.out)
    with pytest.raises(ValueError):
        a.write_point(None)
    # the compressed encoding round-trips through the reference decoder
    P = rp.g1_mul(rp.G1, 0xABCDEF)
    assert rp.point_from_bytes(transcript.point_to_bytes(P)) == P
    assert rp.point_from_bytes(transcript.point_to_bytes(rp.g1_neg(P))) == rp.g1_neg(P)


def test_jacobian_normalisation():
    P = rp.g1_mul(rp.G1, 777)
    z = 0x1234567890ABCDEF
    xyz = [P[0] * z * z % rp.Q, P[1] * z ** 3 % rp.Q, z]
    limbs = []
    for v in xyz:
        m = (v << 256) % rp.Q
        limbs += [(m >> (64 * i)) & (2 ** 64 - 1) for i in range(4)]
    assert transcript.jacobian_to_affine(limbs) == P
    assert transcript.jacobian_to_affine([0] * 4 + limbs[4:8] + [0] * 4) is None


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_permutation_mapping_matches_reference(seed):
    rnd = random.Random(seed)
    ncols, n = 4, 32
    copies = []
    for _ in range(rnd.randrange(1, 60)):
        copies.append(((rnd.randrange(ncols), rnd.randrange(n)), (rnd.randrange(ncols), rnd.randrange(n))))
    want = rp.permutation_mapping(ncols, n, copies)
    mc, mr = prover.permutation_mapping(ncols, n, [(l[0], l[1], r[0], r[1]) for l, r in copies])
    got = [[(int(mc[c][j]), int(mr[c][j])) for j in range(n)] for c in range(ncols)]
    assert got == want
    mc, mr = prover.permutation_mapping(2, 8, np.zeros((0, 4)))
    assert (mc == np.repeat(np.arange(2), 8).reshape(2, 8)).all() and (mr == np.tile(np.arange(8), (2, 1))).all()


def _run_program(g, parts, adv, fix):
    inter = []

    def get(v):
        if v.kind == ev.VS_CONSTANT:
            return g.constants[v.index]
        if v.kind == ev.VS_INTERMEDIATE:
            return inter[v.index]
        return (fix if v.kind == ev.VS_FIXED else adv)(v.index, g.rotations[v.rot])

    for c in g.calculations:
        a = get(c.a)
        inter.append({ev.CALC_ADD: lambda: a + get(c.b), ev.CALC_SUB: lambda: a - get(c.b), ev.CALC_MUL: lambda: a * get(c.b),
                      ev.CALC_NEGATE: lambda: -a, ev.CALC_STORE: lambda: a}[c.op]() % rp.R)
    return [get(p) for p in parts]


@pytest.mark.parametrize("make,ref", [(circuits.mini_plonk, rp.MiniPlonk), (rot_gate_cs, rp.RotGate)])
def test_constraint_system_and_gate_program(make, ref):
    cs = make()
    assert cs.advice_queries == ref.advice_queries and cs.fixed_queries == ref.fixed_queries
    assert cs.perm_columns == ref.perm_columns
    assert cs.degree() == ref.degree and cs.blinding_factors() == ref.blinding_factors
    assert (cs.num_advice, cs.num_fixed) == (ref.num_advice, ref.num_fixed)
    g, parts = hc.compile_gates(cs)
    rnd = random.Random(5)
    for _ in range(10):
        vals = {}
        adv = lambda c, r: vals.setdefault(("a", c, r), rnd.randrange(rp.R))  # noqa: E731
        fix = lambda c, r: vals.setdefault(("f", c, r), rnd.randrange(rp.R))  # noqa: E731
        assert _run_program(g, parts, adv, fix) == ref.gates(adv, fix)


def test_mini_plonk_witness_matches_reference():
    for k in (4, 6):
        adv, fixed, copies = circuits.mini_plonk_synthesize(k)
        radv, rfixed, rcopies = rp.MiniPlonk.synthesize(k)
        assert [[int(v) for v in c[:, 0]] for c in adv] == radv and all(not c[:, 1:].any() for c in adv)
        assert [[int(v) for v in c[:, 0]] for c in fixed] == rfixed
        assert sorted(map(tuple, copies.tolist())) == sorted((l[0], l[1], r[0], r[1]) for l, r in rcopies)


@pytest.mark.parametrize("k,degree", [(4, 3), (7, 4), (9, 6), (22, 3)])
def test_domain_scalars(k, degree):
    d = prover.Domain(k, degree)
    assert pow(d.omega, 1 << k, rp.R) == 1 and pow(d.omega, 1 << (k - 1), rp.R) != 1
    assert pow(d.extended_omega, 1 << d.extended_k, rp.R) == 1
    assert (1 << d.extended_k) >= (degree - 1) << k > (1 << (d.extended_k - 1))
    if k <= 9:
        r = rp.Domain(k, degree)
        assert (d.omega, d.extended_omega, d.extended_k) == (r.omega, r.extended_omega, r.extended_k)
        zn = pow(prover.ZETA, 1 << k, rp.R)
        for i, t in enumerate(d.t_evaluations):
            assert t * (zn * pow(d.extended_omega, i << k, rp.R) - 1) % rp.R == 1
    assert d.rotate_omega(5, -3) * pow(d.omega, 3, rp.R) % rp.R == 5


def test_coset_plan_and_unmix_matrix():
    """parallel.coset_plan / coset_unmix_matrix: the multi-GPU split of the extended domain"""
    from halo2_gpu_specific_amd import parallel

    assert parallel.coset_plan(2, 8, 5) == (2, [1]) and parallel.coset_plan(8, 2, 1) == (2, [1, 3, 5, 7])
    assert parallel.coset_plan(4, 4, 2) == (4, [2]) and parallel.coset_plan(4, 1, 0) == (1, [0, 1, 2, 3])
    covered = sorted(j for r in range(4) for j in parallel.coset_plan(8, 4, r)[1])
    assert covered == list(range(8))
    rnd = random.Random(5)
    for k, deg in ((4, 3), (5, 5), (3, 9)):
        d = prover.Domain(k, deg)
        c = 1 << (d.extended_k - d.k)
        gammas = [pow(prover.ZETA * pow(d.extended_omega, j, rp.R) % rp.R, d.n, rp.R) for j in range(c)]
        # the vanishing polynomial is the constant gamma_j - 1 on coset j: t_evaluations of the reference
        assert [pow(g - 1, -1, rp.R) for g in gammas] == d.t_evaluations
        h = [[rnd.randrange(rp.R) for _ in range(3)] for _ in range(c)]                 # c pieces of 3 coefficients
        P = [[sum(pow(g, m, rp.R) * h[m][t] for m in range(c)) % rp.R for t in range(3)] for g in gammas]
        M = parallel.coset_unmix_matrix(gammas, d.quotient_poly_degree)
        for m in range(d.quotient_poly_degree):
            assert [sum(M[m][j] * P[j][t] for j in range(c)) % rp.R for t in range(3)] == h[m]


def test_rng_is_a_fixed_stream():
    a, b = ProverRng(9), ProverRng.deterministic(9)
    assert [a.next_u64() for _ in range(4)] == [b.next_u64() for _ in range(4)]
    assert a.fr() == b.fr() < rp.R and a.u16() == b.u16() < 65536
    limbs = ProverRng.random_poly_limbs(a.random_poly_key(), 16)
    ints = b.random_poly(16)
    assert [transcript.fr_from_mont_limbs(r) for r in limbs] == ints and all(v < rp.R for v in ints)


def test_rng_defaults_to_os_entropy():
    """no constant default seed -- ProverRng() draws from os.urandom; the seeded stream is opt-in"""
    a, b = ProverRng(), ProverRng()
    assert a.secure and b.secure and not ProverRng(1).secure
    assert [a.next_u64() for _ in range(4)] != [b.next_u64() for _ in range(4)]
    assert a.random_poly_key() != b.random_poly_key() and len(a.random_poly_key()) == 32
    assert a.fr() < rp.R and a.u16() < 65536


def test_keyed_rng_is_shared_and_unpredictable_without_the_key():
    """the multi-rank mode of the OS-entropy rng: the same key gives the same draws, another key different ones; the
    blinding polynomial's key comes out of the same stream"""
    import os as _os

    k1, k2 = _os.urandom(32), _os.urandom(32)
    a, b, c = ProverRng(key=k1), ProverRng(key=k1), ProverRng(key=k2)
    assert a.secure and a.key == k1
    da = [a.u16() for _ in range(5)] + [a.fr() for _ in range(3)] + [a.random_poly_key()]
    db = [b.u16() for _ in range(5)] + [b.fr() for _ in range(3)] + [b.random_poly_key()]
    dc = [c.u16() for _ in range(5)] + [c.fr() for _ in range(3)] + [c.random_poly_key()]
    assert da == db != dc and all(v < rp.R for v in da[5:8]) and len(da[8]) == 32
    assert ProverRng(3).shared() is not None and ProverRng(key=k1).shared().key == k1     # already shareable: unchanged


def test_chacha20_keystream_known_answer():
    """the all-zero key / nonce / counter block of ChaCha20 (the classic test vector of the cipher, RFC 7539 A.1 #1)"""
    from halo2_gpu_specific_amd.rng import chacha20_blocks

    want = ("76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7"
            "da41597c5157488d7724e03fb8d84a376a43b8f41518a11cc387b669b2ee6586")
    blocks = chacha20_blocks(bytes(32), 2)
    assert blocks[0].tobytes().hex() == want
    # block 1 of the same key (RFC 7539 A.1 #2 uses counter 1)
    assert blocks[1].tobytes().hex().startswith("9f07e7be5551387a98ba977c732d080d")


def test_max_scalar_bits():
    col = np.zeros((8, 4), dtype=np.uint64)
    assert prover.max_scalar_bits(col) == 0
    col[3, 0] = 30
    assert prover.max_scalar_bits(col) == 5
    col[5, 2] = 1
    assert prover.max_scalar_bits(col) == 129


def test_prover_refuses_to_run_without_a_device():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="no CPU path"):
        prover.Device()


def test_chunk_lookups_like_the_reference():
    """plonk/logup.rs:203-281 (`test_chunks_normal`, `test_chunks_order`) restated on the host mirror"""
    cs = hc.ConstraintSystem()
    input_0, input_1, _l0, _l1 = (cs.advice_column() for _ in range(4))
    s_0, _s1, table_0, _t1 = (cs.fixed_column() for _ in range(4))
    cs.lookup("table1", [(cs.query_advice(input_0), cs.query_fixed(table_0))])
    cs.lookup("table2", [(cs.query_advice(input_1), cs.query_fixed(table_0))])
    cs.chunk_lookups()                                            # degree = 4
    assert cs.degree() == 4 and len(cs.lookups) == 1
    assert len(cs.lookups[0][1]) == 1 and len(cs.lookups[0][2]) == 2
    t0 = cs.query_fixed(table_0)
    cs.lookup("table3", [(cs.query_advice(input_1) * cs.query_fixed(s_0), t0), (cs.query_advice(input_1), t0)])
    cs.chunk_lookups()                                            # degree = 5
    assert cs.degree() == 5 and len(cs.lookups) == 2
    assert len(cs.lookups[0][1]) == 1 and len(cs.lookups[0][2]) == 1
    assert len(cs.lookups[1][1]) == 2 and len(cs.lookups[1][2]) == 1
    # the big-degree expression in the middle
    cs = hc.ConstraintSystem()
    input_0, input_1 = cs.advice_column(), cs.advice_column()
    s_0, table_0 = cs.fixed_column(), cs.fixed_column()
    cs.lookup("table1", [(cs.query_advice(input_0), cs.query_fixed(table_0))])
    cs.lookup("table2", [(cs.query_advice(input_1) * cs.query_fixed(s_0), cs.query_fixed(table_0))])
    cs.lookup("table3", [(cs.query_advice(input_1), cs.query_fixed(table_0))])
    cs.chunk_lookups()
    assert len(cs.lookups) == 1 and len(cs.lookups[0][1]) == 1 and len(cs.lookups[0][2]) == 2
    hc.compile_evaluator(cs)                                      # and the result feeds the evaluator


def test_chunk_shuffles_first_fit():
    cs = hc.ConstraintSystem()
    cols = [cs.advice_column() for _ in range(8)]
    q = [cs.query_advice(c) for c in cols]
    s0 = cs.query_fixed(cs.fixed_column())
    cs.shuffle("a", [(q[0], q[1])])
    cs.shuffle("b", [(q[2] * s0, q[3])])          # degree 2
    cs.shuffle("c", [(q[4], q[5])])
    cs.shuffle("d", [(q[6], q[7])])
    cs.chunk_shuffles()                            # cs degree 4 -> at most summed degree 2 per group
    assert cs.degree() == 4
    assert [[u[0] for u in g] for g in cs.shuffles] == [["a", "c"], ["b"], ["d"]]
    cs.set_minimum_degree(6)
    cs.chunk_shuffles()
    assert [[u[0] for u in g] for g in cs.shuffles] == [["a", "b", "c"], ["d"]]


def _golden_case(case):
    cs = {"mini-plonk": rp.MiniPlonk, "rot-gate": rp.RotGate, "lookup-shuffle": rp.LookupShuffle}[case["circuit"]]
    syn = cs.synthesize(case["k"])
    inst = [[int(v, 16) for v in col] for col in case["instances"]]
    return cs, syn[0], syn[1], syn[2], inst


def test_big_integer_prover_reproduces_committed_proofs():
    """tests/golden/proof_kat.json (gen_proof_golden.py): the big-integer prover's conventions are pinned byte for byte"""
    from h2util import load_golden

    for case in load_golden("proof_kat.json"):
        cs, adv, fixed, copies, inst = _golden_case(case)
        pk = rp.keygen(cs, case["k"], int(case["trapdoor"], 16), fixed, copies)
        assert pk.transcript_repr == int(case["vk_digest"], 16)
        gwc = case["scheme"] == "gwc"
        proof = rp.create_proof(pk, adv, ProverRng(case["seed"]), use_gwc=gwc, instances=inst)
        assert proof.hex() == case["proof"], (case["circuit"], case["scheme"])
        assert rp.verify_proof(pk, bytes.fromhex(case["proof"]), use_gwc=gwc, instances=inst)


def test_code_object_cache_directory_must_be_private(tmp_path, monkeypatch):
    """a cache directory that others can write to is not used (a planted code object would be loaded into the
    prover by hipModuleLoadData); the library makes its own, mode 0700"""
    import os

    from halo2_gpu_specific_amd import prover

    b = prover.program_descriptor(circuits.mini_plonk(), 5, 7)
    good = tmp_path / "mine"
    monkeypatch.setenv("H2_JIT_CACHE", str(good))
    assert ev.compile_only(b)["from_cache"] == 0
    assert (os.stat(good).st_mode & 0o777) == 0o700 and len(os.listdir(good)) == 1
    assert ev.compile_only(b)["from_cache"] == 2
    shared = tmp_path / "shared"
    shared.mkdir()
    os.chmod(shared, 0o777)
    monkeypatch.setenv("H2_JIT_CACHE", str(shared))
    assert ev.compile_only(b)["from_cache"] == 0         # not found: the shared directory is neither read nor written
    assert os.listdir(shared) == []


def test_code_object_cache_survives_damage_and_concurrent_writers(tmp_path, monkeypatch):
    """the disk cache of generated code objects: a truncated or bit-flipped file is a miss (SHA-256 trailer) and is replaced by
    the rebuild; h2_evalh_compile called from several threads at once -- for one program and for different ones -- leaves
    one intact file per program (every writer has a temporary file of its own, renamed into place)"""
    import os
    import threading

    from halo2_gpu_specific_amd import prover

    cache = tmp_path / "c"
    monkeypatch.setenv("H2_JIT_CACHE", str(cache))
    b = prover.program_descriptor(circuits.mini_plonk(), 5, 7)
    first = ev.compile_only(b)
    assert first["from_cache"] == 0
    (name,) = os.listdir(cache)
    path = cache / name
    whole = path.read_bytes()
    for damaged in (whole[:len(whole) // 2], whole[:-1], whole[:100] + bytes([whole[100] ^ 1]) + whole[101:], whole + b"x"):
        path.write_bytes(damaged)
        again = ev.compile_only(b)
        assert again["from_cache"] == 0 and again["products_per_row"] == first["products_per_row"]
        assert path.read_bytes() == whole                    # rebuilt: the same bytes (the generator is deterministic)
        assert ev.compile_only(b)["from_cache"] == 2
    # concurrent writers
    for f in os.listdir(cache):
        os.unlink(cache / f)
    descs = [b, prover.program_descriptor(circuits.mini_plonk(), 6, 8), prover.program_descriptor(circuits.wide(4), 6, 8)]
    results, errors = [], []

    def work(i):
        try:
            results.append((i % len(descs), ev.compile_only(descs[i % len(descs)])["products_per_row"]))
        except Exception as e:                               # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(9)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors and len(results) == 9
    assert len({r for r in results}) == len({i for i, _ in results})       # one answer per program
    files = sorted(os.listdir(cache))
    assert len(files) == 2 and not [f for f in files if ".tmp." in f]    # (k is not part of a program: two distinct ones)
    for d in descs:
        assert ev.compile_only(d)["from_cache"] == 2


def test_missing_hiprtc_is_an_error_message_not_a_crash(tmp_path):
    """a machine without libhiprtc.so (H2_HIPRTC_LIB names a file that is not there; own process: the library loads hipRTC
    once): h2_evalh_compile returns a status with the loader's message -- the error string used to be built from a second
    dlerror() call, which returns NULL -- and a program already in the disk cache still loads without hipRTC"""
    import subprocess
    import sys

    code = (
        "import sys; sys.path[:0] = %r\n"
        "import halo2_gpu_specific_amd as h2\n"
        "from halo2_gpu_specific_amd import circuits, prover, evaluation as ev\n"
        "b = prover.program_descriptor(circuits.mini_plonk(), 5, 7)\n"
        "try:\n"
        "    r = ev.compile_only(b); print('COMPILED', r['from_cache'])\n"
        "except Exception as e:\n"
        "    print('ERROR', e)\n"
    ) % [ROOT, os.path.join(ROOT, "tests")]
    env = dict(os.environ, H2_JIT_CACHE=str(tmp_path / "c"), H2_HIPRTC_LIB=str(tmp_path / "no_such_libhiprtc.so"))
    out = subprocess.run([sys.executable, "-c", code], text=True, env=env, capture_output=True)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "ERROR" in out.stdout and "libhiprtc.so could not be loaded" in out.stdout and "no_such_libhiprtc.so" in out.stdout, out.stdout
    env.pop("H2_HIPRTC_LIB")
    out = subprocess.run([sys.executable, "-c", code], text=True, env=env, capture_output=True)
    assert "COMPILED 0" in out.stdout, out.stdout + out.stderr[-2000:]
    env["H2_HIPRTC_LIB"] = str(tmp_path / "no_such_libhiprtc.so")
    out = subprocess.run([sys.executable, "-c", code], text=True, env=env, capture_output=True)
    assert "COMPILED 2" in out.stdout, out.stdout + out.stderr[-2000:]      # from the disk cache: hipRTC not needed


def test_per_process_cache_directory_goes_with_the_process(tmp_path):
    """a cache directory that is not private to the user is neither read nor written: the process makes one of its own under
    TMPDIR -- and removes it, with its files, when it exits (it used to stay behind, one per process)"""
    import subprocess
    import sys

    shared = tmp_path / "shared"
    shared.mkdir()
    os.chmod(shared, 0o777)
    code = (
        "import sys; sys.path[:0] = %r\n"
        "from halo2_gpu_specific_amd import circuits, prover, evaluation as ev\n"
        "print('FROM', ev.compile_only(prover.program_descriptor(circuits.mini_plonk(), 5, 7))['from_cache'])\n"
    ) % [ROOT, os.path.join(ROOT, "tests")]
    out = subprocess.run([sys.executable, "-c", code], text=True, capture_output=True,
                         env=dict(os.environ, H2_JIT_CACHE=str(shared), TMPDIR=str(tmp_path)))
    assert "FROM 0" in out.stdout and "per-process directory" in out.stderr, out.stdout + out.stderr[-1000:]
    assert os.listdir(shared) == [] and sorted(os.listdir(tmp_path)) == ["shared"]


def test_code_objects_of_another_compiler_version_are_rebuilt(tmp_path, monkeypatch):
    """the cache file's header names the hipRTC version that made its code objects: an intact file (valid SHA-256 trailer) from
    ANOTHER version is a miss for a process that can compile -- after a ROCm upgrade the objects are rebuilt instead of being
    loaded for ever -- and the rebuild puts the current version back"""
    import hashlib

    from halo2_gpu_specific_amd import prover

    cache = tmp_path / "c"
    monkeypatch.setenv("H2_JIT_CACHE", str(cache))
    b = prover.program_descriptor(circuits.mini_plonk(), 5, 7)
    assert ev.compile_only(b)["from_cache"] == 0
    (name,) = os.listdir(cache)
    whole = (cache / name).read_bytes()
    assert whole[:6] == b"H2EVG3" and whole[6:8] != b"\0\0"
    body = bytearray(whole[:-32])
    body[6] ^= 0x01                                              # "made by hipRTC of another major version"
    (cache / name).write_bytes(bytes(body) + hashlib.sha256(bytes(body)).digest())
    assert ev.compile_only(b)["from_cache"] == 0               # not trusted: rebuilt
    assert (cache / name).read_bytes() == whole and ev.compile_only(b)["from_cache"] == 2


def test_generated_evaluate_h_compiles_for_gfx950(tmp_path, monkeypatch):
    """csrc/evalh_gen.cpp behind the C ABI, no GPU needed: the straight-line HIP generated from a circuit's program (all three
    test circuits, with their permutation / lookup / shuffle terms) goes through hipRTC for gfx950 without spills; one store
    per stage; grouping by factor never spends more products than the reference's fold order; many-stage forms accumulate"""
    from halo2_gpu_specific_amd import prover

    monkeypatch.setenv("H2_JIT_CACHE", str(tmp_path))
    for make in (circuits.mini_plonk, rot_gate_cs, lookup_shuffle_cs, lambda: circuits.wide(4)):
        b = prover.program_descriptor(make(), 6, 8)
        src = ev.generated_source(b)
        assert src.count("fp_store(a.values") == 1 and "h2_evalh_gen" in src and "fp_load(a.values" not in src
        info = ev.compile_only(b)
        assert info["stages"] == 1 and info["scratch_bytes"] == 0 and 0 < info["max_registers"] <= 256
        # cost in products' worth: a pair under one reduction (fp_mul2) is a product and a half; never more than the formulas
        # as written spend
        cost = lambda i: i["products_per_row"] - 0.5 * i["fused_pairs_per_row"]  # noqa: E731
        assert cost(info) <= info["reference_products_per_row"]
        monkeypatch.setenv("H2_JIT_FACTOR", "0")
        plain = ev.compile_only(b)
        assert plain["products_per_row"] >= info["products_per_row"] - 1 and plain["terms"] == info["terms"]
        monkeypatch.delenv("H2_JIT_FACTOR")
        monkeypatch.setenv("H2_JIT_STAGE_PRODUCTS", "6")
        staged = ev.compile_only(b)
        if info["products_per_row"] > 12:
            assert staged["stages"] > 1
            assert "fp_add(fp_load(a.values + idx)" in ev.generated_source(b, 1)
        monkeypatch.delenv("H2_JIT_STAGE_PRODUCTS")
    # the wide circuit: 16 gates q (a b c - d) share their selector, the lookups' terms share l_0 / l_last / l_active_row
    wide = ev.compile_only(prover.program_descriptor(circuits.wide(16), 6, 8))
    assert wide["products_per_row"] <= 0.75 * wide["reference_products_per_row"]
    # a gate set with a constant per gate and 30 columns read at three rotations in no order (tools/evalh_bench.py's shape).
    # With everything a kernel argument read field by field that compiled to 464 registers and scratch; the constants are module
    # data, the loaded values stay within the live budget, a build that still exceeds the register file is rebuilt with the
    # argument block read through LDS, and the selectors come out of the nested products: ONE stage, two waves per SIMD, fewer
    # products than written
    rnd = random.Random(1)
    big = hc.ConstraintSystem("big")
    adv = [big.advice_column() for _ in range(30)]
    fix = [big.fixed_column() for _ in range(8)]
    for g in range(40):
        q = big.query_fixed(rnd.choice(fix))
        a = [big.query_advice(rnd.choice(adv), rnd.choice([0, 0, 0, 1, -1])) for _ in range(5)]
        big.create_gate("g%d" % g, [q * (a[0] * a[1] + a[2] - a[3]) * (a[4] + rnd.randrange(1, 100))])
    bb = prover.program_descriptor(big, 6, 8)
    info = ev.compile_only(bb)
    assert info["stages"] == 1 and info["scratch_bytes"] == 0 and info["max_registers"] <= 256
    assert info["products_per_row"] < 0.9 * info["reference_products_per_row"]
    big_src = ev.generated_source(bb)
    assert "h2_consts[" in big_src and big_src.count("fp_load(h2_consts + ") >= 30 and "h2_consts" not in src        # a constant per gate: module data, not kernel arguments
    monkeypatch.setenv("H2_JIT_LDS_ARGS", "1")
    assert "sh_cols" in ev.generated_source(bb) and "sh_cols" not in big_src
    monkeypatch.delenv("H2_JIT_LDS_ARGS")
    with pytest.raises(Exception):
        ev.generated_source(b, 99)
    # a malformed program is refused with an error, not compiled into a kernel that reads outside its tables
    import copy as _copy

    for breakage in ("intermediate", "constant", "column", "rotation", "perm"):
        bad = prover.program_descriptor(circuits.mini_plonk(), 6, 8)
        d = bad.desc
        calcs = ctypes.cast(d.calculations, ctypes.POINTER(ev.Calculation))
        if breakage == "intermediate":
            calcs[0].a = ev.vs(ev.VS_INTERMEDIATE, 5)          # used before it is defined
        elif breakage == "constant":
            calcs[0].a = ev.vs(ev.VS_CONSTANT, 10 ** 6)
        elif breakage == "column":
            calcs[0].a = ev.vs(ev.VS_ADVICE, d.n_advice)
        elif breakage == "rotation":
            calcs[0].a = ev.vs(ev.VS_ADVICE, 0, d.n_rotations)
        else:
            ctypes.cast(d.perm_col_index, ctypes.POINTER(ctypes.c_uint32))[0] = d.n_advice + 7
        with pytest.raises(Exception, match="out of range|before it is defined"):
            ev.compile_only(bad)


# ---- CircuitData (plonk.rs:126-204, helpers.rs write_cs / read_cs) ------------------------------------------------
def _u32s(*vals):
    import struct

    return b"".join(struct.pack("<I", v & 0xFFFFFFFF) for v in vals)


def test_constraint_system_bytes_follow_write_cs():
    """the mini-PLONK constraint system, byte for byte as helpers.rs:406-456 / :687-757 lay it out (written out by hand
    from those functions: every integer a little-endian u32, rotation as i32, Any = {Advice 0, Fixed 1, Instance 2},
    expression codes Constant 0 .. Scaled 7)"""
    from halo2_gpu_specific_amd import formats

    cs = hc.ConstraintSystem("tiny")
    a, b = cs.advice_column(), cs.advice_column()
    s = cs.fixed_column()
    cs.enable_equality(a)
    cs.enable_equality(b)
    qa, qb1, qs = cs.query_advice(a), cs.query_advice(b, -1), cs.query_fixed(s)
    cs.create_gate("g", [qs * (qa + qb1 * 5)])
    want = _u32s(2, 0, 0, 1)                        # advice, instance, selectors, fixed
    want += _u32s(2, 1, 2)                          # num_advice_queries: a once, b twice (cur + prev)
    want += _u32s(0) + _u32s(0)                     # selector_map, constants
    want += _u32s(3, 0, 0, 1, 0, 1, -1)             # advice queries (a,0) (b,0) (b,-1)
    want += _u32s(0)                                # instance queries
    want += _u32s(1, 0, 0)                          # fixed queries (s,0)
    want += _u32s(2, 0, 0, 1, 0)                    # permutation columns: (index, Any::Advice) x 2
    want += _u32s(0) + _u32s(0) + _u32s(0) + _u32s(0)   # lookups, shuffles, range checks, named advices
    want += _u32s(1)                                # gates
    want += _u32s(1)                                # one polynomial: Product(Fixed, Sum(Advice, Scaled(Advice, 5)))
    want += _u32s(6) + _u32s(1, 0, 0, 0)            # Product, Fixed{query 0, column 0, rot 0}
    want += _u32s(5) + _u32s(2, 0, 0, 0)            # Sum, Advice{query 0, column 0, rot 0}
    want += _u32s(7) + _u32s(2, 2, 1, -1) + (5).to_bytes(32, "little")   # Scaled(Advice{query 2, column 1, rot -1}, 5)
    want += _u32s(3) + _u32s(0, 1, 0) + _u32s(0, 0, 0) + _u32s(1, 0, -1)  # queried cells: (column, Any, rotation)
    assert formats.cs_store(cs) == want


def test_verifying_key_preimage_is_derived_twice():
    """The constraint-system bytes hashed into the verifying-key digest, derived independently on both sides: the product
    serialises its ConstraintSystem (formats.cs_store), the big-integer prover traces the value closures of its own
    circuit classes into expression trees and serialises those (ref_plonk.write_cs, written from helpers.rs:406-456 /
    :687-757) -- byte-equal for every circuit the proof tests use, and the hand-written layout of the test above pins
    one of them to the reference's field order"""
    import product_circuits as pc
    from halo2_gpu_specific_amd import formats

    pairs = [(pc.mini_plonk_cs(), rp.MiniPlonk), (pc.rot_gate_cs(), rp.RotGate), (pc.lookup_shuffle_cs(), rp.LookupShuffle),
             (pc.lookup_api_cs(), rp.LookupApi), (pc.shuffle_api_group_cs(), rp.ShuffleApiGroup),
             (pc.wide_cs(2), rp.wide_class(2)), (pc.wide_cs(16), rp.wide_class(16)),
             (pc.range_check_cs(0, 0xFFFF, 2), rp.range_check_class(0, 0xFFFF, 2)),
             (pc.range_check_cs(3, 40, 1), rp.range_check_class(3, 40, 1)),
             (pc.lookup_api_set_cs(), rp.LookupApiSet), (pc.shuffle_api_cs(), rp.ShuffleApi),
             (pc.shuffle_gates_cs(), rp.shuffle_gates_class()), (pc.shuffle_gates_cs(3, 7, 9), rp.shuffle_gates_class(3, 7, 9))]
    for cs, ref in pairs:
        assert formats.cs_store(cs) == rp.write_cs(ref), ref.name
    # the tracer itself: the tree shapes of plonk/circuit.rs's operator overloading
    a, f = rp._Sym(("advice", 0, 0)), rp._Sym(("fixed", 0, 0))
    assert (a - f).node == ("sum", ("advice", 0, 0), ("neg", ("fixed", 0, 0)))
    assert (a * 3 % rp.R).node == ("scaled", ("advice", 0, 0), 3) and (a * (-1)).node[2] == rp.R - 1
    assert (7 - a).node == ("sum", ("const", 7), ("neg", ("advice", 0, 0)))
    assert (a * f + 1).node == ("sum", ("prod", ("advice", 0, 0), ("fixed", 0, 0)), ("const", 1))


@pytest.mark.parametrize("make", [circuits.mini_plonk, rot_gate_cs, lookup_shuffle_cs, circuits.range_check, lambda: circuits.wide(2)])
def test_constraint_system_round_trip(make):
    from halo2_gpu_specific_amd import formats

    cs = make()
    raw = formats.cs_store(cs)
    r = formats._Reader(raw)
    back = formats.cs_fetch(r, cs.name)
    assert r.pos == len(raw)
    back.set_minimum_degree(cs.degree())
    assert formats.cs_store(back) == raw
    assert (back.degree(), back.blinding_factors(), back.perm_columns) == (cs.degree(), cs.blinding_factors(), cs.perm_columns)
    assert back.range_checks == cs.range_checks
    # the prover sees the same program: Evaluator::new on the fetched system
    g0, parts0, lk0, sh0 = hc.compile_evaluator(cs)
    g1, parts1, lk1, sh1 = hc.compile_evaluator(back)
    key = lambda g: ([(c.op, c.a.kind, c.a.index, c.a.rot, c.b.kind, c.b.index, c.b.rot) for c in g.calculations],  # noqa: E731
                     g.constants, g.rotations)
    assert key(g0) == key(g1) and len(parts0) == len(parts1) and len(lk0) == len(lk1) and len(sh0) == len(sh1)
    with pytest.raises(IOError):
        formats.cs_fetch(formats._Reader(raw[:-5]))


def test_circuit_data_file_round_trip(tmp_path):
    """CircuitData::write / read around a key whose device tensors are stood in by arrays (no GPU here): layout of
    the commitments, the raw fixed columns and the (u32, u32) mapping pairs"""
    from halo2_gpu_specific_amd import formats

    k, n = 4, 16
    cs = rot_gate_cs()
    rng = random.Random(9)

    class FakeDevice:
        def download(self, t):
            return t

    class PK:
        pass

    pk = PK()
    pk.cs = cs
    pk.fixed_values = [np.array([[rng.getrandbits(64) for _ in range(4)] for _ in range(n)], dtype=np.uint64) for _ in range(2)]
    pk.fixed_commitments = [rp.g1_mul(rp.G1, 5), None]
    pk.perm_commitments = [rp.g1_mul(rp.G1, 7 + i) for i in range(4)]
    copies = [(0, 1, 2, 3), (1, 0, 3, 5), (2, 3, 0, 7)]
    pk.mapping = prover.permutation_mapping(4, n, copies)

    class P:
        pass

    params = P()
    params.k, params.n = k, n
    path = str(tmp_path / "circuit.data")
    formats.circuit_data_write(path, FakeDevice(), params, pk)
    info = formats.circuit_data_read(path, "rot-gate")
    assert (info["j"], info["k"]) == (cs.degree(), k)
    assert info["fixed_commitments"] == [transcript.point_to_bytes(P_) for P_ in pk.fixed_commitments]
    assert info["perm_commitments"] == [transcript.point_to_bytes(P_) for P_ in pk.perm_commitments]
    assert all((x == y).all() for x, y in zip(info["fixed"], pk.fixed_values))
    assert all((x == y).all() for x, y in zip(info["mapping"][0], pk.mapping[0]))
    assert all((x == y).all() for x, y in zip(info["mapping"][1], pk.mapping[1]))
    raw = open(path, "rb").read()
    # the tail is the mapping: 4 columns x 16 pairs x 8 bytes after the u32 count and the 4 u32 lengths
    tail = raw[-(4 * n * 8):]
    assert raw[-(4 * n * 8) - 20:-(4 * n * 8)] == _u32s(4, n, n, n, n)
    assert tail[8 * 1:8 * 2] == _u32s(int(pk.mapping[0][0][1]), int(pk.mapping[1][0][1]))
    with pytest.raises(IOError):
        open(path, "wb").write(raw[:-3])
        formats.circuit_data_read(path)


def test_range_check_argument_front_end_and_witness_completion():
    """`advice_column_range` (plonk/circuit.rs:1769-1826) and the witness completion of create_proof
    (plonk/prover.rs:1699-1783, `sort` :164-200) on the host: the constraint system of examples/range-check.rs, the
    RangeCheckRelAssigner sequence, the planted range and the sorted companion against the big-integer twin, and the
    twin's prover / verifier accept the completed witness and reject a value outside the range"""
    import ref_plonk as rp
    from halo2_gpu_specific_amd import prover
    from halo2_gpu_specific_amd.rng import ProverRng

    cs = circuits.range_check()
    assert (cs.degree(), cs.blinding_factors(), cs.num_advice, cs.num_fixed) == (4, 5, 2, 3)
    assert cs.advice_queries == [(1, 0), (1, 1), (0, 0)] and cs.fixed_queries == [(0, 0), (2, 0), (1, 0)]
    assert cs.perm_columns == [] and len(cs.shuffles) == 1 and cs.range_checks == [(0, 1, 0, 0xFFFF, 2)]
    seq = prover.range_check_assigner(0, 0xFFFF, 2)
    assert len(seq) == 32769 and seq[:2] == [0, 2] and seq[-2:] == [65534, 65535]
    assert prover.range_check_assigner(3, 10, 4) == [3, 7, 10] and prover.range_check_assigner(5, 5, 1) == [5]
    k, vmax, step = 7, 30, 2
    small = circuits.range_check(0, vmax, step)
    adv, fixed, _ = circuits.range_check_synthesize(k, vmin=0, vmax=vmax, count=60)
    W = rp.range_check_class(0, vmax, step)
    want = W.complete(k, [int(v) for v in adv[0][:, 0]])
    prover.complete_range_check_witness(small, 1 << k, adv)
    assert [int(v) for v in adv[0][:, 0]] == want[0] and [int(v) for v in adv[1][:, 0]] == want[1]
    rpk = rp.keygen(W, k, S_TRAPDOOR, [[int(v) for v in f[:, 0]] for f in fixed], [])
    assert rp.verify_proof(rpk, rp.create_proof(rpk, want, ProverRng(1)))
    bad = [c.copy() for c in adv]
    bad[0][2, 0] = vmax + 1
    with pytest.raises(ValueError):
        prover.complete_range_check_witness(small, 1 << k, bad)
    with pytest.raises(ValueError):                         # the range does not fit the unused cells
        prover.complete_range_check_witness(small, 1 << k, [c.copy() for c in adv], first_unassigned={0: 120})
    # the reference's assertion (prover.rs:1731): first_unassigned <= lo - 1 with lo the lowest planted row
    usable = (1 << k) - 6
    lo = usable - len(prover.range_check_assigner(0, vmax, step))
    fresh = lambda: circuits.range_check_synthesize(k, vmin=0, vmax=vmax, count=60)[0]      # noqa: E731
    prover.complete_range_check_witness(small, 1 << k, fresh(), first_unassigned={0: lo - 1})
    with pytest.raises(ValueError):
        prover.complete_range_check_witness(small, 1 << k, fresh(), first_unassigned={0: lo})
    # without assignment tracking: completing twice is idempotent, a witness that uses a cell of the planted range is refused
    again = [c.copy() for c in adv]
    prover.complete_range_check_witness(small, 1 << k, again)
    assert all(np.array_equal(a, b) for a, b in zip(again, adv))
    for row in (lo - 1, lo, usable - 1):
        used = fresh()
        used[0][row, 0] = 7
        with pytest.raises(ValueError, match="already uses"):
            prover.complete_range_check_witness(small, 1 << k, used)


def test_constraint_system_round_trip_random_circuits():
    """write_cs / read_cs (helpers.rs:458-757) over the random circuits of tools/prover_fuzz.py: every expression code
    (constants, scalings, negations, sums, products, queries at rotations -2 .. 2 of all three column kinds), multi-set
    lookups, shuffles: the fetched system stores to the same bytes and compiles to the same evaluator program"""
    import os
    import sys

    from h2util import ROOT
    from halo2_gpu_specific_amd import formats

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import prover_fuzz

    key = lambda g: ([(c.op, c.a.kind, c.a.index, c.a.rot, c.b.kind, c.b.index, c.b.rot) for c in g.calculations],  # noqa: E731
                     g.constants, g.rotations)
    for seed in range(300, 340):
        cs = prover_fuzz.random_case(seed, satisfiable=seed % 2 == 0, k=6)[0]
        raw = formats.cs_store(cs)
        r = formats._Reader(raw)
        back = formats.cs_fetch(r, cs.name)
        assert r.pos == len(raw)
        back.set_minimum_degree(cs.degree())
        assert formats.cs_store(back) == raw, seed
        g0, parts0, lk0, sh0 = hc.compile_evaluator(cs)
        g1, parts1, lk1, sh1 = hc.compile_evaluator(back)
        assert key(g0) == key(g1) and len(parts0) == len(parts1) and len(lk0) == len(lk1) and len(sh0) == len(sh1), seed
