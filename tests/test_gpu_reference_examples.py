"""The reference's live end-to-end examples on the device prover.

examples/range-check.rs -- the range-check argument (`advice_column_range`, plonk/circuit.rs:1769-1826;
plonk/range_check.rs; witness completion in create_proof, plonk/prover.rs:1699-1783): the circuit proves to the bytes of
its big-integer twin at small sizes, at the example's own size (k = 18, 0 ..= 65535, step 2) the proof is accepted by the
twin's verifier, and a CircuitData file carrying the argument rebuilds the same key.

examples/lookup_api.rs and examples/shuffle_api_group.rs -- the traced `lookup` / `lookup_any` / `shuffle` front end with
the chunking passes (plonk/logup.rs:73-153, plonk/shuffle.rs:57-103): bytes equal to the twins at k = 6 / 7, accepted at
the examples' k = 10.

examples/lookup_api_set.rs (six lookups into one table, packed into four input sets), examples/shuffle_api.rs (one
shuffle of two expression pairs under selectors) and examples/shuffle.rs (a shuffle written as three gates over a
running-product advice column, constants theta / beta in the gate): bytes equal to the twins at k = 6 / 7, accepted at the
examples' own k (10, 10 and 8)."""
import numpy as np
import pytest

import ref_plonk as rp
from test_gpu_plonk import srs
from test_plonk_host import S_TRAPDOOR

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def device():
    from halo2_gpu_specific_amd import prover

    return prover.Device()


@pytest.mark.parametrize("k,vmax,step,count", [(7, 30, 2, 60), (8, 61, 4, 150), (9, 100, 1, 300)])
def test_range_check_proof_bytes_match_big_integer_prover(oracle, device, k, vmax, step, count):
    from halo2_gpu_specific_amd import circuits, prover
    from halo2_gpu_specific_amd.rng import ProverRng

    cs = circuits.range_check(0, vmax, step)
    W = rp.range_check_class(0, vmax, step)
    adv, fixed, copies = circuits.range_check_synthesize(k, vmin=0, vmax=vmax, count=count)
    radv = W.complete(k, [int(v) for v in adv[0][:, 0]])
    rfixed = [[int(v) for v in f[:, 0]] for f in fixed]
    params = srs(oracle, device, k)
    pk = prover.keygen(device, params, cs, fixed, copies)
    rpk = rp.keygen(W, k, S_TRAPDOOR, rfixed, [])
    assert pk.transcript_repr == rpk.transcript_repr and pk.fixed_commitments == rpk.fixed_commitments
    for seed, use_gwc in ((1, False), (2, True)):
        mine = [c.copy() for c in adv]
        proof = prover.create_proof_ext(device, params, pk, mine, ProverRng(seed), use_gwc)
        # the prover completed the witness in place as the reference does: planted range, sorted companion
        assert [int(v) for v in mine[0][:(1 << k) - 6, 0]] == radv[0][:(1 << k) - 6]
        assert [int(v) for v in mine[1][:(1 << k) - 6, 0]] == radv[1][:(1 << k) - 6]
        want = rp.create_proof(rpk, radv, ProverRng(seed), use_gwc=use_gwc)
        first = next((i for i in range(min(len(proof), len(want))) if proof[i] != want[i]), None)
        assert len(proof) == len(want) and first is None, "differs from the big-integer prover at byte %s" % first
        assert rp.verify_proof(rpk, proof, use_gwc=use_gwc)
    bad = [c.copy() for c in adv]
    bad[0][3, 0] = np.uint64(vmax + 9)
    with pytest.raises(ValueError):                       # the reference's `sort` panics on a value outside the range
        prover.create_proof_ext(device, params, pk, bad, ProverRng(1), False)
    # ... and a proof over a forged companion column (the value smuggled past the completion step) is rejected
    forged = W.complete(k, [int(v) for v in bad[0][:, 0]])
    assert not rp.verify_proof(rpk, rp.create_proof(rpk, forged, ProverRng(1)))


def test_range_check_example_size(oracle, device):
    """examples/range-check.rs: k = 18, one column range-checked into 0 ..= 65535 with step 2, 65535 random values"""
    from halo2_gpu_specific_amd import circuits, prover
    from halo2_gpu_specific_amd.rng import ProverRng

    k = 18
    cs = circuits.range_check()
    adv, fixed, copies = circuits.range_check_synthesize(k)
    params = srs(oracle, device, k)
    pk = prover.keygen(device, params, cs, fixed, copies)
    proof = prover.create_proof_with_shplonk(device, params, pk, adv, ProverRng(7))
    usable = (1 << k) - 6
    assert np.array_equal(np.sort(adv[0][:usable, 0]), adv[1][:usable, 0]) and int(adv[1][usable - 1, 0]) == 0xFFFF
    vk = rp.Keys()
    vk.cs, vk.dom, vk.s = rp.range_check_class(0, 0xFFFF, 2), rp.Domain(k, cs.degree()), S_TRAPDOOR
    vk.fixed_commitments, vk.perm_commitments, vk.transcript_repr = pk.fixed_commitments, pk.perm_commitments, pk.transcript_repr
    assert rp.verify_proof(vk, proof)
    flipped = bytearray(proof)
    flipped[40] ^= 1
    try:
        accepted = rp.verify_proof(vk, bytes(flipped))
    except AssertionError:                                               # a flipped byte may also stop being a curve point
        accepted = False
    assert not accepted


def test_circuit_data_with_a_range_check_argument(oracle, device, tmp_path):
    """CircuitData::{write, read} (plonk.rs:126-204; write_cs / read_cs carry the range-check relations,
    helpers.rs:444-451, 520-536): a key rebuilt from the file knows the argument and proves to the same bytes"""
    from halo2_gpu_specific_amd import circuits, formats, prover
    from halo2_gpu_specific_amd.rng import ProverRng

    k = 8
    cs = circuits.range_check(0, 61, 4)
    adv, fixed, copies = circuits.range_check_synthesize(k, vmin=0, vmax=61, count=150)
    params = srs(oracle, device, k)
    pk = prover.keygen(device, params, cs, fixed, copies)
    want = prover.create_proof_with_shplonk(device, params, pk, [c.copy() for c in adv], ProverRng(3))
    path = tmp_path / "range.circuit.data"
    formats.circuit_data_write(path, device, params, pk)
    info = formats.circuit_data_read(path)
    assert info["cs"].range_checks == [(0, 1, 0, 61, 4)]
    pk2 = prover.keygen_from_info(device, params, info)
    assert prover.create_proof_with_shplonk(device, params, pk2, [c.copy() for c in adv], ProverRng(3)) == want


@pytest.mark.parametrize("which,k", [("lookup", 6), ("shuffle", 7), ("lookup", 10), ("shuffle", 10)])
def test_lookup_api_and_shuffle_api_group_examples(oracle, device, which, k):
    from halo2_gpu_specific_amd import circuits, prover
    from halo2_gpu_specific_amd.rng import ProverRng

    W, make, syn = ((rp.LookupApi, circuits.lookup_api, circuits.lookup_api_synthesize) if which == "lookup" else
                    (rp.ShuffleApiGroup, circuits.shuffle_api_group, circuits.shuffle_api_group_synthesize))
    cs = make()
    assert cs.degree() == W.degree and cs.advice_queries == W.advice_queries and cs.fixed_queries == W.fixed_queries
    adv, fixed, copies = syn(k)
    params = srs(oracle, device, k)
    pk = prover.keygen(device, params, cs, fixed, copies)
    if k <= 7:
        radv, rfixed, rcopies = W.synthesize(k)
        rpk = rp.keygen(W, k, S_TRAPDOOR, rfixed, rcopies)
        assert pk.transcript_repr == rpk.transcript_repr
        for seed, use_gwc in ((1, False), (2, True)):
            proof = prover.create_proof_ext(device, params, pk, adv, ProverRng(seed), use_gwc)
            assert proof == rp.create_proof(rpk, radv, ProverRng(seed), use_gwc=use_gwc)
            assert rp.verify_proof(rpk, proof, use_gwc=use_gwc)
    else:
        vk = rp.Keys()
        vk.cs, vk.dom, vk.s = W, rp.Domain(k, cs.degree()), S_TRAPDOOR
        vk.fixed_commitments, vk.perm_commitments, vk.transcript_repr = pk.fixed_commitments, pk.perm_commitments, pk.transcript_repr
        assert rp.verify_proof(vk, prover.create_proof_with_shplonk(device, params, pk, adv, ProverRng(5)))
    if which == "shuffle":          # not a permutation: the product does not close (the reference's prover panics / MockProver fails)
        bad, _, _ = circuits.shuffle_api_group_synthesize(k, input1=(4, 1, 1, 3))
        with pytest.raises(ValueError):
            prover.create_proof_with_shplonk(device, params, pk, bad, ProverRng(1))
    else:                           # an input value missing from the table
        bad = [c.copy() for c in adv]
        bad[2][1, 0] = 77
        with pytest.raises(Exception):
            prover.create_proof_with_shplonk(device, params, pk, bad, ProverRng(1))


@pytest.mark.parametrize("which,k", [("lookup-api-set", 7), ("shuffle-api", 6), ("shuffle-gates", 6),
                                     ("lookup-api-set", 10), ("shuffle-api", 10), ("shuffle-gates", 8)])
def test_lookup_api_set_shuffle_api_and_shuffle_examples(oracle, device, which, k):
    from halo2_gpu_specific_amd import circuits, prover
    from halo2_gpu_specific_amd.rng import ProverRng

    W, make, syn = {"lookup-api-set": (rp.LookupApiSet, circuits.lookup_api_set, circuits.lookup_api_set_synthesize),
                    "shuffle-api": (rp.ShuffleApi, circuits.shuffle_api, circuits.shuffle_api_synthesize),
                    "shuffle-gates": (rp.shuffle_gates_class(), circuits.shuffle_gates, circuits.shuffle_gates_synthesize)}[which]
    cs = make()
    assert cs.degree() == W.degree and cs.advice_queries == W.advice_queries and cs.fixed_queries == W.fixed_queries
    adv, fixed, copies = syn(k)
    params = srs(oracle, device, k)
    pk = prover.keygen(device, params, cs, fixed, copies)
    if k <= 7:
        radv, rfixed, rcopies = W.synthesize(k)
        rpk = rp.keygen(W, k, S_TRAPDOOR, rfixed, rcopies)
        assert pk.transcript_repr == rpk.transcript_repr and pk.fixed_commitments == rpk.fixed_commitments
        for seed, use_gwc in ((1, False), (2, True)):
            proof = prover.create_proof_ext(device, params, pk, adv, ProverRng(seed), use_gwc)
            want = rp.create_proof(rpk, radv, ProverRng(seed), use_gwc=use_gwc)
            first = next((i for i in range(min(len(proof), len(want))) if proof[i] != want[i]), None)
            assert len(proof) == len(want) and first is None, "differs from the big-integer prover at byte %s" % first
            assert rp.verify_proof(rpk, proof, use_gwc=use_gwc)
    else:
        vk = rp.Keys()
        vk.cs, vk.dom, vk.s = W, rp.Domain(k, cs.degree()), S_TRAPDOOR
        vk.fixed_commitments, vk.perm_commitments, vk.transcript_repr = pk.fixed_commitments, pk.perm_commitments, pk.transcript_repr
        assert rp.verify_proof(vk, prover.create_proof_with_shplonk(device, params, pk, adv, ProverRng(5)))
        assert rp.verify_proof(vk, prover.create_proof(device, params, pk, adv, ProverRng(6)), use_gwc=True)
    bad = [c.copy() for c in adv]
    if which == "lookup-api-set":                 # 10 * 11 is not in the table
        bad[3][1, 0] = 11
        with pytest.raises(Exception):
            prover.create_proof_with_shplonk(device, params, pk, bad, ProverRng(1))
    elif which == "shuffle-api":                  # not a permutation: the product does not close
        bad[2][0, 0] = 5
        with pytest.raises(ValueError):
            prover.create_proof_with_shplonk(device, params, pk, bad, ProverRng(1))
    elif k <= 7:                                  # a shuffled cell changed: the proof is made, and rejected
        bad[5][3, 0] += np.uint64(1)
        assert not rp.verify_proof(rpk, prover.create_proof_with_shplonk(device, params, pk, bad, ProverRng(1)))
