"""Proof-level parity on the GPU: keygen + create_proof on the device-resident C ABI
(halo2-gpu-specific_amd/prover.py) against the independent big-integer prover / verifier of ref_plonk.py on the
same SRS trapdoor, witness and seeded blinding stream.  Small k: proof bytes identical.  Larger k: the proof is
accepted by the big-integer verifier (the pairing check done with the setup trapdoor)."""
import os

import numpy as np
import pytest

import ref_plonk as rp
from h2util import fr_mont, ints_to_arr
from test_plonk_host import S_TRAPDOOR, lookup_shuffle_cs, rot_gate_cs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def device():
    from halo2_gpu_specific_amd import prover

    return prover.Device()


def srs(oracle, device, k):
    """Params::unsafe_setup with the fixed trapdoor: the oracle restatement (pinned by tests/golden/setup_kat.json) up
    to 2^16 points, the device setup (checked against the oracle in test_device_unsafe_setup_matches_oracle, and by
    every verified proof: a wrong SRS cannot satisfy the trapdoor / pairing checks) above"""
    from halo2_gpu_specific_amd import prover

    if k > 16:
        return prover.Params.unsafe_setup(device, k, S_TRAPDOOR)
    n = 1 << k
    g = np.zeros((n, 8), dtype=np.uint64)
    gl = np.zeros((n, 8), dtype=np.uint64)
    s = fr_mont(S_TRAPDOOR)
    oracle.lib.oracle_unsafe_setup(k, s.ctypes.data, g.ctypes.data, gl.ctypes.data)
    return prover.Params(device, k, g, gl)


def cols_to_arr(cols):
    return [ints_to_arr(c) for c in cols]


CASES = [("mini", 4), ("mini", 8), ("mini", 9), ("rot", 5), ("rot", 8)]   # ("mini", 8): the k BASELINE configs[0] names


@pytest.mark.parametrize("which,k", CASES)
def test_proof_bytes_match_big_integer_prover(oracle, device, which, k):
    from halo2_gpu_specific_amd import circuits, prover
    from halo2_gpu_specific_amd.rng import ProverRng

    ref_cs = rp.MiniPlonk if which == "mini" else rp.RotGate
    cs = circuits.mini_plonk() if which == "mini" else rot_gate_cs()
    adv, fixed, copies = ref_cs.synthesize(k)
    params = srs(oracle, device, k)
    pk = prover.keygen(device, params, cs, cols_to_arr(fixed), [(l[0], l[1], r[0], r[1]) for l, r in copies])
    rpk = rp.keygen(ref_cs, k, S_TRAPDOOR, fixed, copies)
    # keygen: commitments to the fixed and sigma columns, the sigma columns themselves, the vk digest
    assert pk.fixed_commitments == rpk.fixed_commitments
    assert pk.perm_commitments == rpk.perm_commitments
    for got, want in zip(pk.sigma_values, rpk.sigma_values):
        assert device.get_rows(got, 0, 1 << k) == want
    assert pk.transcript_repr == rpk.transcript_repr
    for seed, use_gwc in ((1, False), (2, False), (3, True)):
        proof = prover.create_proof_ext(device, params, pk, cols_to_arr(adv), ProverRng(seed), use_gwc)
        want = rp.create_proof(rpk, adv, ProverRng(seed), use_gwc=use_gwc)
        assert len(proof) == len(want)
        first = next((i for i in range(len(proof)) if proof[i] != want[i]), None)
        assert first is None, "proof differs from the big-integer prover at byte %d (field %d)" % (first, first // 32)
        assert rp.verify_proof(rpk, proof, use_gwc=use_gwc)


@pytest.mark.parametrize("which,k", [("mini", 5), ("lookup", 6)])
def test_several_circuit_instances_match_big_integer_prover(oracle, device, which, k):
    """`circuits: &[ConcreteCircuit]` (plonk/prover.rs:206-232): two (three) circuit instances in one proof -- the device
    prover runs every phase circuit by circuit and combines the per-circuit quotients by powers of y; bytes equal to the
    big-integer prover's single Horner fold over all circuits, accepted by its verifier"""
    from halo2_gpu_specific_amd import circuits, prover
    from halo2_gpu_specific_amd.rng import ProverRng
    from test_plonk_host import _second_lookup_shuffle_witness

    if which == "mini":
        ref_cs, cs = rp.MiniPlonk, circuits.mini_plonk()
        adv_a, fixed, copies = ref_cs.synthesize(k, a=5)
        advs = [adv_a, ref_cs.synthesize(k, a=7)[0], ref_cs.synthesize(k, a=11)[0]]
        insts = [(), (), ()]
    else:
        ref_cs, cs = rp.LookupShuffle, lookup_shuffle_cs()
        adv_a, fixed, copies, inst_a = ref_cs.synthesize(k)
        adv_b, inst_b = _second_lookup_shuffle_witness(k)
        advs, insts = [adv_a, adv_b], [inst_a, inst_b]
    params = srs(oracle, device, k)
    pk = prover.keygen(device, params, cs, cols_to_arr(fixed), [(l[0], l[1], r[0], r[1]) for l, r in copies])
    rpk = rp.keygen(ref_cs, k, S_TRAPDOOR, fixed, copies)
    for seed, use_gwc in ((1, False), (2, True)):
        proof = prover.create_proof_ext(device, params, pk, [cols_to_arr(a) for a in advs], ProverRng(seed), use_gwc,
                                        instances=insts)
        want = rp.create_proof(rpk, advs, ProverRng(seed), use_gwc=use_gwc, instances=insts)
        assert len(proof) == len(want)
        first = next((i for i in range(len(proof)) if proof[i] != want[i]), None)
        assert first is None, "proof differs from the big-integer prover at byte %d (field %d)" % (first, first // 32)
        assert rp.verify_proof(rpk, proof, use_gwc=use_gwc, instances=insts, circuits=len(advs))
    # one circuit in the list form is the single-circuit proof
    one = prover.create_proof_ext(device, params, pk, [cols_to_arr(advs[0])], ProverRng(5), False, instances=[insts[0]])
    assert one == prover.create_proof_ext(device, params, pk, cols_to_arr(advs[0]), ProverRng(5), False, instances=insts[0])
    # the coset decomposition of the multi-GPU path (one quotient per coset, every circuit folded into it)
    D2 = prover.Device(force_cosets=True)
    params2 = prover.Params(D2, k, params.g, params.g_lagrange)
    pk2 = prover.keygen(D2, params2, cs, cols_to_arr(fixed), [(l[0], l[1], r[0], r[1]) for l, r in copies])
    got = prover.create_proof_ext(D2, params2, pk2, [cols_to_arr(a) for a in advs], ProverRng(1), False, instances=insts)
    assert got == rp.create_proof(rpk, advs, ProverRng(1), instances=insts)


def test_bad_witness_is_rejected(oracle, device):
    from halo2_gpu_specific_amd import circuits, prover
    from halo2_gpu_specific_amd.rng import ProverRng

    k = 5
    adv, fixed, copies = circuits.mini_plonk_synthesize(k)
    params = srs(oracle, device, k)
    cs = circuits.mini_plonk()
    pk = prover.keygen(device, params, cs, fixed, copies)
    vk = rp.Keys()
    vk.cs, vk.dom, vk.s = rp.MiniPlonk, rp.Domain(k, 3), S_TRAPDOOR
    vk.fixed_commitments, vk.perm_commitments, vk.transcript_repr = pk.fixed_commitments, pk.perm_commitments, pk.transcript_repr
    assert rp.verify_proof(vk, prover.create_proof_with_shplonk(device, params, pk, adv, ProverRng(5)))
    assert rp.verify_proof(vk, prover.create_proof(device, params, pk, adv, ProverRng(5)), use_gwc=True)
    adv[2][0, 0] += 1
    assert not rp.verify_proof(vk, prover.create_proof_with_shplonk(device, params, pk, adv, ProverRng(5)))
    assert not rp.verify_proof(vk, prover.create_proof(device, params, pk, adv, ProverRng(5)), use_gwc=True)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("k", [16, int(os.environ.get("H2_TEST_PLONK_K", "22")), 24])
def test_large_proof_is_accepted_by_big_integer_verifier(oracle, device, k):
    """mini-PLONK at a size where the scans, the multi-pass NTTs and the two-level MSM sort all take their
    multi-workgroup paths, and at BASELINE config 4's full size (k = 22; most of its ~90 s is the oracle's
    `unsafe_setup` of the 2 x 2^22-point SRS on the host cores): checked by the verifier (gate / permutation
    identities at x + the opening equation, through the trapdoor and through the pairing)"""
    from halo2_gpu_specific_amd import circuits, prover
    from halo2_gpu_specific_amd.rng import ProverRng

    adv, fixed, copies = circuits.mini_plonk_synthesize(k)
    params = srs(oracle, device, k)
    cs = circuits.mini_plonk()
    pk = prover.keygen(device, params, cs, fixed, copies)
    # the fixed-column commitments against the trapdoor: [p(s)]G with p(s) from the oracle's Horner
    dom = rp.Domain(k, 3)
    for col, com in zip(fixed, pk.fixed_commitments):
        if k > 22:
            break  # BASELINE configs[4]'s size (k = 24): 2^20 Python inversions -- the k <= 22 runs cover this check
        lag = [int(v) for v in col[:, 0]]
        # p(s) = sum_i v_i L_i(s) with L_i(s) = (s^n - 1) w^i / (n (s - w^i)); the columns are 0/1 valued
        sn1 = (pow(S_TRAPDOOR, dom.n, rp.R) - 1) * dom.n_inv % rp.R
        acc, w = 0, 1
        for v in lag:
            if v:
                acc = (acc + w * rp.inv((S_TRAPDOOR - w) % rp.R)) % rp.R
            w = w * dom.omega % rp.R
        assert com == rp.g1_mul(rp.G1, acc * sn1 % rp.R)
        break  # one column is enough (2^k modular inversions in Python)
    vk = rp.Keys()
    vk.cs, vk.dom, vk.s = rp.MiniPlonk, dom, S_TRAPDOOR
    vk.fixed_commitments, vk.perm_commitments, vk.transcript_repr = pk.fixed_commitments, pk.perm_commitments, pk.transcript_repr
    timings = {}
    proof = prover.create_proof_with_shplonk(device, params, pk, adv, ProverRng(22), timings=timings)
    assert len(proof) == 32 * (3 + 3 + 1 + 2 + 3 + 4 + 1 + 3 + 8 + 2)
    assert rp.verify_proof(vk, proof)
    gwc = prover.create_proof(device, params, pk, adv, ProverRng(23))
    assert len(gwc) == len(proof) + 32 and rp.verify_proof(vk, gwc, use_gwc=True)
    # and through the real pairing e(L, [s]G2) = e(R, G2) (N3) rather than the trapdoor
    assert rp.verify_proof(vk, proof, pairing=True) and rp.verify_proof(vk, gwc, use_gwc=True, pairing=True)
    print("create_proof k=%d:" % k, {n: round(t * 1e3, 2) for n, t in timings.items()})


def test_range_split_msm_matches_full(oracle, device):
    """the multi-GPU split of every commitment (gpu_multiexp_bound, arithmetic.rs:413-440): per-range partial
    points folded on the host equal the single-device MSM -- ranks simulated one after the other on this GPU"""
    from halo2_gpu_specific_amd import parallel
    from halo2_gpu_specific_amd.transcript import jacobian_to_affine

    k = 12
    n = 1 << k
    params = srs(oracle, device, k)
    cols = [device.upload(oracle.random_fr(300 + j, n)) for j in range(3)]
    extra = device.upload(oracle.random_fr(310, n))
    full = device.msm_batch(cols, params.g_lagrange, n, 254, also=(extra, params.g))
    for world in (2, 4, 8, 3):
        parts = []
        for rank in range(world):
            lo, hi = parallel.msm_split_range(n, world, rank)
            parts.append(device.msm_partial(cols, params.g_lagrange, lo, hi, 254, also=(extra, params.g)))
        parts = np.stack(parts)
        got = [jacobian_to_affine(parallel.g1_sum(parts[:, j, :])) for j in range(4)]
        assert got == full


@pytest.mark.timeout(600)
def test_range_split_msm_matches_full_k24(device):
    """config 5's size: 2^24-point commitments split over simulated worlds of 2, 4 and 8 ranks fold to the single-device
    commitment (uniform scalars from the device generator, SRS-shaped bases from the device setup)"""
    from halo2_gpu_specific_amd import parallel, prover
    from halo2_gpu_specific_amd._lib import check
    from halo2_gpu_specific_amd.transcript import jacobian_to_affine

    k = 24
    n = 1 << k
    params = prover.Params.unsafe_setup(device, k, S_TRAPDOOR)
    cols = []
    for j in range(2):
        t = device.empty(n)
        check(device.L.h2_dev_random_fr(bytes([j + 1]) * 32, n, t.data_ptr(), device.stream), "h2_dev_random_fr")
        cols.append(t)
    with device.torch.cuda.stream(device.tstream):
        cols[1][: n // 3] = 0                                   # a sparse stretch: ranks see different digit mixes
    full = device.msm_batch(cols, params.g_lagrange, n, 254, also=(cols[0], params.g))
    for world in (2, 4, 8):
        parts = []
        for rank in range(world):
            lo, hi = parallel.msm_split_range(n, world, rank)
            parts.append(device.msm_partial(cols, params.g_lagrange, lo, hi, 254, also=(cols[0], params.g)))
        parts = np.stack(parts)
        got = [jacobian_to_affine(parallel.g1_sum(parts[:, j, :])) for j in range(3)]
        assert got == full, world


def test_proof_bytes_match_committed_hashes(oracle, device):
    """tests/golden/proof_hash_kat.json (gen_proof_hash_golden.py: the big-integer big-integer prover run once, k = 10 ..
    18): the device prover reproduces the same proof BYTES -- verifier acceptance alone would not catch a
    wrong-but-valid blinding or ordering change at these sizes"""
    import hashlib

    from h2util import load_golden
    from halo2_gpu_specific_amd import circuits, prover
    from halo2_gpu_specific_amd.rng import ProverRng

    makers = {"mini-plonk": (rp.MiniPlonk, circuits.mini_plonk), "rot-gate": (rp.RotGate, rot_gate_cs),
              "lookup-shuffle": (rp.LookupShuffle, lookup_shuffle_cs)}
    for case in load_golden("proof_hash_kat.json"):
        ref_cs, make = makers[case["circuit"]]
        k = case["k"]
        if ref_cs is rp.MiniPlonk:
            adv, fixed, copies = circuits.mini_plonk_synthesize(k)       # numpy: the Python lists are slow at 2^18
            inst = []
        else:
            syn = ref_cs.synthesize(k)
            adv, fixed = cols_to_arr(syn[0]), cols_to_arr(syn[1])
            copies = [(l[0], l[1], r[0], r[1]) for l, r in syn[2]]
            inst = syn[3] if len(syn) > 3 else []
        params = prover.Params.unsafe_setup(device, k, int(case["trapdoor"], 16))
        pk = prover.keygen(device, params, make(), fixed, copies)
        assert pk.transcript_repr == int(case["vk_digest"], 16), (case["circuit"], k)
        proof = prover.create_proof_ext(device, params, pk, adv, ProverRng(case["seed"]), case["scheme"] == "gwc",
                                        instances=inst)
        assert len(proof) == case["length"]
        assert hashlib.sha256(proof).hexdigest() == case["sha256"], (case["circuit"], k, case["scheme"])


@pytest.mark.parametrize("which,k", [("mini", 7), ("rot", 6), ("lookup", 6)])
def test_coset_path_reproduces_the_proof(oracle, device, which, k):
    """the multi-GPU decomposition on one device: with force_cosets the quotient is evaluated coset by coset (n-point
    transforms with zeta := zeta w_ext^j, the inverse coset transform, the inverse-Vandermonde un-mixing of the pieces)
    instead of on the whole extended domain -- c = 2 (degree 3), 4 (degree 4) and 8 (degree 6) cosets; the proof bytes
    must not change, and equal the big-integer prover's"""
    from halo2_gpu_specific_amd import circuits, prover
    from halo2_gpu_specific_amd.rng import ProverRng

    ref_cs, make = {"mini": (rp.MiniPlonk, circuits.mini_plonk), "rot": (rp.RotGate, rot_gate_cs),
                    "lookup": (rp.LookupShuffle, lookup_shuffle_cs)}[which]
    syn = ref_cs.synthesize(k)
    adv, fixed, copies = cols_to_arr(syn[0]), cols_to_arr(syn[1]), [(l[0], l[1], r[0], r[1]) for l, r in syn[2]]
    inst = syn[3] if len(syn) > 3 else []
    params = srs(oracle, device, k)
    pk = prover.keygen(device, params, make(), fixed, copies)
    want = prover.create_proof_ext(device, params, pk, adv, ProverRng(31), False, instances=inst)
    D2 = prover.Device(force_cosets=True)
    dom = pk.domain
    assert D2.coset_plan(dom)[2] == list(range(dom.quotient_poly_degree))     # the cosets that determine the quotient
    params2 = prover.Params(D2, k, params.g, params.g_lagrange)
    pk2 = prover.keygen(D2, params2, make(), fixed, copies)
    assert pk2.fixed_cosets is None and sorted(pk2.coset) == D2.coset_plan(dom)[2]
    # the per-coset tables are the stride-c subsets of the extended tables
    c = 1 << (dom.extended_k - dom.k)
    for j in (0, dom.quotient_poly_degree - 1):
        assert device.torch.equal(pk2.coset[j]["l_active_row"], pk.l_active_row[j::c])
        assert device.torch.equal(pk2.coset[j]["fixed"][0], pk.fixed_cosets[0][j::c])
    for use_gwc in (False, True):
        got = prover.create_proof_ext(D2, params2, pk2, adv, ProverRng(31), use_gwc, instances=inst)
        if not use_gwc:
            assert got == want
        rpk = rp.keygen(ref_cs, k, S_TRAPDOOR, syn[1], syn[2])
        assert got == rp.create_proof(rpk, syn[0], ProverRng(31), use_gwc=use_gwc, instances=inst)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,which,k", [(2, "mini", 10), (2, "mini", 17), (4, "mini", 12), (8, "mini", 10), (2, "lookup", 9),
                                           (4, "lookup", 8), (4, "wide", 9), (3, "mini", 10), (3, "lookup", 8),
                                           (2, "fuzz:201", 8), (4, "fuzz:202", 8), (3, "fuzz:203", 8), (2, "fuzz:204", 9),
                                           (4, "fuzz:205", 9), (8, "fuzz:206", 9), (8, "wide", 9), (8, "mini", 12),
                                           (8, "fuzz:207", 9), (4, "fuzz:208", 10), (8, "wide16", 10)])
def test_gloo_ranks_on_one_gpu_prove_the_single_device_bytes(oracle, device, tmp_path, world, which, k):
    """config 5's data flow with 2 / 4 / 8 ranks (here processes sharing cuda:0 over gloo; RCCL refuses two ranks on one
    device): every MSM range-split + all-gather + fold; the extended domain split by coset, the per-coset quotients
    scattered as coefficient ranges + un-mixed; the permutation / lookup / shuffle products, the evaluations and the
    multiopen argument (SHPLONK and GWC) on row / coefficient ranges with one field element per rank exchanged per scan /
    Kate division.  Every rank must emit the single-device proof, for the mini-PLONK circuit, the lookup + shuffle +
    instance circuit (degree 6: 5 cosets over 2 or 4 ranks), the wide circuit (degree 5, eight grand sums) and random
    satisfied circuits (tools/prover_fuzz.py: gates with rotations -2 .. 2 reaching across the ranks' row ranges, lookups
    with two input sets, shuffles, instance columns).  With more ranks than cosets (8 or 4 ranks over the 2 cosets of
    mini-PLONK, 8 over the 4 of the wide circuit) the ranks of a coset deal its columns for the coset transforms, exchange row
    slices with the rotations' halo and evaluate the quotient on a row range each (parallel.exchange_row_slices)."""
    import subprocess
    import sys

    from h2util import ROOT
    from halo2_gpu_specific_amd import circuits, prover
    from halo2_gpu_specific_amd.rng import ProverRng

    cs, adv, fixed, copies, inst = _multi_rank_case(which, k)
    params = prover.Params.unsafe_setup(device, k, S_TRAPDOOR)
    pk = prover.keygen(device, params, cs, fixed, copies)
    want = [prover.create_proof_ext(device, params, pk, adv, ProverRng(9), gwc, instances=inst) for gwc in (False, True)]
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % (ROOT, os.path.join(ROOT, "tests"), which, k, S_TRAPDOOR))
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)],
                         capture_output=True, text=True, timeout=800,
                         # (H2_POISON_EMPTY: the workers' "uninitialised" vectors hold all-ones words, so a pass that reads rows
                         # outside the range / slice its rank owns changes the proof on every run)
                         env=dict(os.environ, H2_TEST_BACKEND="gloo", H2_POISON_EMPTY="1"))
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    for tag, proof in (("PROOF", want[0]), ("GWC", want[1])):
        got = [l.split()[1] for l in res.stdout.splitlines() if l.startswith(tag + " ")]
        assert len(got) == world and all(_same_proof(h, proof) for h in got), tag
    secure = [l.split()[1] for l in res.stdout.splitlines() if l.startswith("SECURE ")]
    assert len(secure) == world and len(set(secure)) == 1 and not _same_proof(secure[0], want[0])
    if which == "mini":
        vk = rp.Keys()
        vk.cs, vk.dom, vk.s = rp.MiniPlonk, rp.Domain(k, 3), S_TRAPDOOR
        vk.fixed_commitments, vk.perm_commitments, vk.transcript_repr = pk.fixed_commitments, pk.perm_commitments, pk.transcript_repr
        assert rp.verify_proof(vk, bytes.fromhex(secure[0]))


def _same_proof(text, proof):
    """a worker's line against the expected bytes: hex for short proofs, "sha256:<digest>" for long ones (_WORKER.emit)"""
    import hashlib

    if text.startswith("sha256:"):
        return text[7:] == hashlib.sha256(proof).hexdigest()
    return bytes.fromhex(text) == proof


def _multi_rank_case(which, k):
    """(constraint system, advice, fixed, copies, instances) of the circuits the multi-rank tests prove"""
    from halo2_gpu_specific_amd import circuits

    if which == "mini":
        adv, fixed, copies = circuits.mini_plonk_synthesize(k)
        return circuits.mini_plonk(), adv, fixed, copies, []
    if which.startswith("wide"):                       # "wide": 4 quads (16 columns, 2 lookups); "wide16": the bench's 64 columns
        quads = int(which[4:] or 4)
        adv, fixed, copies = circuits.wide_synthesize(k, quads)
        return circuits.wide(quads), adv, fixed, copies, []
    if which.startswith("fuzz:"):                      # a random satisfied circuit of tools/prover_fuzz.py
        import sys

        from h2util import ROOT

        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import prover_fuzz

        cs, k2, adv, fixed, copies, inst = prover_fuzz.random_case(int(which[5:]), satisfiable=True, k=k)
        assert k2 == k, "the circuit needs more rows than 2^%d" % k
        return cs, adv, fixed, copies, inst
    syn = rp.LookupShuffle.synthesize(k)
    return (lookup_shuffle_cs(), cols_to_arr(syn[0]), cols_to_arr(syn[1]), [(l[0], l[1], r[0], r[1]) for l, r in syn[2]], syn[3])


def _visible_devices():
    import torch

    return torch.cuda.device_count()


@pytest.mark.skipif("_visible_devices() < 2", reason="needs >= 2 GPUs (runs on any multi-GPU box)")
def test_msm_multi_over_the_device_pool(oracle):
    """gpu_multiexp_bound (arithmetic.rs:413-440) with N_GPU > 1: h2_msm_multi cuts the MSM into ceil(n / N_GPU) chunks,
    one pooled device each, and folds the partial points on the host"""
    import ctypes

    import halo2_gpu_specific_amd as h2

    L = h2.lib()
    assert L.h2_device_count() >= 2
    for n in (1 << 15) + 3, 1 << 18:
        s, p = oracle.random_fr(501, n), oracle.random_g1(502, n)
        out = np.zeros(12, dtype=np.uint64)
        assert L.h2_msm_multi(s.ctypes.data, p.ctypes.data, n, 254, out.ctypes.data) == 0, L.h2_last_error()
        one = np.zeros(12, dtype=np.uint64)
        assert L.h2_msm(s.ctypes.data, p.ctypes.data, n, 254, one.ctypes.data) == 0
        aff = lambda r: oracle.to_affine(r.reshape(1, 12)).tobytes()  # noqa: E731
        assert aff(out) == aff(one) == aff(oracle.best_multiexp(s, p))


@pytest.mark.skipif("_visible_devices() < 2", reason="needs >= 2 GPUs (runs on any multi-GPU box)")
@pytest.mark.timeout(600)
def test_two_rank_rccl_proof_equals_single_device_proof(oracle, device, tmp_path):
    """config 5's exchange on real links: two processes, one GPU each, RCCL group -- every commitment is range-split,
    the partial points all-gathered on the device and folded; both ranks must emit the single-device proof bytes"""
    import subprocess
    import sys

    from h2util import ROOT
    from halo2_gpu_specific_amd import circuits, prover
    from halo2_gpu_specific_amd.rng import ProverRng

    k = 12
    adv, fixed, copies = circuits.mini_plonk_synthesize(k)
    params = prover.Params.unsafe_setup(device, k, S_TRAPDOOR)
    pk = prover.keygen(device, params, circuits.mini_plonk(), fixed, copies)
    want = prover.create_proof_with_shplonk(device, params, pk, adv, ProverRng(9))
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % (ROOT, os.path.join(ROOT, "tests"), "mini", k, S_TRAPDOOR))
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)],
                         capture_output=True, text=True, timeout=500)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    proofs = [l.split()[1] for l in res.stdout.splitlines() if l.startswith("PROOF ")]
    assert len(proofs) == 2 and all(bytes.fromhex(h) == want for h in proofs)


# A stream-order check shared by the worker scripts below: a background thread keeps torch's DEFAULT stream busy (one 30 ms sleep
# kernel always queued) for as long as the proofs run.  Everything of a proof runs on the device's own streams (created with
# torch.cuda.Stream(): not ordered with the default stream), so a tensor that some helper builds on the default stream by mistake
# -- an index vector, a mask -- is not there yet when the proof's stream reads it, and the proof comes out wrong HERE instead of
# only when eight processes contend for one GPU (parallel.exchange_row_slices once built its row indices that way).
_STALLER = r"""
import threading
_stall = {"on": True, "device": torch.cuda.current_device()}
def _keep_default_stream_busy():
    torch.cuda.set_device(_stall["device"])
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); torch.cuda._sleep(20_000_000); b.record(); b.synchronize()
    cycles = int(0.03 * 20_000_000 / max(a.elapsed_time(b) * 1e-3, 1e-6))
    done = torch.cuda.Event()
    while _stall["on"]:
        torch.cuda._sleep(cycles); done.record(); done.synchronize()
_staller = threading.Thread(target=_keep_default_stream_busy, daemon=True)
if os.environ.get("H2_TEST_STALL", "1") != "0":
    _staller.start()
def stop_staller():
    _stall["on"] = False
    if _staller.is_alive():
        _staller.join()
    torch.cuda.synchronize()
"""

_WORKER = r"""
import os, sys
sys.path.insert(0, %r)
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
rank = int(os.environ["RANK"]); local = int(os.environ.get("LOCAL_RANK", rank))
if os.environ.get("H2_TEST_BACKEND") == "gloo":      # ranks sharing cuda:0 (RCCL refuses two ranks on one device)
    local = 0
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
else:
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
from halo2_gpu_specific_amd import prover
from halo2_gpu_specific_amd.rng import ProverRng
from test_gpu_plonk import _multi_rank_case
D = prover.Device(local)
which, k = %r, %d
params = prover.Params.unsafe_setup(D, k, %d)
cs, adv, fixed, copies, inst = _multi_rank_case(which, k)
pk = prover.keygen(D, params, cs, fixed, copies)
# the O(n) passes really are range-sharded (a world that does not divide 2^k keeps them replicated: uneven MSM ranges only)
assert (D.row_range(1 << k) != (0, 1 << k)) == ((1 << k) %% dist.get_world_size() == 0)
__STALLER__
import hashlib
def emit(tag, data):
    # one write() of at most PIPE_BUF bytes is atomic on the pipe the ranks share: a long proof goes out as its hash
    text = data.hex() if len(data) <= 1500 else "sha256:" + hashlib.sha256(data).hexdigest()
    os.write(1, (tag + " " + text + "\n").encode())
proof = prover.create_proof_ext(D, params, pk, adv, ProverRng(9), False, instances=inst)
# more ranks than cosets (a multiple of them): the ranks of a coset really shared its work (coset rank groups)
c_, w_ = pk.domain.quotient_poly_degree, dist.get_world_size()
assert bool(getattr(D, "_coset_groups", {})) == (w_ %% c_ == 0 and w_ // c_ >= 2), (c_, w_)
emit("PROOF", proof)
gwc = prover.create_proof_ext(D, params, pk, adv, ProverRng(9), True, instances=inst)
emit("GWC", gwc)
# OS-entropy blinding: rank 0's key is broadcast (ProverRng.shared), so the ranks still agree on every byte
secure = prover.create_proof_ext(D, params, pk, adv, ProverRng(), False, instances=inst)
emit("SECURE", secure)
sys.stdout.flush()
stop_staller()
dist.barrier()
dist.destroy_process_group()
""".replace("__STALLER__", _STALLER)


def _free_port():
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def test_collective_proof_equals_single_device_proof(oracle, device, tmp_path):
    """one rank, RCCL process group, every commitment through the range-split + all-gather + fold path: the
    proof bytes are those of the plain single-device run (a second process: the group is process-global)"""
    import subprocess
    import sys

    from h2util import ROOT
    from halo2_gpu_specific_amd import circuits, prover
    from halo2_gpu_specific_amd.rng import ProverRng

    k = 7
    adv, fixed, copies = circuits.mini_plonk_synthesize(k)
    params = srs(oracle, device, k)
    pk = prover.keygen(device, params, circuits.mini_plonk(), fixed, copies)
    want = prover.create_proof_with_shplonk(device, params, pk, adv, ProverRng(9))
    g_path, gl_path = tmp_path / "g.npy", tmp_path / "gl.npy"
    np.save(g_path, device.download(params.g).reshape(-1, 8))
    np.save(gl_path, device.download(params.g_lagrange).reshape(-1, 8))
    script = tmp_path / "worker.py"
    script.write_text(r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
torch.cuda.init()
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from halo2_gpu_specific_amd import circuits, prover
from halo2_gpu_specific_amd.rng import ProverRng
D = prover.Device(0, force_collective=True)
__STALLER__
k = %d
params = prover.Params(D, k, np.load(%r), np.load(%r))
adv, fixed, copies = circuits.mini_plonk_synthesize(k)
pk = prover.keygen(D, params, circuits.mini_plonk(), fixed, copies)
proof = prover.create_proof_with_shplonk(D, params, pk, adv, ProverRng(9))
sys.stdout.write("PROOF " + proof.hex() + "\n")
stop_staller()
dist.destroy_process_group()
""".replace("__STALLER__", _STALLER) % (ROOT, k, str(g_path), str(gl_path)))
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    res = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=280)
    assert res.returncode == 0, res.stdout + res.stderr
    line = [l for l in res.stdout.splitlines() if l.startswith("PROOF ")][0]
    assert bytes.fromhex(line.split()[1]) == want


def test_logup_counts_by_row_ranges_sum_to_the_multiplicities(oracle, device):
    """h2_dev_logup_counts over the row ranges of 1, 2, 4 and 8 ranks, the raw counters summed (what the ranks' all-reduce
    does) and h2_dev_logup_emit: the field elements h2_dev_logup_multiplicity gives for the whole columns (itself checked
    against the oracle's multiplicities in the proof tests); a value missing from the table is counted in the last word"""
    import ctypes

    import torch

    from halo2_gpu_specific_amd._lib import check
    from halo2_gpu_specific_amd import parallel

    L, dev = device.L, device.dev
    n, usable = 1 << 12, (1 << 12) - 6
    rng = np.random.default_rng(3)
    table = np.zeros((n, 4), dtype=np.uint64)
    table[:300, 0] = np.arange(300) * 7 + 1            # 300 distinct values, then padding zeros (a duplicated value)
    ins = []
    for _ in range(3):
        col = np.zeros((n, 4), dtype=np.uint64)
        col[:usable, 0] = table[rng.integers(0, 400, size=usable), 0]
        ins.append(col)
    up = lambda a: torch.from_numpy(a.view(np.int64)).to(dev)  # noqa: E731
    d_table, d_ins = up(table), [up(c) for c in ins]
    ptrs = (ctypes.c_void_p * 3)(*[t.data_ptr() for t in d_ins])
    nbytes = L.h2_logup_scratch_bytes(n)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    want = torch.empty((n, 4), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    check(L.h2_dev_logup_multiplicity(d_table.data_ptr(), ptrs, 3, usable, n, want.data_ptr(), scratch.data_ptr(), nbytes, None),
          "h2_dev_logup_multiplicity")
    for world in (1, 2, 4, 8):
        total = torch.zeros(n + 1, dtype=torch.int32, device=dev)
        for rank in range(world):
            lo, hi = parallel.msm_split_range(n, world, rank)
            part = torch.empty(n + 1, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            check(L.h2_dev_logup_counts(d_table.data_ptr(), ptrs, 3, usable, n, lo, hi, part.data_ptr(), scratch.data_ptr(), nbytes,
                                        None), "h2_dev_logup_counts")
            torch.cuda.synchronize()
            total += part
        assert int(total[-1]) == 0 and int(total[:n].sum()) == 3 * usable
        got = torch.empty((n, 4), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        check(L.h2_dev_logup_emit(total.data_ptr(), usable, n, got.data_ptr(), None), "h2_dev_logup_emit")
        torch.cuda.synchronize()
        assert torch.equal(got, want), world
    bad = ins[1].copy()
    bad[77, 0] = 5                                      # not a table value
    d_bad = up(bad)
    ptrs2 = (ctypes.c_void_p * 1)(d_bad.data_ptr())
    part = torch.empty(n + 1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    check(L.h2_dev_logup_counts(d_table.data_ptr(), ptrs2, 1, usable, n, 0, 128, part.data_ptr(), scratch.data_ptr(), nbytes, None),
          "h2_dev_logup_counts")
    torch.cuda.synchronize()
    assert int(part[-1]) == 1 and int(part[:n].sum()) == 127


def test_rccl_collective_shapes_on_one_rank(tmp_path):
    """every RCCL-only code path of parallel.py (the in-place all_gather_into_tensor of allgather_rows, the scatter of
    scatter_cosets, gather_rows_to, the asynchronous broadcasts of broadcast_columns_begin on a second communicator and a
    side stream, the device-side fold, the small all-gathers / all-reduces) under a ONE-rank RCCL group, each at least
    twice and with the view sizes a larger world would use: a mis-sized view or a wrong stream order fails here, on one
    GPU, not on eight.  Round 5: a 256 MiB broadcast on the second communicator / side stream in flight while the first
    communicator runs 96-byte all-gathers in a loop (the two-communicator overlap of DESIGN section 6)"""
    import subprocess
    import sys

    from h2util import ROOT

    script = tmp_path / "worker.py"
    script.write_text(r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
torch.cuda.init()
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from halo2_gpu_specific_amd import parallel, prover
D = prover.Device(0, force_collective=True)
__STALLER__
bulk = dist.new_group()
dev, stream, side = D.dev, D.tstream, D.copy_stream
g = torch.Generator(device=dev); g.manual_seed(5)
for rep in range(3):
    n = 1 << (10 + rep)
    with torch.cuda.stream(stream):
        t = torch.randint(-2**62, 2**62, (n, 4), dtype=torch.int64, device=dev, generator=g)
        want = t.clone()
    parallel.allgather_rows(t, 0, n, stream=stream)
    parallel.gather_rows_to(t, 0, n, 0, stream=stream)
    stream.synchronize()
    assert torch.equal(t, want), "allgather_rows / gather_rows_to"
    cols = [t, want.clone(), want.clone()]
    arrival = parallel.broadcast_columns_begin(cols, [0, 0, 0], group=bulk, stream=stream, side=side)
    arrival.wait(); arrival.wait()
    stream.synchronize()
    assert all(torch.equal(c, want) for c in cols), "broadcast_columns_begin"
    mine = {j: want.clone() + j for j in range(3)}
    got = parallel.scatter_cosets(mine, 3, 1, 0, n, stream=stream)
    stream.synchronize()
    assert all(torch.equal(got[j], want + j) for j in range(3)), "scatter_cosets"
    got = parallel.exchange_cosets(mine, 3, 1, stream=stream)
    stream.synchronize()
    assert all(torch.equal(got[j], want + j) for j in range(3)), "exchange_cosets"
    # the coset rank group's row-slice exchange (all-to-all under RCCL): one member keeps its columns, nothing moves
    for _ in range(2):
        keep = [want.clone(), want.clone() + 1]
        parallel.exchange_row_slices(keep, [0, 0], n, 1, 0, 6, 1, stream=stream)
        stream.synchronize()
        assert torch.equal(keep[0], want) and torch.equal(keep[1], want + 1), "exchange_row_slices"
    with torch.cuda.stream(stream):
        cnt = torch.arange(n + 1, dtype=torch.int32, device=dev); cnt[-1] = 0
    assert parallel.allreduce_counts(cnt, stream=stream) == 0 and int(cnt[5]) == 5
    assert parallel.allreduce_max([3, 254, rep], device=dev) == [3, 254, rep]
    vals = [(1 << 200) + rep, 7, 0]
    assert parallel.allgather_scalars(vals, device=dev) == [vals]
    pts = np.zeros((2, 12), dtype=np.uint64); pts[:, 4] = 1          # two identities (0 : 1 : 0) in canonical limbs ...
    out = parallel.allgather_fold_many(pts, device=dev, stream=stream)
    assert out.shape == (2, 12) and not out[:, 8:].any()               # ... fold to the identity (z = 0)
# the overlap DESIGN section 6 depends on: a BULK broadcast (8 x 32 MiB = 256 MiB) in flight on the second communicator and
# the side stream WHILE the first communicator runs the commitments' 96-byte all-gathers on the compute stream -- both
# complete, in either order of completion, with the right contents
with torch.cuda.stream(stream):
    big = [torch.randint(-2**62, 2**62, (1 << 20, 4), dtype=torch.int64, device=dev, generator=g) for _ in range(8)]
    want_big = [b.clone() for b in big]
for rep in range(2):
    arrival = parallel.broadcast_columns_begin(big, [0] * 8, group=bulk, stream=stream, side=side)
    pts = np.zeros((3, 12), dtype=np.uint64); pts[:, 4] = 1
    for i in range(150):
        out = parallel.allgather_fold_many(pts, device=dev, stream=stream)
        assert out.shape == (3, 12) and not out[:, 8:].any()
        if i == 75 and rep == 1:
            arrival.wait()                           # ... and a wait in the middle of the small collectives
    arrival.wait()
    stream.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(big, want_big)), "bulk broadcast under small all-gathers"
del big, want_big
# the sharded inverse transforms of a proof, through Device: one rank owns every column
dom = prover.Domain(12, 3)
with torch.cuda.stream(stream):
    cols = [torch.randint(0, 2**60, (dom.n, 4), dtype=torch.int64, device=dev, generator=g) for _ in range(3)]
    for c in cols: c[:, 1:] = 0
    want = [D.intt(c.clone(), dom) for c in cols]
D.group_size = 1                                   # (row_range of a one-rank group is the whole vector either way)
out, arrival = D.intt_columns_begin(cols, dom, complete=True, keep=True)
assert arrival is None
stream.synchronize()
assert all(torch.equal(a, b) for a, b in zip(out, want))
sys.stdout.write("COLLECTIVES OK\n")
stop_staller()
dist.destroy_process_group()
""".replace("__STALLER__", _STALLER) % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    res = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=280)
    assert res.returncode == 0 and "COLLECTIVES OK" in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]


@pytest.mark.parametrize("k", [5, 8])
def test_lookup_shuffle_instance_proof_bytes(oracle, device, k):
    """instance column + logup lookups (two input sets, a duplicated table row) + a shuffle group, end to end:
    multiplicities from the device hash table, grand sums / products from the scans, the lookup and shuffle terms of
    the fused evaluate_h -- bytes equal to the big-integer prover's, SHPLONK and GWC"""
    from halo2_gpu_specific_amd import prover
    from halo2_gpu_specific_amd.rng import ProverRng

    adv, fixed, copies, inst = rp.LookupShuffle.synthesize(k)
    params = srs(oracle, device, k)
    pk = prover.keygen(device, params, lookup_shuffle_cs(), cols_to_arr(fixed), [(l[0], l[1], r[0], r[1]) for l, r in copies])
    rpk = rp.keygen(rp.LookupShuffle, k, S_TRAPDOOR, fixed, copies)
    assert pk.transcript_repr == rpk.transcript_repr
    for seed, use_gwc in ((1, False), (2, True)):
        proof = prover.create_proof_ext(device, params, pk, cols_to_arr(adv), ProverRng(seed), use_gwc, instances=inst)
        want = rp.create_proof(rpk, adv, ProverRng(seed), use_gwc=use_gwc, instances=inst)
        assert len(proof) == len(want)
        first = next((i for i in range(len(proof)) if proof[i] != want[i]), None)
        assert first is None, "proof differs from the big-integer prover at byte %d (field %d)" % (first, first // 32)
        assert rp.verify_proof(rpk, proof, use_gwc=use_gwc, instances=inst)
    # a value missing from the table: the reference panics, the library reports it
    bad = [c[:] for c in adv]
    bad[4][2] = 5
    from halo2_gpu_specific_amd._lib import H2Error
    with pytest.raises(H2Error, match="missing from the table"):
        prover.create_proof_with_shplonk(device, params, pk, cols_to_arr(bad), ProverRng(1), instances=inst)
    bad = [c[:] for c in adv]
    bad[10][1] += 1
    with pytest.raises(ValueError, match="shuffle"):
        prover.create_proof_with_shplonk(device, params, pk, cols_to_arr(bad), ProverRng(1), instances=inst)


def test_lookup_proof_at_a_multi_workgroup_size_is_accepted(oracle, device):
    """the lookup / shuffle / instance circuit at k = 12 (hash table, additive and multiplicative scans and the
    logup kernels of evaluate_h beyond one workgroup), checked by the big-integer verifier on the device keygen's
    commitments"""
    from halo2_gpu_specific_amd import prover
    from halo2_gpu_specific_amd.rng import ProverRng

    k = 12
    adv, fixed, copies, inst = rp.LookupShuffle.synthesize(k)
    params = srs(oracle, device, k)
    pk = prover.keygen(device, params, lookup_shuffle_cs(), cols_to_arr(fixed), [(l[0], l[1], r[0], r[1]) for l, r in copies])
    vk = rp.Keys()
    vk.cs, vk.dom, vk.s = rp.LookupShuffle, rp.Domain(k, 6), S_TRAPDOOR
    vk.fixed_commitments, vk.perm_commitments, vk.transcript_repr = pk.fixed_commitments, pk.perm_commitments, pk.transcript_repr
    proof = prover.create_proof_with_shplonk(device, params, pk, cols_to_arr(adv), ProverRng(3), instances=inst)
    assert rp.verify_proof(vk, proof, instances=inst)
    assert not rp.verify_proof(vk, proof, instances=[[42, 8]])


def test_params_file_and_witness_file_round_trip(oracle, device, tmp_path):
    """N4: Params::{write, read} with device-side point (de)compression, store_witness / fetch_witness, and
    create_proof_from_witness producing the same bytes as create_proof on the in-memory inputs"""
    from halo2_gpu_specific_amd import circuits, formats, prover
    from halo2_gpu_specific_amd.rng import ProverRng
    from halo2_gpu_specific_amd.transcript import point_to_bytes
    from h2util import arr_to_points, to_mont

    k = 7
    params = srs(oracle, device, k)
    path = tmp_path / "params.bin"
    formats.params_write(device, params, path, additional_data=b"\x01" * 64)
    raw = path.read_bytes()
    assert len(raw) == 4 + 2 * 32 * (1 << k) + 4 + 64 and raw[:4] == bytes([k, 0, 0, 0])
    # the stored encodings, point by point, against the host encoder on the oracle's normalised points
    g_pts = arr_to_points(device.download(params.g).reshape(-1, 8))
    for i in (0, 1, 5, (1 << k) - 1):
        assert raw[4 + 32 * i:4 + 32 * i + 32] == point_to_bytes(g_pts[i])
    back, extra = formats.params_read(device, path)
    assert extra == b"\x01" * 64 and back.k == k
    assert np.array_equal(device.download(back.g), device.download(params.g))
    assert np.array_equal(device.download(back.g_lagrange), device.download(params.g_lagrange))
    # a corrupted abscissa is rejected (the reference unwraps from_bytes)
    bad = bytearray(raw)
    bad[4 + 32 * 3] ^= 1
    good_count = 0
    for tweak in range(1, 6):             # about half of all x are not on the curve: one of a few tweaks must fail
        bad[4 + 32 * 3 + 1] = (raw[4 + 32 * 3 + 1] + tweak) & 0xFF
        (tmp_path / "bad.bin").write_bytes(bytes(bad))
        try:
            formats.params_read(device, tmp_path / "bad.bin")
            good_count += 1
        except Exception as e:  # noqa: BLE001
            assert "not curve points" in str(e)
    assert good_count < 5
    # witness file -> create_proof_from_witness
    adv, fixed, copies = circuits.mini_plonk_synthesize(k)
    pk = prover.keygen(device, back, circuits.mini_plonk(), fixed, copies)
    want = prover.create_proof(device, back, pk, adv, ProverRng(6))
    from h2util import from_mont  # noqa: F401

    mont_cols = [to_mont([int(v) for v in c[:, 0]]) for c in adv]
    wpath = tmp_path / "witness.bin"
    formats.witness_store(wpath, k, mont_cols)
    assert wpath.stat().st_size == 4 + 3 * (32 << k)
    cols = formats.witness_fetch(wpath, k)
    assert len(cols) == 3 and np.array_equal(cols[1], mont_cols[1])
    assert prover.create_proof_from_witness(device, back, pk, cols, ProverRng(6)) == want


def test_device_point_codec_on_every_srs_point(oracle, device):
    """N4 (poly/commitment.rs:241-294): h2_dev_points_compress over EVERY point of both tables of a k = 13 setup against
    the big-integer encoder (ref_plonk.point_to_bytes on the oracle's normalised points: written independently of the
    product's transcript.point_to_bytes) and against the C oracle's encoder; h2_dev_points_decompress of those bytes
    against the oracle's square root (oracle_points_decompress), bit for bit; invalid encodings are refused"""
    import torch

    from halo2_gpu_specific_amd._lib import check
    from h2util import arr_to_points

    k = 13
    n = 1 << k
    params = srs(oracle, device, k)
    for table in (params.g, params.g_lagrange):
        pts = device.download(table).reshape(-1, 8)
        with torch.cuda.stream(device.tstream):
            out = torch.empty((n, 32), dtype=torch.uint8, device=device.dev)
        check(device.L.h2_dev_points_compress(table.data_ptr(), n, out.data_ptr(), device.stream), "h2_dev_points_compress")
        with torch.cuda.stream(device.tstream):
            raw = out.cpu().numpy()
        assert np.array_equal(raw, oracle.points_compress(pts))
        ints = arr_to_points(pts)
        assert [bytes(r) for r in raw] == [rp.point_to_bytes(None if P == (0, 0) else P) for P in ints]
        want, bad = oracle.points_decompress(raw)
        assert bad == 0 and np.array_equal(want, pts)
        with torch.cuda.stream(device.tstream):
            back = torch.empty((n, 8), dtype=torch.int64, device=device.dev)
        check(device.L.h2_dev_points_decompress(out.data_ptr(), n, back.data_ptr(), device.stream), "h2_dev_points_decompress")
        assert np.array_equal(device.download(back), want)
    # random points of both parities plus the identity, and a batch with junk in it
    pts = oracle.random_g1(91, 5000)
    pts[17] = 0
    raw = oracle.points_compress(pts)
    with torch.cuda.stream(device.tstream):
        d_raw = torch.from_numpy(raw).to(device.dev)
        back = torch.empty((len(pts), 8), dtype=torch.int64, device=device.dev)
    check(device.L.h2_dev_points_decompress(d_raw.data_ptr(), len(pts), back.data_ptr(), device.stream), "h2_dev_points_decompress")
    assert np.array_equal(device.download(back), pts)
    junk = raw.copy()
    x = 1
    from h2util import Q_MOD
    while pow((x ** 3 + 3) % Q_MOD, (Q_MOD - 1) // 2, Q_MOD) == 1:
        x += 1
    junk[123] = np.frombuffer(x.to_bytes(32, "little"), dtype=np.uint8)
    assert oracle.points_decompress(junk)[1] == 1
    with torch.cuda.stream(device.tstream):
        d_junk = torch.from_numpy(junk).to(device.dev)
    rc = device.L.h2_dev_points_decompress(d_junk.data_ptr(), len(pts), back.data_ptr(), device.stream)
    assert rc != 0, "an abscissa off the curve must be refused"


def test_device_prover_reproduces_committed_proofs(oracle, device):
    """the committed proof fixtures (tests/golden/proof_kat.json, made by the big-integer prover): same bytes from the
    device prover"""
    from h2util import load_golden
    from halo2_gpu_specific_amd import circuits, prover
    from halo2_gpu_specific_amd.rng import ProverRng
    from test_plonk_host import _golden_case

    product_cs = {"mini-plonk": circuits.mini_plonk, "rot-gate": rot_gate_cs, "lookup-shuffle": lookup_shuffle_cs}
    for case in load_golden("proof_kat.json"):
        assert int(case["trapdoor"], 16) == S_TRAPDOOR
        _, adv, fixed, copies, inst = _golden_case(case)
        params = srs(oracle, device, case["k"])
        pk = prover.keygen(device, params, product_cs[case["circuit"]](), cols_to_arr(fixed),
                           [(l[0], l[1], r[0], r[1]) for l, r in copies])
        assert pk.transcript_repr == int(case["vk_digest"], 16)
        proof = prover.create_proof_ext(device, params, pk, cols_to_arr(adv), ProverRng(case["seed"]),
                                        case["scheme"] == "gwc", instances=inst)
        assert proof.hex() == case["proof"], (case["circuit"], case["scheme"])


@pytest.mark.parametrize("k", [1, 3, 8, 13])
def test_device_unsafe_setup_matches_oracle(oracle, device, k):
    """Params::unsafe_setup on the device (powers by the prefix-product scan, l_i(s) by the elementwise kernels and the
    batch inversion, h2_dev_fixed_base_mul) against the oracle's restatement of poly/commitment.rs:56-124"""
    from halo2_gpu_specific_amd import prover

    n = 1 << k
    g = np.zeros((n, 8), dtype=np.uint64)
    gl = np.zeros((n, 8), dtype=np.uint64)
    s = fr_mont(S_TRAPDOOR)
    oracle.lib.oracle_unsafe_setup(k, s.ctypes.data, g.ctypes.data, gl.ctypes.data)
    params = prover.Params.unsafe_setup(device, k, S_TRAPDOOR)
    assert np.array_equal(device.download(params.g).reshape(n, 8), g)
    assert np.array_equal(device.download(params.g_lagrange).reshape(n, 8), gl)


def test_fixed_base_mul_edge_scalars(oracle, device):
    """scalars 0, 1, 2, r - 1 and random ones against the reference curve arithmetic"""
    from halo2_gpu_specific_amd import prover
    from h2util import arr_to_points

    params = prover.Params.unsafe_setup(device, 2, 5)          # g = [1, 5, 25, 125] G
    pts = arr_to_points(device.download(params.g).reshape(4, 8))
    assert pts == [rp.g1_mul(rp.G1, e) for e in (1, 5, 25, 125)]
    params = prover.Params.unsafe_setup(device, 1, rp.R - 1)   # g = [1, -1] G
    pts = arr_to_points(device.download(params.g).reshape(2, 8))
    assert pts == [rp.G1, rp.g1_neg(rp.G1)]


def test_proof_parity_over_random_trapdoors_sizes_and_seeds(oracle, device):
    """a sweep instead of fixed cases: other SRS trapdoors, sizes 4..9, blinding seeds, all three circuits, both
    multiopen schemes -- device bytes == reference bytes, and accepted"""
    import random

    from halo2_gpu_specific_amd import circuits, prover
    from halo2_gpu_specific_amd.rng import ProverRng

    rnd = random.Random(20261002)
    makers = {"mini": (rp.MiniPlonk, circuits.mini_plonk), "rot": (rp.RotGate, rot_gate_cs),
              "lookup": (rp.LookupShuffle, lookup_shuffle_cs)}
    for trial in range(8):
        which = rnd.choice(sorted(makers))
        ref_cs, make = makers[which]
        k = rnd.randrange(5 if which != "mini" else 4, 10 if which != "lookup" else 8)
        trapdoor, seed, use_gwc = rnd.randrange(2, rp.R), rnd.randrange(1 << 30), rnd.random() < 0.5
        syn = ref_cs.synthesize(k)
        adv, fixed, copies = syn[:3]
        inst = syn[3] if len(syn) > 3 else []
        params = prover.Params.unsafe_setup(device, k, trapdoor)
        pk = prover.keygen(device, params, make(), cols_to_arr(fixed), [(l[0], l[1], r[0], r[1]) for l, r in copies])
        rpk = rp.keygen(ref_cs, k, trapdoor, fixed, copies)
        assert pk.fixed_commitments == rpk.fixed_commitments and pk.perm_commitments == rpk.perm_commitments
        proof = prover.create_proof_ext(device, params, pk, cols_to_arr(adv), ProverRng(seed), use_gwc, instances=inst)
        want = rp.create_proof(rpk, adv, ProverRng(seed), use_gwc=use_gwc, instances=inst)
        assert proof == want, (trial, which, k, use_gwc)
        assert rp.verify_proof(rpk, proof, use_gwc=use_gwc, instances=inst)


def test_circuit_data_file_to_proving_key(oracle, device, tmp_path):
    """N4: CircuitData::write -> read -> into_proving_key (plonk.rs:126-204): a key rebuilt from the file (constraint
    system, raw fixed columns, permutation mapping) proves to the same bytes as the key it was written from -- and as
    the big-integer prover; a file whose columns disagree with its commitments is refused"""
    from halo2_gpu_specific_amd import formats, prover
    from halo2_gpu_specific_amd.rng import ProverRng

    k = 6
    adv, fixed, copies, inst = rp.LookupShuffle.synthesize(k)
    params = srs(oracle, device, k)
    cs = lookup_shuffle_cs()
    pk = prover.keygen(device, params, cs, cols_to_arr(fixed), [(l[0], l[1], r[0], r[1]) for l, r in copies])
    path = str(tmp_path / "circuit.data")
    formats.circuit_data_write(path, device, params, pk)
    info = formats.circuit_data_read(path, cs.name)
    assert (info["j"], info["k"]) == (cs.degree(), k)
    pk2 = prover.keygen_from_info(device, params, info)
    assert pk2.transcript_repr == pk.transcript_repr
    rpk = rp.keygen(rp.LookupShuffle, k, S_TRAPDOOR, fixed, copies)
    for seed, use_gwc in ((3, False), (4, True)):
        a = prover.create_proof_ext(device, params, pk, cols_to_arr(adv), ProverRng(seed), use_gwc, instances=inst)
        b = prover.create_proof_ext(device, params, pk2, cols_to_arr(adv), ProverRng(seed), use_gwc, instances=inst)
        assert a == b == rp.create_proof(rpk, adv, ProverRng(seed), use_gwc=use_gwc, instances=inst)
    # one fixed value changed in the file: the commitments no longer match
    raw = bytearray(open(path, "rb").read())
    n = 1 << k
    tail = len(info["mapping"][0]) * (4 + 8 * n) + 4           # mapping section
    first_fixed = len(raw) - tail - len(info["fixed"]) * (4 + 32 * n) + 4
    raw[first_fixed + 32 * 3] ^= 1
    open(path, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="do not match"):
        prover.keygen_from_info(device, params, formats.circuit_data_read(path, cs.name))
    with pytest.raises(ValueError, match="under params"):
        prover.keygen_from_info(device, srs(oracle, device, k + 1), info)
