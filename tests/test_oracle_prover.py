"""The CPU prover of tests/oracle_prover.py (the host orchestration of prover.py over the C oracle's loops) against the
independent big-integer prover of tests/ref_plonk.py: proof bytes equal.  This pins the prover-level passes added to
oracle/oracle.c (prefix scans, permutation terms / sigma, logup multiplicities, linear combinations, the replayable random
polynomial) -- the CPU reference the device prover is compared with at k = 20 .. 24 on the GPU box."""
import numpy as np
import pytest

import ref_plonk as rp
from h2util import fr_mont, ints_to_arr
from test_plonk_host import S_TRAPDOOR, lookup_shuffle_cs, rot_gate_cs


def oracle_params(oracle, device, k):
    from halo2_gpu_specific_amd import prover

    n = 1 << k
    g = np.zeros((n, 8), dtype=np.uint64)
    gl = np.zeros((n, 8), dtype=np.uint64)
    s = fr_mont(S_TRAPDOOR)
    oracle.lib.oracle_unsafe_setup(k, s.ctypes.data, g.ctypes.data, gl.ctypes.data)
    return prover.Params(device, k, g, gl, tables=False)


def cols_to_arr(cols):
    return [ints_to_arr(c) for c in cols]


@pytest.mark.parametrize("which,k", [("mini", 5), ("mini", 8), ("rot", 6), ("lookup", 6)])   # ("mini", 8): BASELINE configs[0], literally
def test_cpu_prover_bytes_match_big_integer_prover(oracle, which, k):
    import oracle_prover as op
    from halo2_gpu_specific_amd import circuits, prover
    from halo2_gpu_specific_amd.rng import ProverRng

    insts = ()
    if which == "lookup":
        ref_cs, cs = rp.LookupShuffle, lookup_shuffle_cs()
        adv, fixed, copies, insts = ref_cs.synthesize(k)
    else:
        ref_cs = rp.MiniPlonk if which == "mini" else rp.RotGate
        cs = circuits.mini_plonk() if which == "mini" else rot_gate_cs()
        adv, fixed, copies = ref_cs.synthesize(k)
    rpk = rp.keygen(ref_cs, k, S_TRAPDOOR, fixed, copies)
    for kw in ({}, {"force_cosets": True}):
        D = op.OracleDevice(threads=2, **kw)
        params = oracle_params(oracle, D, k)
        pk = op.keygen(D, params, cs, cols_to_arr(fixed), [(l[0], l[1], r[0], r[1]) for l, r in copies])
        assert pk.fixed_commitments == rpk.fixed_commitments
        assert pk.perm_commitments == rpk.perm_commitments
        assert pk.transcript_repr == rpk.transcript_repr
        for seed, use_gwc in ((1, False), (3, True)):
            proof = prover.create_proof_ext(D, params, pk, cols_to_arr(adv), ProverRng(seed), use_gwc, instances=insts)
            want = rp.create_proof(rpk, adv, ProverRng(seed), use_gwc=use_gwc, instances=insts)
            first = next((i for i in range(min(len(proof), len(want))) if proof[i] != want[i]), None)
            assert first is None and len(proof) == len(want), "differs at byte %s (field %s)" % (first, first and first // 32)


def test_advice_uploads_are_queued_a_few_groups_ahead(oracle, monkeypatch):
    """round 5: with every column of a wide witness in the copy queue at once, the first commitment group returned only when
    the last column had crossed PCIe -- the columns are queued `H2_ADVICE_AHEAD` groups ahead of the group being committed.
    Here on the CPU device: the ORDER of the upload requests relative to the group commitments, and the same proof bytes
    whatever the group size."""
    import oracle_prover as op
    from halo2_gpu_specific_amd import circuits, prover
    from halo2_gpu_specific_amd.rng import ProverRng

    k = 6
    cs = circuits.wide(4)                                   # 16 advice columns
    adv, fixed, copies = circuits.wide_synthesize(k, 4)
    events = []

    class Logged(op.OracleDevice):
        def upload(self, a):
            if getattr(a, "shape", None) == (1 << k, 4):
                events.append("U")
            return super().upload(a)

        def msm_partial(self, columns, bases, lo, hi, max_bits=254, also=None):
            if len(columns) >= 2:
                events.append("C%d" % len(columns))
            return super().msm_partial(columns, bases, lo, hi, max_bits, also)

    D = Logged(threads=2)
    params = oracle_params(oracle, D, k)
    pk = op.keygen(D, params, cs, fixed, copies)
    proofs = []
    for group, ahead in (("4", "1"), ("16", "2"), ("3", "2")):
        monkeypatch.setenv("H2_ADVICE_GROUP", group)
        monkeypatch.setenv("H2_ADVICE_AHEAD", ahead)
        del events[:]
        proofs.append(prover.create_proof_ext(D, params, pk, adv, ProverRng(5), False))
        if group == "4":
            # uploads seen before the 1st .. 4th group commitment: group + ahead, then one more group each time, never more than 16
            seen, at = [], 0
            for e in events:
                if e == "U":
                    at += 1
                elif len(seen) < 4:
                    seen.append(at)
            assert seen == [8, 12, 16, 16], (seen, events[:40])
        elif group == "16":
            assert events[:16] == ["U"] * 16                # one group: everything goes up first
    assert proofs[0] == proofs[1] == proofs[2]
    # round 6: a witness that arrives in several groups has each group's columns taken to coefficient form (and to the extended
    # domain) right behind the group's commitment -- on the side stream of a HIP device, under the later groups' transfers -- instead
    # of after the whole phase: same bytes
    calls = []
    orig = D.intt_on_side_stream
    D.intt_on_side_stream = lambda cols, dom, extend=False: (calls.append(len(cols)), orig(cols, dom, extend))[1]
    monkeypatch.setenv("H2_ADVICE_GROUP", "4")
    monkeypatch.setenv("H2_SIDE_GROUPS", "1")
    assert prover.create_proof_ext(D, params, pk, adv, ProverRng(5), False) == proofs[0]
    assert calls == [4, 4, 4, 4], calls
    monkeypatch.setenv("H2_SIDE_GROUPS", "0")
    del calls[:]
    assert prover.create_proof_ext(D, params, pk, adv, ProverRng(5), False) == proofs[0] and calls == [16]


def test_product_device_still_needs_a_gpu():
    """the oracle device is injected by tests only: the product's own Device has no CPU path"""
    import torch

    from halo2_gpu_specific_amd import prover

    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    with pytest.raises(RuntimeError, match="no CPU path"):
        prover.Device()
