"""Memory-bounded proving and the wide circuit (the reference's counterpart is the
coefficient-form proving key + extended-FFT cache of plonk/evaluation_gpu.rs:335-468, HALO2_PROOF_GPU_EVAL_CACHE):

* `circuits.wide` -- many advice columns, degree 5, logup range lookups -- proves to the bytes of its big-integer twin
  (ref_plonk.wide_class) and is accepted by its verifier;
* the same proof bytes whatever the residency: every extended coset resident, coset by coset under H2_DEVICE_MEM_BUDGET
  with proving-key tables evicted and rebuilt, and with the reference's HALO2_PROOF_GPU_EVAL_CACHE knob;
* the library's own tables (NTT plans, last-pass twiddle tables) obey their budget, are evicted least-recently-used first
  and can be released; transforms give the oracle's values before, between and after."""
import numpy as np
import pytest

import halo2_gpu_specific_amd as h2
import ref_plonk as rp
from h2util import R_MOD, fr_mont
from test_gpu_plonk import cols_to_arr, srs
from test_plonk_host import S_TRAPDOOR, lookup_shuffle_cs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def device():
    from halo2_gpu_specific_amd import prover

    return prover.Device()


@pytest.mark.parametrize("quads,k", [(2, 6), (4, 8)])
def test_wide_circuit_matches_big_integer_prover(oracle, device, quads, k):
    from halo2_gpu_specific_amd import circuits, prover
    from halo2_gpu_specific_amd.rng import ProverRng

    W = rp.wide_class(quads)
    adv, fixed, copies = circuits.wide_synthesize(k, quads)
    radv, rfixed, rcopies = W.synthesize(k)
    params = srs(oracle, device, k)
    pk = prover.keygen(device, params, circuits.wide(quads), fixed, copies)
    rpk = rp.keygen(W, k, S_TRAPDOOR, rfixed, rcopies)
    assert pk.transcript_repr == rpk.transcript_repr
    for seed, use_gwc in ((1, False), (2, True)):
        proof = prover.create_proof_ext(device, params, pk, adv, ProverRng(seed), use_gwc)
        want = rp.create_proof(rpk, radv, ProverRng(seed), use_gwc=use_gwc)
        first = next((i for i in range(min(len(proof), len(want))) if proof[i] != want[i]), None)
        assert len(proof) == len(want) and first is None, "differs from the big-integer prover at byte %s" % first
        assert rp.verify_proof(rpk, proof, use_gwc=use_gwc)
    bad = [c.copy() for c in adv]
    bad[3][2, 0] += np.uint64(1)                                  # d != a b c on one row
    assert not rp.verify_proof(rpk, prover.create_proof_ext(device, params, pk, bad, ProverRng(1), False))


@pytest.mark.parametrize("which,k", [("wide", 9), ("lookup", 8), ("mini", 10)])
def test_same_proof_bytes_under_a_memory_budget(oracle, device, which, k):
    """extended cosets resident (the default on 288 GB) == coset by coset from coefficient forms with the proving key's
    coset tables evicted and rebuilt (budget just above the coset-mode footprint; keep = 0 and 1) == the reference's
    HALO2_PROOF_GPU_EVAL_CACHE knob"""
    from halo2_gpu_specific_amd import circuits, prover
    from halo2_gpu_specific_amd.rng import ProverRng

    inst = []
    if which == "wide":
        cs_make = lambda: circuits.wide(4)  # noqa: E731
        adv, fixed, copies = circuits.wide_synthesize(k, 4)
    elif which == "mini":
        cs_make = circuits.mini_plonk
        adv, fixed, copies = circuits.mini_plonk_synthesize(k)
    else:
        cs_make = lookup_shuffle_cs
        syn = rp.LookupShuffle.synthesize(k)
        adv, fixed = cols_to_arr(syn[0]), cols_to_arr(syn[1])
        copies, inst = [(l[0], l[1], r[0], r[1]) for l, r in syn[2]], syn[3]
    params = srs(oracle, device, k)
    pk = prover.keygen(device, params, cs_make(), fixed, copies)
    assert pk.residency == "extended" and pk.coset is None
    want = [prover.create_proof_ext(device, params, pk, adv, ProverRng(s), gwc, instances=inst) for s, gwc in ((1, False), (2, True))]
    dom = pk.domain
    c = dom.quotient_poly_degree
    fp = prover.footprint(pk.cs, dom, 1)
    assert fp["cosets"] < fp["extended"]
    for kwargs, keep in ((dict(mem_budget=fp["cosets"]), 1), (dict(mem_budget=fp["cosets"] - 32 * dom.n), 0),
                         (dict(eval_cache=1), 1), (dict(eval_cache=64), c)):
        D2 = prover.Device(**kwargs)
        assert D2.residency(pk.cs, dom) == ("cosets", keep)
        params2 = prover.Params(D2, k, params.g, params.g_lagrange)
        pk2 = prover.keygen(D2, params2, cs_make(), fixed, copies)
        assert pk2.residency == "cosets" and pk2.fixed_cosets is None and pk2.l0 is None
        assert pk2.transcript_repr == pk.transcript_repr
        got = [prover.create_proof_ext(D2, params2, pk2, adv, ProverRng(s), gwc, instances=inst) for s, gwc in ((1, False), (2, True))]
        assert got == want, kwargs
        # two proofs over c cosets: with fewer than c table sets retained every visit is a rebuild
        assert pk2.coset.misses == (2 * c if keep < c else c) and len(pk2.coset.tabs) == min(keep, c), (pk2.coset.misses, keep)
        assert h2.lib().h2_release_plans() == 0                    # the library's plans go; the next proof rebuilds them
        assert prover.create_proof_ext(D2, params2, pk2, adv, ProverRng(1), False, instances=inst) == want[0]
    with pytest.raises(MemoryError):
        prover.Device(mem_budget=fp["cosets"] // 4).residency(pk.cs, dom)


def test_several_instances_switch_to_the_coset_route_when_they_do_not_fit(oracle, device):
    """residency is decided at keygen for ONE circuit instance; a proof of several instances whose polynomials do not fit
    the budget runs coset by coset from tables built on demand -- the same bytes as the unbudgeted device"""
    from halo2_gpu_specific_amd import circuits, prover
    from halo2_gpu_specific_amd.rng import ProverRng

    k, cs = 10, circuits.mini_plonk()
    advs = [circuits.mini_plonk_synthesize(k, a=a)[0] for a in (5, 9, 11)]
    _, fixed, copies = circuits.mini_plonk_synthesize(k)
    params = srs(oracle, device, k)
    pk = prover.keygen(device, params, cs, fixed, copies)
    want = prover.create_proof_ext(device, params, pk, advs, ProverRng(4), False, instances=[(), (), ()])
    dom = pk.domain
    one, three = prover.footprint(cs, dom, None, 1), prover.footprint(cs, dom, None, 3)
    assert three["extended"] > one["extended"] and three["cosets"] > one["cosets"]
    D2 = prover.Device(mem_budget=one["extended"])
    assert D2.residency(cs, dom) == ("extended", None) and D2.residency(cs, dom, 3)[0] == "cosets"
    params2 = prover.Params(D2, k, params.g, params.g_lagrange)
    pk2 = prover.keygen(D2, params2, cs, fixed, copies)
    assert pk2.residency == "extended" and pk2.coset is None
    assert prover.create_proof_ext(D2, params2, pk2, advs, ProverRng(4), False, instances=[(), (), ()]) == want
    assert pk2._multi_coset.misses >= 1                      # the coset route ran
    # one instance on the same key still takes the extended route
    single = prover.create_proof_ext(device, params, pk, advs[0], ProverRng(5), False)
    assert prover.create_proof_ext(D2, params2, pk2, advs[0], ProverRng(5), False) == single


def test_library_tables_budget_lru_and_release(oracle):
    """last-pass twiddle tables (32 B x n each, transforms of >= 2^18 points): inside the budget, least recently used out
    first -- but only for a key that has missed twice (no eviction to build a table that may be used once: a small budget
    under alternating transforms would rebuild a table per call) --, none with a zero budget, all gone after
    h2_release_plans -- and the same transform values throughout"""
    from halo2_gpu_specific_amd import arithmetic as ar

    L = h2.lib()
    root = 0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C
    sizes = (18, 19)
    xs = {ln: oracle.random_fr(900 + ln, 1 << ln) for ln in sizes}
    ws = {ln: fr_mont(pow(root, 1 << (28 - ln), R_MOD)) for ln in sizes}
    want = {ln: oracle.best_fft(xs[ln], ws[ln], ln, threads=8) for ln in sizes}

    def run(ln):
        assert np.array_equal(ar.best_fft(xs[ln].copy(), ws[ln], ln), want[ln]), ln

    tab = lambda ln: 32 << ln  # noqa: E731
    try:
        L.h2_set_table_budget(0)
        run(19)                                                       # grows the host-API staging arenas to their final size
        assert L.h2_release_plans() == 0
        base = L.h2_library_memory_bytes()
        assert L.h2_set_table_budget(0) == 0
        run(18)
        small = L.h2_library_memory_bytes() - base                   # the plan's own tables, no last-pass table
        assert 0 < small < tab(18) // 2
        assert L.h2_release_plans() == 0 and L.h2_library_memory_bytes() == base
        L.h2_set_table_budget(tab(19))                                # room for either table, not for both
        run(18)
        assert L.h2_library_memory_bytes() - base >= tab(18)
        run(19)                                # first miss of a key that would have to evict: it composes its twiddles instead
        held = L.h2_library_memory_bytes() - base
        assert tab(18) <= held < tab(19)
        run(19)                                # second miss: now the 2^18 table leaves (least recently used) and 2^19 is built
        held = L.h2_library_memory_bytes() - base
        assert tab(19) <= held < tab(19) + tab(18)
        run(18)                                # ... the same the other way round: one miss changes nothing,
        assert tab(19) <= L.h2_library_memory_bytes() - base < tab(19) + tab(18)
        run(18)                                # the second brings the table back
        held = L.h2_library_memory_bytes() - base
        assert tab(18) <= held < tab(19)
        L.h2_set_table_budget(tab(19) + tab(18))
        run(19)
        run(18)
        assert L.h2_library_memory_bytes() - base >= tab(19) + tab(18)
        assert L.h2_release_plans() == 0 and L.h2_library_memory_bytes() == base
        run(19)
    finally:
        L.h2_set_table_budget(2**64 - 1)                              # back to the default (H2_NTT_TABLE_BUDGET / 1/32 of memory)
        L.h2_release_plans()
