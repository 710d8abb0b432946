"""The literal drop-in's data flow (halo2-gpu-specific_amd/host_api.py): every polynomial in host memory, every vector
operation one host-slice entry point of include/halo2_hip.h -- the calls `integration/hip.rs` binds -- including the cuda
shape of the evaluator (coefficient forms in, one h2_evaluate_h_coeff call).  Same SRS, witness and randomness as the
device-resident prover: the proof bytes must be equal (mini-PLONK, the lookup / shuffle / instance circuit, the wide
circuit; GWC and SHPLONK; two circuit instances)."""
import numpy as np
import pytest

import ref_plonk as rp
from test_gpu_plonk import cols_to_arr, srs
from test_plonk_host import lookup_shuffle_cs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def device():
    from halo2_gpu_specific_amd import prover

    return prover.Device()


@pytest.mark.parametrize("which,k", [("mini", 9), ("lookup", 8), ("wide", 9), ("mini", 14)])
def test_host_slice_api_gives_the_resident_provers_bytes(oracle, device, which, k):
    from halo2_gpu_specific_amd import circuits, host_api, prover
    from halo2_gpu_specific_amd.rng import ProverRng

    inst = ()
    if which == "mini":
        cs, (adv, fixed, copies) = circuits.mini_plonk(), circuits.mini_plonk_synthesize(k)
    elif which == "wide":
        cs, (adv, fixed, copies) = circuits.wide(4), circuits.wide_synthesize(k, 4)
    else:
        cs = lookup_shuffle_cs()
        syn = rp.LookupShuffle.synthesize(k)
        adv, fixed = cols_to_arr(syn[0]), cols_to_arr(syn[1])
        copies, inst = [(l[0], l[1], r[0], r[1]) for l, r in syn[2]], syn[3]
    params = srs(oracle, device, k)
    pk = prover.keygen(device, params, cs, fixed, copies)
    H = host_api.HostApiDevice()
    hparams = host_api.params_like(H, params)
    hpk = prover.keygen(H, hparams, cs, fixed, copies)
    assert hpk.transcript_repr == pk.transcript_repr and hpk.fixed_commitments == pk.fixed_commitments
    assert hpk.perm_commitments == pk.perm_commitments
    for seed, gwc in ((1, False), (2, True)):
        want = prover.create_proof_ext(device, params, pk, adv, ProverRng(seed), gwc, instances=inst)
        got = prover.create_proof_ext(H, hparams, hpk, adv, ProverRng(seed), gwc, instances=inst)
        first = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), None)
        assert len(got) == len(want) and first is None, "host-slice proof differs at byte %s" % first
    calls = H.L.calls
    assert calls["h2_evaluate_h_coeff"] == 2 and calls["h2_msm"] > 10 and calls["h2_intt"] > 3
    assert not any(name.startswith("oracle") for name in calls)
    if which == "mini" and k == 9:          # two circuit instances in one proof
        adv2 = circuits.mini_plonk_synthesize(k, a=9)[0]
        want = prover.create_proof_ext(device, params, pk, [adv, adv2], ProverRng(5), False, instances=[(), ()])
        assert prover.create_proof_ext(H, hparams, hpk, [adv, adv2], ProverRng(5), False, instances=[(), ()]) == want


def test_host_slice_device_needs_a_gpu_and_touches_no_oracle():
    import inspect

    from halo2_gpu_specific_amd import host_api

    src = inspect.getsource(host_api)
    assert "oracle" not in src.replace("no oracle", "") and "liboracle" not in src
