"""Proof bytes of the device prover == proof bytes of a CPU run (BASELINE configs[3]: "proof bytes == CPU"), on the same
SRS, witness and randomness: the CPU side is tests/oracle_prover.py -- the host orchestration of prover.py with every
kernel replaced by the C oracle's restatement of the reference's rayon loop (pinned on small circuits against the
independent big-integer prover: tests/test_oracle_prover.py).  Up to k = 22 (configs[3]) and k = 24 (configs[4]'s size):
every launch of a proof -- transforms, commitments, scans, the evaluator, the multiopen passes -- is checked in context
at the metric's sizes, not only kernel by kernel."""
import os
import time

import numpy as np
import pytest

import ref_plonk as rp
from test_plonk_host import S_TRAPDOOR, lookup_shuffle_cs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def device():
    from halo2_gpu_specific_amd import prover

    return prover.Device()


K24_SEED_INLINE = 22


def prove_both(device, cs, k, adv, fixed, copies, insts=(), modes=((1, False), (2, True)), cpu_kw=None, fixed_ints=False):
    """keygen + create_proof on the device and on the CPU; returns the timings of the last proof (device s, cpu s)"""
    import oracle_prover as op
    from halo2_gpu_specific_amd import prover
    from halo2_gpu_specific_amd.rng import ProverRng

    params = prover.Params.unsafe_setup(device, k, S_TRAPDOOR)
    pk = prover.keygen(device, params, cs, fixed, copies)
    cpu = op.OracleDevice(**(cpu_kw or {}))
    cparams = op.params_like(cpu, params)
    t0 = time.perf_counter()
    cpk = op.keygen(cpu, cparams, cs, fixed, copies)
    t_keygen = time.perf_counter() - t0
    assert pk.fixed_commitments == cpk.fixed_commitments and pk.perm_commitments == cpk.perm_commitments
    assert pk.transcript_repr == cpk.transcript_repr
    t_dev = t_cpu = 0.0
    for seed, use_gwc in modes:
        prover.create_proof_ext(device, params, pk, adv, ProverRng(seed), use_gwc, instances=insts)   # warm
        t0 = time.perf_counter()
        proof = prover.create_proof_ext(device, params, pk, adv, ProverRng(seed), use_gwc, instances=insts)
        t_dev = time.perf_counter() - t0
        t0 = time.perf_counter()
        want = prover.create_proof_ext(cpu, cparams, cpk, adv, ProverRng(seed), use_gwc, instances=insts)
        t_cpu = time.perf_counter() - t0
        first = next((i for i in range(min(len(proof), len(want))) if proof[i] != want[i]), None)
        assert first is None and len(proof) == len(want), \
            "device proof differs from the CPU proof at byte %s (field %s)" % (first, first and first // 32)
    print("k=%d: device %.3f s, cpu %.1f s (keygen %.1f s) on %d threads" % (k, t_dev, t_cpu, t_keygen, cpu.L.threads))
    return t_dev, t_cpu


@pytest.mark.parametrize("which,k", [("mini", 12), ("lookup", 10), ("wide", 12), ("wide16", 15), ("range", 17)])
def test_device_proof_bytes_equal_cpu_proof_bytes(device, which, k):
    from halo2_gpu_specific_amd import circuits
    from h2util import ints_to_arr

    insts = ()
    if which == "mini":
        cs = circuits.mini_plonk()
        adv, fixed, copies = circuits.mini_plonk_synthesize(k)
    elif which.startswith("wide"):
        quads = 16 if which == "wide16" else 2          # wide16: bench.py's create_proof_wide circuit (64 advice columns)
        cs = circuits.wide(quads)
        adv, fixed, copies = circuits.wide_synthesize(k, quads)
    elif which == "range":
        cs = circuits.range_check()
        adv, fixed, copies = circuits.range_check_synthesize(k)
    else:
        cs = lookup_shuffle_cs()
        adv, fixed, copies, insts = rp.LookupShuffle.synthesize(k)
        adv, fixed = [ints_to_arr(c) for c in adv], [ints_to_arr(c) for c in fixed]
        copies = [(l[0], l[1], r[0], r[1]) for l, r in copies]
    prove_both(device, cs, k, adv, fixed, copies, insts, modes=((1, False),) if which == "wide16" else ((1, False), (2, True)))
    # the CPU run by the coset route (coefficient forms only) gives the same bytes again
    if which in ("mini", "lookup"):
        prove_both(device, cs, k, adv, fixed, copies, insts, modes=((3, False),), cpu_kw={"eval_cache": 0})


def test_one_advice_column_proof_and_the_side_stream_extension(device):
    """with ONE advice column (or an extended domain above 2^23) Device.coeffs_to_extended falls back to
    the single-vector transform, which used to launch on the compute stream while the side stream was still producing its
    input.  A one-column circuit's proof (k <= 20: the side-stream route) against the CPU run, and the fallback itself on the
    side stream against the compute-stream result."""
    import torch

    from halo2_gpu_specific_amd import circuit as hc, prover

    cs = hc.ConstraintSystem("one-column")
    a, q = cs.advice_column(), cs.fixed_column()
    cs.enable_equality(a)
    qa = cs.query_advice(a)
    cs.create_gate("boolean", [cs.query_fixed(q) * (qa * qa - qa)])
    k = 12
    n = 1 << k
    rng = np.random.default_rng(5)
    adv = [np.zeros((n, 4), dtype=np.uint64)]
    adv[0][: n - 8, 0] = rng.integers(0, 2, n - 8)
    adv[0][1, 0] = adv[0][0, 0]
    fixed = [np.zeros((n, 4), dtype=np.uint64)]
    fixed[0][: n - 8, 0] = 1
    prove_both(device, cs, k, adv, fixed, [(0, 0, 0, 1)])
    dom = prover.Domain(14, 4)
    x = device.upload(rng.integers(0, 1 << 60, (dom.n, 4), dtype=np.int64).astype(np.uint64))      # below r: canonical
    want = device.coeffs_to_extended([x], dom)[0]
    device.sync()
    side = torch.cuda.Stream(device=device.dev)
    with torch.cuda.stream(side):
        import ctypes

        y = x.clone()                                       # produced on the side stream, as intt_on_side_stream's copies are
        got = device.coeffs_to_extended([y], dom, stream=ctypes.c_void_p(side.cuda_stream))[0]
    side.synchronize()
    assert torch.equal(got, want)


@pytest.mark.timeout(2400)
@pytest.mark.parametrize("k", [20, 22] + ([] if os.environ.get("H2_TEST_CPU_PROVER_K24") == "0" else [24]))
def test_full_size_proof_bytes_equal_cpu_proof_bytes(device, k, k24_cpu_worker):
    """configs[3] (k = 22) and configs[4]'s size (k = 24: ~3 minutes of CPU work, part of the default run;
    H2_TEST_CPU_PROVER_K24=0 skips it): the CPU side of k = 24 runs coset by coset (2 x 2^24 points instead of one
    2^26-point extended domain per column: the same bytes, a quarter of the host memory).  In a whole-suite run the CPU half
    of k = 24 has been running in a process of its own since the session began (tests/conftest.py, tests/cpu_prover_worker.py)
    and this case runs last: same SRS, witness, seed and code path, the minutes overlap with the tests in between."""
    from halo2_gpu_specific_amd import circuits

    adv, fixed, copies = circuits.mini_plonk_synthesize(k)
    if k == 24 and k24_cpu_worker is not None:
        import json

        from conftest import K24_SEED
        from halo2_gpu_specific_amd import prover
        from halo2_gpu_specific_amd.rng import ProverRng

        params = prover.Params.unsafe_setup(device, k, S_TRAPDOOR)
        pk = prover.keygen(device, params, circuits.mini_plonk(), fixed, copies)
        prover.create_proof_ext(device, params, pk, adv, ProverRng(K24_SEED), False)   # warm
        t0 = time.perf_counter()
        proof = prover.create_proof_ext(device, params, pk, adv, ProverRng(K24_SEED), False)
        t_dev = time.perf_counter() - t0
        w = k24_cpu_worker
        try:
            rc = w["proc"].wait(timeout=1800)
        finally:
            tail = open(w["log"]).read()[-3000:]
        assert rc == 0, "the CPU worker failed:\n" + tail
        doc = json.load(open(w["out"]))
        assert (doc["k"], doc["seed"], doc["use_gwc"]) == (24, K24_SEED, False)
        assert repr(pk.fixed_commitments) == doc["fixed_commitments"] and repr(pk.perm_commitments) == doc["perm_commitments"]
        assert repr(pk.transcript_repr) == doc["transcript_repr"]
        want = bytes.fromhex(doc["proof"])
        first = next((i for i in range(min(len(proof), len(want))) if proof[i] != want[i]), None)
        assert first is None and len(proof) == len(want), \
            "device proof differs from the CPU proof at byte %s (field %s)" % (first, first and first // 32)
        print("k=24: device %.3f s, cpu %.1f s (keygen %.1f s) on %d threads, in a process of its own"
              % (t_dev, doc["cpu_seconds"], doc["keygen_seconds"], doc["threads"]))
        return
    prove_both(device, circuits.mini_plonk(), k, adv, fixed, copies, modes=((K24_SEED_INLINE, False),),
               cpu_kw={"eval_cache": 0} if k >= 24 else None)


def test_two_circuit_instances_and_budgeted_device_equal_cpu(device):
    """several circuit instances in one proof (plonk/prover.rs:206-232) and the memory-budgeted route of the device (coset by
    coset, tables evicted) against the CPU run by the extended domain"""
    import oracle_prover as op
    from halo2_gpu_specific_amd import circuits, prover
    from halo2_gpu_specific_amd.rng import ProverRng

    k = 11
    cs = circuits.mini_plonk()
    adv_a, fixed, copies = circuits.mini_plonk_synthesize(k, a=5)
    adv_b = circuits.mini_plonk_synthesize(k, a=9)[0]
    params = prover.Params.unsafe_setup(device, k, S_TRAPDOOR)
    tight = prover.Device(eval_cache=1)
    tparams = prover.Params(tight, k, params.g, params.g_lagrange)
    cpu = op.OracleDevice()
    cparams = op.params_like(cpu, params)
    cpk = op.keygen(cpu, cparams, cs, fixed, copies)
    want = prover.create_proof_ext(cpu, cparams, cpk, [adv_a, adv_b], ProverRng(4), False, instances=[(), ()])
    for D, pr in ((device, params), (tight, tparams)):
        pk = prover.keygen(D, pr, cs, fixed, copies)
        assert prover.create_proof_ext(D, pr, pk, [adv_a, adv_b], ProverRng(4), False, instances=[(), ()]) == want


def test_compact_and_resident_witness_columns_give_the_same_bytes(device):
    """a witness column as 8-byte cells (h2_dev_widen_u64) or as a tensor already on the device: the bytes of the proof
    made from 32-byte host cells -- for the wide circuit (pinned and pageable sources) and for the range-check argument,
    whose completion (planted range values, sorted companion) then runs on the compact host columns"""
    import torch

    from halo2_gpu_specific_amd import circuits, prover
    from halo2_gpu_specific_amd.rng import ProverRng

    k = 12
    cs = circuits.wide(2)
    adv, fixed, copies = circuits.wide_synthesize(k, 2)
    params = prover.Params.unsafe_setup(device, k, S_TRAPDOOR)
    pk = prover.keygen(device, params, cs, fixed, copies)
    want = prover.create_proof_ext(device, params, pk, adv, ProverRng(9), False)
    for alloc in (None, device.pinned_columns):
        small = circuits.wide_synthesize(k, 2, alloc=alloc, compact=True)[0]
        assert all(c.ndim == 1 and np.array_equal(c, a[:, 0]) for c, a in zip(small, adv))
        assert prover.create_proof_ext(device, params, pk, small, ProverRng(9), False) == want
    resident = [device.upload(a) for a in adv]
    before = [t.clone() for t in resident]
    assert prover.create_proof_ext(device, params, pk, resident, ProverRng(9), False) == want
    assert all(torch.equal(a, b) for a, b in zip(resident, before)), "the caller's resident columns were modified"
    # mixed forms in one witness
    mixed = [small[i] if i % 3 == 0 else (resident[i] if i % 3 == 1 else adv[i]) for i in range(len(adv))]
    gwc = prover.create_proof_ext(device, params, pk, adv, ProverRng(9), True)
    assert prover.create_proof_ext(device, params, pk, mixed, ProverRng(9), True) == gwc
    # the widening on its own: every value, a ragged length
    src = torch.randint(-(2**63), 2**63 - 1, (100003,), dtype=torch.int64, device=device.dev)
    torch.cuda.synchronize()                       # generated on torch's stream, consumed on the device's own
    wide = device.widen(src)
    device.sync()
    assert torch.equal(wide[:, 0], src) and not wide[:, 1:].any()
    # range check
    k = 17
    cs = circuits.range_check()
    adv, fixed, copies = circuits.range_check_synthesize(k)
    params = prover.Params.unsafe_setup(device, k, S_TRAPDOOR)
    pk = prover.keygen(device, params, cs, fixed, copies)
    small = [a[:, 0].copy() for a in adv]
    want = prover.create_proof_ext(device, params, pk, adv, ProverRng(3), False)
    assert prover.create_proof_ext(device, params, pk, small, ProverRng(3), False) == want


def test_random_circuits_device_bytes_equal_cpu_bytes(device):
    """tools/prover_fuzz.py: circuits drawn from seeds (random gate expressions with rotations, constants and scalings,
    random equality columns and copies, optional logup lookup with one or two input sets, optional shuffle, optional
    instance column) -- the generated gate kernel of each and every phase around it against the CPU prover's interpreter
    and loops; a longer run of the same generator: `python tools/prover_fuzz.py 600`"""
    import sys

    from h2util import ROOT

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import prover_fuzz

    # the fixed part of the corpus: the reference's live examples (simple-example-2, range-check, lookup_api, lookup_api_set,
    # shuffle, shuffle_api, shuffle_api_group) with their own witnesses, by the extended route and both coset routes
    assert len(prover_fuzz.run_examples(device)) == 7
    for seed in range(1, 17):
        prover_fuzz.run_case(device, seed)
    # satisfied circuits (every gate sel (E - d) with d set to E, copies that hold): the quotient is a polynomial, so the
    # coset routes of the device -- all cosets at once, and coset by coset with no table set retained -- give the same bytes
    for seed in range(101, 107):
        prover_fuzz.run_case(device, seed, satisfiable=True)
