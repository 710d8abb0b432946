"""the literal drop-in's k = 22 proof from page-locked vectors, five times in a row with per-phase timings (and with a host
profile of each): does any proof of the sequence differ from the others?  python tools/experiments/hostapi_phases_repeat.py"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# (as bench.py: OpenMP teams -- torch's CPU operators here -- sleep instead of spinning; a spinning team of 256 on a box whose cgroup
# grants 16 CPUs gets the whole process throttled, and every number below with it)
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
os.environ.setdefault("GOMP_SPINCOUNT", "0")
from halo2_gpu_specific_amd import circuits, host_api, prover  # noqa: E402
from halo2_gpu_specific_amd.rng import ProverRng  # noqa: E402

k = 22
D = prover.Device()
params = prover.Params.unsafe_setup(D, k, 0x1D0C5F0A3B7E91C2A4D6F8091B2C3D4E5F60718293A4B5C6D7E8F9010203)
adv, fixed, copies = circuits.mini_plonk_synthesize(k)
H = host_api.HostApiDevice(pinned=True)
hparams = host_api.params_like(H, params)
hpk = prover.keygen(H, hparams, circuits.mini_plonk(), fixed, copies)
for i in range(6):
    ph = {} if i % 2 else None
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    prover.create_proof_with_shplonk(H, hparams, hpk, adv, ProverRng(1), timings=ph)
    pr.disable()
    print("proof %d: %.3f s  %s" % (i, time.perf_counter() - t0, {n: round(v * 1e3, 1) for n, v in (ph or {}).items()}))
    if time.perf_counter() - t0 > 0.3:
        pstats.Stats(pr).sort_stats("tottime").print_stats(8)
