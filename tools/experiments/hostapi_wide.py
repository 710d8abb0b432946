"""the wide circuit (64 advice columns, 8 lookups) through the host-slice entry points: python tools/experiments/hostapi_wide.py [k] [pinned|pageable]"""
import os
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

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
mode = sys.argv[2] if len(sys.argv) > 2 else "pinned"
D = prover.Device()
params = prover.Params.unsafe_setup(D, k, 0x1D0C5F0A3B7E91C2A4D6F8091B2C3D4E5F60718293A4B5C6D7E8F9010203)
cs = circuits.wide(16)
adv, fixed, copies = circuits.wide_synthesize(k, 16)
pk = prover.keygen(D, params, cs, fixed, copies)
want = prover.create_proof_with_shplonk(D, params, pk, adv, ProverRng(1))
D.sync()
t0 = time.perf_counter()
for _ in range(3):
    prover.create_proof_with_shplonk(D, params, pk, adv, ProverRng(1))
D.sync()
print("resident: %.3f s" % ((time.perf_counter() - t0) / 3))
H = host_api.HostApiDevice(pinned=(mode == "pinned"))
hparams = host_api.params_like(H, params)
t0 = time.perf_counter()
hpk = prover.keygen(H, hparams, cs, fixed, copies)
print("host-slice keygen: %.3f s" % (time.perf_counter() - t0))
got = prover.create_proof_with_shplonk(H, hparams, hpk, adv, ProverRng(1))
assert got == want, "bytes differ"
for i in range(3):
    ph = {} if i == 2 else None
    H.L.calls.clear()
    H.L.R.reset()
    t0 = time.perf_counter()
    prover.create_proof_with_shplonk(H, hparams, hpk, adv, ProverRng(1), timings=ph)
    dt = time.perf_counter() - t0
    print("host-slice (%s) proof %d: %.3f s, %.3f s inside %d library calls  %s" % (
        mode, i, dt, H.L.R.busy_seconds, sum(H.L.calls.values()), {n: round(v * 1e3, 1) for n, v in (ph or {}).items()}))
print(dict(sorted(H.L.calls.items())))
print({n: round(v * 1e3, 1) for n, v in sorted(H.L.R.by_call.items())})
