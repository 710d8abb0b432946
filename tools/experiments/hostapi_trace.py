"""host-side timeline of the literal drop-in's k = 22 proof (H2_PROVER_HOST_TRACE): where the wall time of each phase goes, call
by call -- python tools/experiments/hostapi_trace.py [pinned|pageable]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# (as bench.py: OpenMP teams -- torch's CPU operators here -- sleep instead of spinning; a spinning team of 256 on a box whose cgroup
# grants 16 CPUs gets the whole process throttled, and every number below with it)
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
os.environ.setdefault("GOMP_SPINCOUNT", "0")
import numpy as np  # noqa: E402

from halo2_gpu_specific_amd import circuits, host_api, prover  # noqa: E402
from halo2_gpu_specific_amd.rng import ProverRng  # noqa: E402

k = 22
mode = sys.argv[1] if len(sys.argv) > 1 else "pinned"
D = prover.Device()
params = prover.Params.unsafe_setup(D, k, 0x1D0C5F0A3B7E91C2A4D6F8091B2C3D4E5F60718293A4B5C6D7E8F9010203)
adv, fixed, copies = circuits.mini_plonk_synthesize(k)
H = host_api.HostApiDevice(pinned=(mode == "pinned"))
hparams = host_api.params_like(H, params)
hpk = prover.keygen(H, hparams, circuits.mini_plonk(), fixed, copies)
prover.create_proof_with_shplonk(H, hparams, hpk, adv, ProverRng(1))
pr = cProfile.Profile()
pr.enable()
prover.create_proof_with_shplonk(H, hparams, hpk, adv, ProverRng(1))
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
