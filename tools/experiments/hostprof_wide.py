"""cProfile of create_proof on the wide circuit (host-side costs between kernels).  usage: hostprof_wide.py [k] [quads]"""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import os  # noqa: E402

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before HIP initialises (halo2-gpu-specific_amd/__init__.py says why)
import torch; torch.cuda.init()
from halo2_gpu_specific_amd import circuits, prover
from halo2_gpu_specific_amd.rng import ProverRng
k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
quads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
D = prover.Device()
params = prover.Params.unsafe_setup(D, k, 0x1D0C5F0A3B7E91C2A4D6F8091B2C3D4E5F60718293A4B5C6D7E8F9010203)
cs = circuits.wide(quads)
adv, fixed, copies = circuits.wide_synthesize(k, quads, alloc=D.pinned_columns, compact=True)
pk = prover.keygen(D, params, cs, fixed, copies)
for rep in range(2): prover.create_proof_with_shplonk(D, params, pk, adv, ProverRng(rep))
pr = cProfile.Profile(); pr.enable()
for rep in range(3): prover.create_proof_with_shplonk(D, params, pk, adv, ProverRng(rep))
D.sync(); pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
st.sort_stats("cumulative").print_stats(45)
