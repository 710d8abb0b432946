"""cProfile of create_proof on the lookup circuit (host-side costs).  usage: python tools/experiments/hostprof.py [k]"""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import os  # noqa: E402

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before HIP initialises (halo2-gpu-specific_amd/__init__.py says why)
import torch; torch.cuda.init()
import numpy as np
import ref_plonk as rp
from halo2_gpu_specific_amd import prover
from halo2_gpu_specific_amd.rng import ProverRng
from test_plonk_host import lookup_shuffle_cs
k = int(sys.argv[1]) if len(sys.argv) > 1 else 18
n = 1 << k
adv, fixed, copies, inst = rp.LookupShuffle.synthesize(k)
def arr(col):
    a = np.zeros((n, 4), dtype=np.uint64)
    for limb in range(4):
        a[:, limb] = np.array([(v >> (64 * limb)) & (2**64 - 1) for v in col], dtype=np.uint64)
    return a
adv, fixed = [arr(c) for c in adv], [arr(c) for c in fixed]
D = prover.Device()
params = prover.Params.synthetic(D, k)
pinned = D.pinned_columns(len(adv), n)
for dst, src in zip(pinned, adv): dst[:] = src
pk = prover.keygen(D, params, lookup_shuffle_cs(), fixed, [(l[0], l[1], r[0], r[1]) for l, r in copies])
for rep in range(2): prover.create_proof_with_shplonk(D, params, pk, pinned, ProverRng(rep), instances=inst)
pr = cProfile.Profile(); pr.enable()
for rep in range(3): prover.create_proof_with_shplonk(D, params, pk, pinned, ProverRng(rep), instances=inst)
D.sync(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
