"""create_proof per phase on the lookup + shuffle + instance test circuit (tests/ref_plonk.LookupShuffle / the
product mirror in tests/test_plonk_host.py), synthetic SRS.   usage: python tools/lookup_bench.py [k]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before HIP initialises (halo2-gpu-specific_amd/__init__.py says why)
import torch  # noqa: E402

torch.cuda.init()
import numpy as np  # noqa: E402

import ref_plonk as rp  # noqa: E402
from halo2_gpu_specific_amd import prover  # noqa: E402
from halo2_gpu_specific_amd.rng import ProverRng  # noqa: E402
from test_plonk_host import lookup_shuffle_cs  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = 1 << k
t0 = time.perf_counter()
adv, fixed, copies, inst = rp.LookupShuffle.synthesize(k)


def arr(col):
    a = np.zeros((n, 4), dtype=np.uint64)
    for limb in range(4):
        a[:, limb] = np.array([(v >> (64 * limb)) & (2**64 - 1) for v in col], dtype=np.uint64)
    return a


adv, fixed = [arr(c) for c in adv], [arr(c) for c in fixed]
print("synthesize %.1f s" % (time.perf_counter() - t0))
D = prover.Device()
params = prover.Params.synthetic(D, k)
pinned = D.pinned_columns(len(adv), n)          # the witness in page-locked memory, as a caller that cares about PCIe keeps it
for dst, src in zip(pinned, adv):
    dst[:] = src
adv = pinned
t0 = time.perf_counter()
pk = prover.keygen(D, params, lookup_shuffle_cs(), fixed, [(l[0], l[1], r[0], r[1]) for l, r in copies])
print("keygen %.3f s" % (time.perf_counter() - t0))
for rep in range(3):
    timings = {}
    ta = time.perf_counter()
    proof = prover.create_proof_with_shplonk(D, params, pk, adv, ProverRng(rep), timings=timings if rep else None, instances=inst)
    D.sync()
    print("rep %d: %.1f ms (%d bytes) %s" % (rep, (time.perf_counter() - ta) * 1e3, len(proof),
                                            {a: round(b * 1e3, 1) for a, b in timings.items()}))
