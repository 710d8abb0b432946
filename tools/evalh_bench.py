"""Throughput of the fused evaluate_h interpreter on a wide synthetic gate set (no permutation / lookups):
A advice + F fixed columns, G gates of the form  q * (a_i * a_j + a_k(rot) - a_l) * (a_m + c).
usage: python tools/evalh_bench.py [k] [A] [G]"""
import ctypes
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before HIP initialises (halo2-gpu-specific_amd/__init__.py says why)
import torch  # noqa: E402

torch.cuda.init()
import numpy as np  # noqa: E402

from halo2_gpu_specific_amd import circuit as hc, evaluation as ev, prover  # noqa: E402
from halo2_gpu_specific_amd._lib import check  # noqa: E402
from halo2_gpu_specific_amd.transcript import fr_to_mont_limbs  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
A = int(sys.argv[2]) if len(sys.argv) > 2 else 30
G = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rnd = random.Random(1)
cs = hc.ConstraintSystem("wide")
adv = [cs.advice_column() for _ in range(A)]
fix = [cs.fixed_column() for _ in range(8)]
for g in range(G):
    q = cs.query_fixed(rnd.choice(fix))
    a = [cs.query_advice(rnd.choice(adv), rnd.choice([0, 0, 0, 1, -1])) for _ in range(5)]
    cs.create_gate("g%d" % g, [q * (a[0] * a[1] + a[2] - a[3]) * (a[4] + rnd.randrange(1, 100))])
graph, parts, _, _ = hc.compile_evaluator(cs)
muls = sum(1 for c in graph.calculations if c.op == ev.CALC_MUL) + len(parts)
print("k=%d columns=%d+%d gates=%d calculations=%d (mul %d) degree=%d" % (k, A, 8, G, len(graph.calculations), muls, cs.degree()))
ek = k + 2
size = 1 << ek
D = prover.Device()
cols_a = [D.empty(size) for _ in range(A)]
cols_f = [D.empty(size) for _ in range(8)]
for i, t in enumerate(cols_a + cols_f):
    check(D.L.h2_dev_random_fr(bytes([100 + i]) * 32, size, t.data_ptr(), D.stream), "rnd")
zero = fr_to_mont_limbs(0)
b = ev.Builder().build(
    k=k, extended_k=ek, blinding_factors=5, chunk_len=1,
    constants=np.array([fr_to_mont_limbs(c) for c in graph.constants], dtype=np.uint64), rotations=graph.rotations,
    calculations=graph.calculations, value_parts=parts, fixed=[t.data_ptr() for t in cols_f],
    advice=[t.data_ptr() for t in cols_a], y=fr_to_mont_limbs(12345), beta=zero, gamma=zero, theta=zero,
    delta=zero, zeta=zero, extended_omega=fr_to_mont_limbs(3))
h = D.empty(size)


def run():
    check(D.L.h2_dev_evaluate_h(ctypes.byref(b.desc), h.data_ptr(), D.stream), "evalh")


touched = len({(c.a.kind, c.a.index) for c in graph.calculations if c.a.kind >= 2} | {(c.b.kind, c.b.index) for c in graph.calculations if c.b.kind >= 2})
ref = None
for mode in ("interpreter", "generated kernels"):
    if mode == "interpreter":
        b.desc.flags = ev.EVALH_INTERPRET
    else:
        b.desc.flags = 0
        t0 = time.perf_counter()
        info = ev.prepare(b)                      # csrc/evalh_gen.cpp + hipRTC inside the library
        print("h2_evalh_prepare: %.1f s  %s" % (time.perf_counter() - t0, info))
        muls = info["products_per_row"]
        ref = D.download(h).copy()
    run()
    D.sync()
    t0 = time.perf_counter()
    for _ in range(3):
        run()
    D.sync()
    dt = (time.perf_counter() - t0) / 3
    print("%-17s evaluate_h: %.2f ms  -> %.2e products/s (%d per row; the multiplier alone: 1.6e11), %.1f GB/s over the %d distinct columns + output"
          % (mode, dt * 1e3, muls * size / dt, muls, 32 * (touched + 1) * size / dt / 1e9, touched))
assert np.array_equal(ref, D.download(h)), "generated kernels and interpreter disagree"
