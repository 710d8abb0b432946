"""create_proof per phase on the wide circuit (circuits.wide: 4 * quads advice columns, degree 5, quads / 2 logup range
lookups), real SRS from the device setup.   usage: python tools/wide_bench.py [k] [quads] [budget|-] [wide|compact|resident]
`budget` (e.g. 8G, or `coset`) runs the memory-bounded route: coefficient forms only, the extended domain coset by coset.
witness form: `wide` = 32-byte cells in pinned host memory (what the reference hands over), `compact` = 8-byte cells (every
value of this circuit fits 64 bits; widened on the device), `resident` = the columns already on the device."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before HIP initialises (halo2-gpu-specific_amd/__init__.py says why)
import torch  # noqa: E402

torch.cuda.init()

from halo2_gpu_specific_amd import circuits, prover  # noqa: E402
from halo2_gpu_specific_amd.rng import ProverRng  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
quads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
budget = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != "-" else None
form = sys.argv[4] if len(sys.argv) > 4 else "wide"
cs = circuits.wide(quads)
if budget == "coset":
    D = prover.Device(eval_cache=1)
elif budget:
    D = prover.Device(mem_budget=prover.parse_bytes(budget))
else:
    D = prover.Device()
params = prover.Params.unsafe_setup(D, k, 0x1D0C5F0A3B7E91C2A4D6F8091B2C3D4E5F60718293A4B5C6D7E8F9010203)
t0 = time.perf_counter()
adv, fixed, copies = circuits.wide_synthesize(k, quads, alloc=D.pinned_columns, compact=form == "compact")
if form == "resident":
    adv = [D.upload(a) for a in adv]
    D.sync()
print("synthesize %.2f s (%s witness)" % (time.perf_counter() - t0, form))
t0 = time.perf_counter()
pk = prover.keygen(D, params, cs, fixed, copies)
D.sync()
print("keygen %.3f s, residency %s, footprint %s GiB" % (time.perf_counter() - t0, pk.residency,
      {m: round(b / 2**30, 1) for m, b in prover.footprint(cs, pk.domain).items()}))
for rep in range(3):
    timings = {}
    torch.cuda.reset_peak_memory_stats()
    ta = time.perf_counter()
    proof = prover.create_proof_with_shplonk(D, params, pk, adv, ProverRng(rep), timings=timings if rep else None)
    D.sync()
    print("rep %d: %.1f ms (%d bytes) peak %.1f GiB + library %.1f GiB %s" % (
        rep, (time.perf_counter() - ta) * 1e3, len(proof), torch.cuda.max_memory_allocated() / 2**30,
        D.L.h2_library_memory_bytes() / 2**30, {a: round(b * 1e3, 1) for a, b in timings.items()}))
