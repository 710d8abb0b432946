"""evaluate_h alone: the generated kernels (gate-only + library argument kernels vs the fused one) on random resident
columns, timed with HIP events.  usage: evalh_probe.py mini|wide <extended log size> [k]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
torch.cuda.init()
import halo2_gpu_specific_amd as h2
from halo2_gpu_specific_amd import jit, circuits, evaluation as ev
from halo2_gpu_specific_amd.circuit import compile_evaluator
from halo2_gpu_specific_amd._lib import check
from halo2_gpu_specific_amd.transcript import fr_to_mont_limbs

which, ek = sys.argv[1], int(sys.argv[2])
cs = circuits.mini_plonk() if which == "mini" else circuits.wide(16)
dk = {3: 1, 5: 2}[cs.degree()]
k = ek - dk
g, parts, lk, sh = compile_evaluator(cs)
chunk = cs.degree() - 2
ncols = len(cs.perm_columns)
A = {"advice": ev.ANY_ADVICE, "fixed": ev.ANY_FIXED, "instance": ev.ANY_INSTANCE}
nsets = (ncols + chunk - 1) // chunk
perm = dict(n_sets=nsets, chunk_len=chunk, columns=[(A[kd], i) for kd, i in cs.perm_columns], last_rotation=-6)
L = h2.lib()
dev = torch.device("cuda", 0)
size = 1 << ek
gen = torch.Generator(device=dev); gen.manual_seed(1)
def col():
    t = torch.randint(-(2**63), 2**63 - 1, (size, 4), dtype=torch.int64, device=dev, generator=gen)
    t[:, 3] &= 0x0FFFFFFFFFFFFFFF
    return t
fixed = [col() for _ in range(cs.num_fixed)]
advice = [col() for _ in range(cs.num_advice)]
l0, ll, lar = col(), col(), col()
pz = [col() for _ in range(nsets)]
sg = [col() for _ in range(ncols)]
nlz = sum(len(sets) for _, _, sets in cs.lookups)
lz = [col() for _ in range(nlz)]
lm = [col() for _ in range(len(cs.lookups))]
out = torch.empty((size, 4), dtype=torch.int64, device=dev)
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
root = pow(0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C, 1 << (28 - ek), R)
def build(fn, covers):
    return ev.Builder().build(
        k=k, extended_k=ek, blinding_factors=5, chunk_len=chunk,
        constants=np.array([fr_to_mont_limbs(c) for c in g.constants], dtype=np.uint64), rotations=g.rotations,
        calculations=g.calculations, value_parts=parts, lookups=lk, shuffles=sh,
        fixed=[t.data_ptr() for t in fixed], advice=[t.data_ptr() for t in advice], instance=[],
        l0=l0.data_ptr(), l_last=ll.data_ptr(), l_active_row=lar.data_ptr(),
        perm_z=[t.data_ptr() for t in pz], perm_columns=perm["columns"], perm_sigma=[t.data_ptr() for t in sg],
        lookup_z=[t.data_ptr() for t in lz], lookup_m=[t.data_ptr() for t in lm], shuffle_z=[],
        y=fr_to_mont_limbs(5), beta=fr_to_mont_limbs(7), gamma=fr_to_mont_limbs(11), theta=fr_to_mont_limbs(13),
        delta=fr_to_mont_limbs(17), zeta=fr_to_mont_limbs(19), extended_omega=fr_to_mont_limbs(root),
        jit_function=fn, jit_covers=covers)
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(b, reps=5):
    check(L.h2_dev_evaluate_h(ctypes.byref(b.desc), out.data_ptr(), stream), "evalh")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        check(L.h2_dev_evaluate_h(ctypes.byref(b.desc), out.data_ptr(), stream), "evalh")
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
path = jit.compile_program(g.rotations, g.calculations, parts, lk, sh)
base = build(jit.load(path), 0)
t_base = run(base)
ref = out.clone()
streams_fused = cs.num_fixed + cs.num_advice + 3 + nsets + ncols + nlz + len(lm) + 1
print("%s 2^%d: gate kernel + library argument kernels %.3f ms" % (which, ek, t_base))
# the gate program alone under the new load scheduling (the library's argument kernels behind it)
for grp, ahead, gap, sb in [(6, 6, 30, True), (10, 6, 20, True), (6, 6, 30, False), (10, 6, 20, False), (16, 8, 30, False)]:
    jit._GROUP, jit._MAX_AHEAD, jit._GAP, jit._STMT_BARRIER = grp, ahead, gap, sb
    src, _ = jit.generate_fused_source(g.rotations, g.calculations, parts, lk, sh, perm, fold_args=False)
    b = build(jit.load(jit.compile_source(src, "_fused")), 0)
    t = run(b)
    print("  gates only, loads a group ahead (group=%d ahead=%d gap=%d stmt barriers=%s) + library argument kernels: %.3f ms  same=%s" % (
        grp, ahead, gap, sb, t, torch.equal(out, ref)))
jit._STMT_BARRIER = True
os.environ["H2_EVALH_FUSED"] = "1"
sweep = [(6, 6, 30)] if len(sys.argv) > 3 and sys.argv[3] == "single" else [(6, 6, 30), (6, 4, 30), (4, 4, 16), (8, 8, 30), (6, 3, 12), (10, 6, 20), (6, 10, 60)]
for grp, ahead, gap in sweep:
    jit._GROUP, jit._MAX_AHEAD, jit._GAP = grp, ahead, gap
    fused, covers = jit.compile_program(g.rotations, g.calculations, parts, lk, sh, perm=perm)
    b = build(jit.load(fused), covers)
    t = run(b)
    ok = torch.equal(out, ref)
    print("  fused group=%d ahead=%d gap=%d: %.3f ms  same=%s  (%.0f GB/s over %d streams; %d products per row: %.3g/s = %.2f of the multiplier's own rate)" % (
        grp, ahead, gap, t, ok, streams_fused * 32 * size / t / 1e6, streams_fused, jit.last_stats["products_per_row"],
        jit.last_stats["products_per_row"] * size / t * 1e3, jit.last_stats["products_per_row"] * size / t * 1e3 / 1.6e11))
