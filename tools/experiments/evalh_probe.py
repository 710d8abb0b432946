"""evaluate_h alone on random resident columns, timed with HIP events around h2_dev_evaluate_h: the interpreter kernels, the
library-generated kernels (csrc/evalh_gen.cpp) under their default options and under the option sets given as
NAME=VALUE,NAME=VALUE arguments.  usage: evalh_probe.py mini|wide <extended log size> [H2_JIT_...=v,... ]..."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
torch.cuda.init()
import halo2_gpu_specific_amd as h2
from halo2_gpu_specific_amd import circuits, evaluation as ev
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
def build(flags=0):
    return ev.Builder().build(
        k=k, extended_k=ek, blinding_factors=5, chunk_len=chunk,
        constants=np.array([fr_to_mont_limbs(c) for c in g.constants], dtype=np.uint64), rotations=g.rotations,
        calculations=g.calculations, value_parts=parts, lookups=lk, shuffles=sh,
        fixed=[t.data_ptr() for t in fixed], advice=[t.data_ptr() for t in advice], instance=[],
        l0=l0.data_ptr(), l_last=ll.data_ptr(), l_active_row=lar.data_ptr(),
        perm_z=[t.data_ptr() for t in pz], perm_columns=[(A[kd], i) for kd, i in cs.perm_columns], perm_sigma=[t.data_ptr() for t in sg],
        lookup_z=[t.data_ptr() for t in lz], lookup_m=[t.data_ptr() for t in lm], shuffle_z=[],
        y=fr_to_mont_limbs(5), beta=fr_to_mont_limbs(7), gamma=fr_to_mont_limbs(11), theta=fr_to_mont_limbs(13),
        delta=fr_to_mont_limbs(17), zeta=fr_to_mont_limbs(19), extended_omega=fr_to_mont_limbs(root), flags=flags)
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(b, reps=10):
    check(L.h2_dev_evaluate_h(ctypes.byref(b.desc), out.data_ptr(), stream), "evalh")
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        check(L.h2_timer_start(stream), "timer")
        for _ in range(reps):
            check(L.h2_dev_evaluate_h(ctypes.byref(b.desc), out.data_ptr(), stream), "evalh")
        ms = ctypes.c_float()
        check(L.h2_timer_stop(stream, ctypes.byref(ms)), "timer")
        best = min(best, ms.value / reps)
    return best
t_int = run(build(ev.EVALH_INTERPRET), reps=3)
ref = out.clone()
print("%s 2^%d: interpreter kernels %.3f ms" % (which, ek, t_int))
for spec in [""] + sys.argv[3:]:
    env = dict(kv.split("=") for kv in spec.split(",") if kv)
    for name, value in env.items():
        os.environ[name] = value
    b = build()
    t0 = time.perf_counter()
    info = ev.prepare(b)
    t_prep = time.perf_counter() - t0
    t = run(b)
    print("  generated %-40s %.3f ms  same=%s  stages=%d products/row=%d (as written %d) vectors=%d regs=%d scratch=%d  %.0f GB/s  %.3g products/s = %.2f of the multiplier's own rate  (prepare %.2f s, cache %d)" % (
        spec or "(defaults)", t, torch.equal(out, ref), info["stages"], info["products_per_row"], info["reference_products_per_row"],
        info["vectors_read"], info["max_registers"], info["scratch_bytes"], (info["vectors_read"] + 1) * 32 * size / t / 1e6,
        info["products_per_row"] * size / t * 1e3, info["products_per_row"] * size / t * 1e3 / 1.6e11, t_prep, info["from_cache"]))
    for name in env:
        del os.environ[name]
