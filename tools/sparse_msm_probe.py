import sys, time
sys.path.insert(0, '/root/repo')
import torch; torch.cuda.init()
from halo2_gpu_specific_amd import circuits, prover
from halo2_gpu_specific_amd._lib import check
k = 22; n = 1 << k
D = prover.Device(); L = D.L
params = prover.Params.synthetic(D, k)
adv, fixed, copies = circuits.mini_plonk_synthesize(k, alloc=D.pinned_columns)
def once(name, f):
    D.sync(); t0 = time.perf_counter(); r = f(); D.sync(); print("%-44s %.2f ms" % (name, (time.perf_counter() - t0) * 1e3)); return r
for rep in range(2):
    for ci in range(3):
        t, ev = D.upload_async(adv[ci]); D.tstream.wait_event(ev); D.sync(); ev.synchronize()
        once("col %d msm raw canonical (no mont) bits16" % ci, lambda: D.msm(t, params.g_lagrange, n, 16))
        D.set_rows_raw(t, n - 6, [60000, 2, 3, 4, 5, 6])
        b = D.max_scalar_bits(t); print("bits", b)
        check(L.h2_dev_batch_mont(t.data_ptr(), n, D.stream), "m")
        once("col %d msm after mont, bits=%d" % (ci, b), lambda: D.msm(t, params.g_lagrange, n, b))
        once("col %d msm again" % ci, lambda: D.msm(t, params.g_lagrange, n, b))
