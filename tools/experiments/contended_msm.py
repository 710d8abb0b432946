"""P processes sharing one GPU, each repeating the same MSMs (single, batched = fused or pipelined) and comparing every result
with its first: does a result ever change when the device is contended?
usage: contended_msm.py <processes> <log_n> <columns> <seconds>   (H2_MSM_NO_FUSE=1: the two-stream pipeline instead of fusing)"""
import ctypes
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    procs, log_n, cols, seconds = (int(v) for v in sys.argv[1:5])
    if "H2_CHILD" not in os.environ:
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=dict(os.environ, H2_CHILD=str(i)))
              for i in range(procs)]
        sys.exit(max(p.wait() for p in ps))
    import numpy as np
    import torch

    import halo2_gpu_specific_amd as h2

    L = h2.lib()
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    n = 1 << log_n
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    bases = torch.empty((n, 8), dtype=torch.int64, device=dev)
    assert L.h2_dev_random_points(0x48414C4F32, n, bases.data_ptr(), stream) == 0
    sc = []
    for _ in range(cols):
        t = torch.randint(-(2**63), 2**63 - 1, (n, 4), dtype=torch.int64, device=dev, generator=g)
        t[:, 3] &= 0x1FFFFFFFFFFFFFFF
        sc.append(t)
    torch.cuda.synchronize()
    sbytes = max(L.h2_msm_batch_scratch_bytes(n, 254, cols), 2 * ((L.h2_msm_scratch_bytes(n, 254) + 255) // 256 * 256))
    scratch = torch.empty(sbytes, dtype=torch.uint8, device=dev)
    ptrs = (ctypes.c_void_p * cols)(*[t.data_ptr() for t in sc])

    from bench import jac_eq          # Jacobian representations differ with the summation order: compare group elements

    def same(a, b):
        return jac_eq(np.frombuffer(a, dtype=np.uint64), np.frombuffer(b, dtype=np.uint64))

    def single(j):
        res = np.zeros(12, dtype=np.uint64)
        assert L.h2_dev_msm(sc[j].data_ptr(), bases.data_ptr(), n, 254, scratch.data_ptr(), sbytes, vp(res), stream) == 0, L.h2_last_error()
        return res.tobytes()

    def batch():
        res = np.zeros((cols, 12), dtype=np.uint64)
        assert L.h2_dev_msm_batch(ptrs, cols, bases.data_ptr(), n, 254, scratch.data_ptr(), sbytes, vp(res), stream) == 0, L.h2_last_error()
        return [res[j].tobytes() for j in range(cols)]

    first_s = [single(j) for j in range(cols)]
    first_b = batch()
    assert all(same(a, b) for a, b in zip(first_s, first_b)), "the first batch differs from the single MSMs"
    bad_s = bad_b = rounds = 0
    t0 = time.time()
    while time.time() - t0 < seconds:
        rounds += 1
        bad_s += sum(not same(single(j), first_s[j]) for j in range(cols))
        bad_b += sum(not same(a, b) for a, b in zip(batch(), first_b))
    print("child %s: 2^%d x %d columns, %d rounds: single results that changed %d, batched %d" % (
        os.environ["H2_CHILD"], log_n, cols, rounds, bad_s, bad_b), flush=True)
    sys.exit(1 if bad_s or bad_b else 0)


main()
