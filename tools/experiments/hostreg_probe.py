"""per-call cost of the host-slice entry points that READ vectors, with and without h2_poly_register, from ordinary and from
page-locked memory (k = 22 vectors): where the literal drop-in's time goes.   usage: python tools/experiments/hostreg_probe.py"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# (as bench.py: OpenMP teams -- torch's CPU operators here -- sleep instead of spinning; a spinning team of 256 on a box whose cgroup
# grants 16 CPUs gets the whole process throttled, and every number below with it)
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
os.environ.setdefault("GOMP_SPINCOUNT", "0")
import torch  # noqa: E402

torch.cuda.init()
import numpy as np  # noqa: E402

import halo2_gpu_specific_amd as h2  # noqa: E402

L = h2.lib()
n = 1 << 22
rng = np.random.default_rng(3)


def vec(pinned):
    t = torch.from_numpy(rng.integers(0, 2**61, size=(n, 4), dtype=np.int64))
    return t.pin_memory() if pinned else t


def timeit(name, f, reps=5):
    f()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    print("  %-64s %8.2f ms" % (name, (time.perf_counter() - t0) / reps * 1e3))


x = (ctypes.c_uint64 * 4)(5, 0, 0, 0)
out = (ctypes.c_uint64 * 4)()
for pinned in (False, True):
    print("== vectors in %s memory" % ("page-locked" if pinned else "ordinary"))
    polys = [vec(pinned) for _ in range(12)]
    res = vec(pinned)
    coeffs = np.ascontiguousarray(rng.integers(0, 2**61, size=(12, 4), dtype=np.int64))
    ptrs = (ctypes.c_void_p * 12)(*[p.data_ptr() for p in polys])
    timeit("h2_eval_polynomial, unregistered", lambda: L.h2_eval_polynomial(polys[0].data_ptr(), n, x, out))
    timeit("h2_lincomb of 12, unregistered", lambda: L.h2_lincomb(res.data_ptr(), ptrs, coeffs.ctypes.data, 12, n), 3)
    t0 = time.perf_counter()
    for p in polys:
        L.h2_poly_register(p.data_ptr(), n)
    L.h2_eval_polynomial(polys[0].data_ptr(), n, x, out)
    print("  %-64s %8.2f ms" % ("register 12 + first read of one (its upload)", (time.perf_counter() - t0) * 1e3))
    timeit("h2_eval_polynomial, registered", lambda: L.h2_eval_polynomial(polys[0].data_ptr(), n, x, out))
    t0 = time.perf_counter()
    L.h2_lincomb(res.data_ptr(), ptrs, coeffs.ctypes.data, 12, n)
    print("  %-64s %8.2f ms" % ("h2_lincomb of 12, registered, first (11 uploads)", (time.perf_counter() - t0) * 1e3))
    timeit("h2_lincomb of 12, registered", lambda: L.h2_lincomb(res.data_ptr(), ptrs, coeffs.ctypes.data, 12, n), 3)
    t0 = time.perf_counter()
    for p in polys:
        L.h2_poly_unregister(p.data_ptr())
    print("  %-64s %8.2f ms" % ("unregister 12", (time.perf_counter() - t0) * 1e3))
    fresh = torch.empty((n, 4), dtype=torch.int64, pin_memory=pinned)
    t0 = time.perf_counter()
    L.h2_lincomb(fresh.data_ptr(), ptrs, coeffs.ctypes.data, 2, n)
    print("  %-64s %8.2f ms" % ("h2_lincomb of 2 into a FRESH result vector (first touch)", (time.perf_counter() - t0) * 1e3))
