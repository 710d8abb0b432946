"""why do host-slice calls on ORDINARY memory cost what they cost?  h2_prefix_product (one upload, one kernel, one download, in
place) on: the same numpy array again and again; a new touched array per call; an array last written by another library call;
from one thread and from four at once.   usage: python tools/experiments/pageable_calls_probe.py"""
import concurrent.futures
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
import numpy as np  # noqa: E402

import halo2_gpu_specific_amd as h2  # noqa: E402

L = h2.lib()
n = 1 << 22
rng = np.random.default_rng(3)


def new():
    return rng.integers(0, 2**61, size=(n, 4), dtype=np.uint64)


def ms(f, reps=1):
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    return (time.perf_counter() - t0) / reps * 1e3


one = np.array([1, 0, 0, 0], dtype=np.uint64)
L.h2_batch_mont(one.ctypes.data, 1)
z0 = np.zeros((n, 4), dtype=np.uint64)


def pp(f, z=None):
    z = z0 if z is None else z
    if isinstance(f, int):
        return L.h2_prefix_product(f, n, one.ctypes.data, z if isinstance(z, int) else z.ctypes.data)
    return L.h2_prefix_product(f.ctypes.data, n, one.ctypes.data, z.ctypes.data)


a = new()
pp(a)
print("same arrays, 5 calls:           ", ["%.2f" % ms(lambda: pp(a)) for _ in range(5)])
arrs = [new() for _ in range(5)]
print("a new touched input per call:   ", ["%.2f" % ms(lambda x=x: pp(x)) for x in arrs])
fresh = [np.empty((n, 4), dtype=np.uint64) for _ in range(5)]
print("a FRESH np.empty result per call:", ["%.2f" % ms(lambda z=z: pp(a, z)) for z in fresh])
print("... those results again:        ", ["%.2f" % ms(lambda z=z: pp(a, z)) for z in fresh])
del fresh
outs = []
for x in arrs:
    o = np.empty((n, 4), dtype=np.uint64)
    ptrs = (ctypes.c_void_p * 1)(x.ctypes.data)
    c = np.array([[1, 0, 0, 0]], dtype=np.uint64)
    L.h2_batch_mont(c.ctypes.data, 1)
    t = ms(lambda: L.h2_lincomb(o.ctypes.data, ptrs, c.ctypes.data, 1, n))
    outs.append((o, t))
print("lincomb of 1 into np.empty:     ", ["%.2f" % t for _, t in outs])
print("prefix product of those results:", ["%.2f" % ms(lambda o=o: pp(o)) for o, _ in outs])
print("... again:                      ", ["%.2f" % ms(lambda o=o: pp(o)) for o, _ in outs])
pool = concurrent.futures.ThreadPoolExecutor(max_workers=4)
for rep in range(2):
    arrs = [new() for _ in range(4)]
    t0 = time.perf_counter()
    zs = [np.empty((n, 4), dtype=np.uint64) for _ in range(4)]
    t0 = time.perf_counter()
    each = list(pool.map(lambda i: ms(lambda: pp(arrs[i], zs[i])), range(4)))
    print("4 threads, 4 new inputs, 4 FRESH results: wall %.2f ms, per call %s" % ((time.perf_counter() - t0) * 1e3, ["%.2f" % t for t in each]))
arrs = [new() for _ in range(4)]
zs = [np.zeros((n, 4), dtype=np.uint64) for _ in range(4)]
for rep in range(2):
    t0 = time.perf_counter()
    each = list(pool.map(lambda i: ms(lambda: pp(arrs[i], zs[i])), range(4)))
    print("4 threads, touched inputs and results: wall %.2f ms, per call %s" % ((time.perf_counter() - t0) * 1e3, ["%.2f" % t for t in each]))
pool2 = concurrent.futures.ThreadPoolExecutor(max_workers=2)
for rep in range(2):
    t0 = time.perf_counter()
    each = list(pool2.map(lambda i: ms(lambda: pp(arrs[i], zs[i])), range(4)))
    print("2 threads, touched inputs and results, 4 calls: wall %.2f ms, per call %s" % ((time.perf_counter() - t0) * 1e3, ["%.2f" % t for t in each]))
pin = []
for _ in range(8):
    p = ctypes.c_void_p()
    assert L.h2_host_alloc_pinned(32 * n, ctypes.byref(p)) == 0
    src = new()
    ctypes.memmove(p.value, src.ctypes.data, 32 * n)
    pin.append(p.value)
print("page-locked, one thread:        ", ["%.2f" % ms(lambda i=i: pp(pin[i], pin[4 + i])) for i in range(4)])
t0 = time.perf_counter()
each = list(pool.map(lambda i: ms(lambda: pp(pin[i], pin[4 + i])), range(4)))
print("page-locked, 4 threads: wall %.2f ms, per call %s" % ((time.perf_counter() - t0) * 1e3, ["%.2f" % t for t in each]))
# alloc / free of the host's own vectors
t0 = time.perf_counter()
x = np.empty((n, 4), dtype=np.uint64)
t1 = time.perf_counter()
x[:] = 0
t2 = time.perf_counter()
del x
t3 = time.perf_counter()
print("np.empty %.2f ms, first touch (fill) %.2f ms, free %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
