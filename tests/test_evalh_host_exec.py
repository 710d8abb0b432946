"""The evaluate_h GENERATOR checked without a GPU: the straight-line source the library generates for a program
(h2_evalh_source) is compiled FOR THE HOST with g++ -- field.hpp has a portable multiplier next to the gfx950 one -- and run
row by row with the argument block the library would pass to the kernel (h2_evalh_stage_args): the result must be the CPU
oracle's evaluate_h (oracle/: the restatement of plonk/evaluation.rs:778-1226), bit for bit, for every grouping / staging /
pairing option of the generator.  What this pins on the CPU: the DAG and its simplifications, the factor grouping, the fold
order, fp_mul2 pairing, stage splitting with accumulation, load eviction, the scalar and column tables of the argument block.
What it cannot: the gfx950 code object itself (tests/test_gpu_evalh.py does, on the device)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from evalh_cases import oracle_evaluate_h, random_case
from h2util import R_MOD, from_mont, to_mont
from halo2_gpu_specific_amd import evaluation as ev

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "halo2-gpu-specific_amd", "csrc")

SHIM = r'''
// host shim (tests only): the generated evaluate_h source compiled by g++ and run one row at a time
#include <cstddef>
#include <cstdint>
#include <cstring>
#define __HIPCC_RTC__ 1
#define __host__
#define __device__
#define __global__
#define __forceinline__ inline
#define __noinline__
#define __launch_bounds__(...)
#define __shared__ static
#define __syncthreads() do {} while (0)
#define __builtin_amdgcn_sched_barrier(x) do {} while (0)
struct uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
struct uint2 { unsigned x, y; };
struct h2_dim3 { unsigned x, y, z; };
static thread_local h2_dim3 threadIdx, blockIdx, blockDim, gridDim;
'''

DRIVER = r'''
extern "C" void h2_host_run(const unsigned char* block, unsigned long long bytes) {
    Args a;
    memset(&a, 0, sizeof a);
    memcpy(&a, block, bytes < sizeof a ? bytes : sizeof a);
    blockDim.x = 256;
    // a stage that reads its arguments through LDS copies them cooperatively (lane t takes entries t, t + 256, ...) before
    // anything else: every lane of a workgroup past the row range fills the (here: static) arrays once, and returns
    blockIdx.x = 0x7fffffu;
    for (unsigned t = 0; t < 256; t++) {
        threadIdx.x = t;
        h2_evalh_gen(a);
    }
    for (unsigned long long r = a.row_begin; r < a.row_end; r++) {
        const unsigned long long i = r - a.row_begin;
        blockIdx.x = (unsigned)(i / 256);
        threadIdx.x = (unsigned)(i % 256);
        h2_evalh_gen(a);
    }
}
'''

SETTINGS = [
    {},
    {"H2_JIT_FACTOR": "0"},
    {"H2_JIT_STAGE_PRODUCTS": "8"},
    {"H2_JIT_STAGE_PRODUCTS": "20", "H2_JIT_GROUP": "3", "H2_JIT_MAX_AHEAD": "2", "H2_JIT_GAP": "4"},
    {"H2_JIT_MUL2": "0"},
    {"H2_JIT_LIVE": "6", "H2_JIT_GAP": "200"},
    {"H2_JIT_MIN_GROUP": "1"},
    {"H2_JIT_LDS_ARGS": "1"},             # scalars and column pointers through the LDS copy of the argument block
    {"H2_JIT_LDS_ARGS": "1", "H2_JIT_STAGE_PRODUCTS": "13"},
]


def _omega_tables(kw):
    ek = kw["extended_k"]
    w = from_mont(kw["extended_omega"])[0]
    lo_n = min(1 << ek, 4096)
    lo, acc = [], 1
    for _ in range(lo_n):
        lo.append(acc)
        acc = acc * w % R_MOD
    step = pow(w, 4096, R_MOD)
    hi, acc = [], 1
    for _ in range(max(1, (1 << ek) >> 12)):
        hi.append(acc)
        acc = acc * step % R_MOD
    return to_mont(lo), to_mont(hi)


def _run_generated(tmp_path, b, kw, tag):
    size = 1 << kw["extended_k"]
    values = np.zeros((size, 4), dtype=np.uint64)
    tw_lo, tw_hi = _omega_tables(kw)
    stage = 0
    while True:
        try:
            src = ev.generated_source(b, stage)
        except Exception:                                   # "no such stage": past the last one
            break
        cpp = tmp_path / ("%s_s%d.cpp" % (tag, stage))
        so = tmp_path / ("%s_s%d.so" % (tag, stage))
        cpp.write_text(SHIM + src + DRIVER)
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-w", "-I", CSRC, str(cpp), "-o", str(so)])
        block = ev.stage_args(b, stage, values.ctypes.data, tw_lo.ctypes.data, tw_hi.ctypes.data, 0, size)
        run = ctypes.CDLL(str(so)).h2_host_run
        run.argtypes = [ctypes.c_char_p, ctypes.c_uint64]
        run.restype = None
        run(block, len(block))
        stage += 1
    assert stage >= 1
    return values, stage


@pytest.mark.parametrize("seed,k,ek,kwargs", [
    (1, 2, 3, {}), (2, 5, 7, {}), (7, 6, 8, dict(with_perm=False, lookup_sets=(), n_shuffles=0, n_calcs=3)),
    (8, 7, 9, dict(lookup_sets=(2,), n_shuffles=1, n_calcs=60)), (6, 13, 13, dict(n_calcs=16, lookup_sets=(1,), n_shuffles=1))])
def test_generated_source_run_on_the_host_matches_the_oracle(oracle, monkeypatch, tmp_path, seed, k, ek, kwargs):
    monkeypatch.setenv("H2_JIT_CACHE", str(tmp_path / "cache"))
    kw = random_case(seed, k, ek, oracle, **({"n_calcs": 30} | kwargs))
    b = ev.Builder().build(**kw)
    want = oracle_evaluate_h(oracle, b)
    settings = SETTINGS if ek <= 9 else SETTINGS[:2]
    for i, env in enumerate(settings):
        for name, value in env.items():
            monkeypatch.setenv(name, value)
        got, stages = _run_generated(tmp_path, b, kw, "c%d_o%d" % (seed, i))
        if "H2_JIT_STAGE_PRODUCTS" in env and kwargs.get("n_calcs", 30) >= 30:
            assert stages >= 2, env
        assert np.array_equal(got, want), (env, int(np.argmax((got != want).any(axis=1))))
        for name in env:
            monkeypatch.delenv(name)


def test_stage_args_layout_and_errors(oracle):
    """the argument block: values / tables / range in the fixed part, then the uniform scalars, then the column pointers; a
    stage past the last one and a descriptor with a null column are refused"""
    kw = random_case(3, 4, 6, oracle, n_calcs=12)
    b = ev.Builder().build(**kw)
    values = np.zeros((1 << 6, 4), dtype=np.uint64)
    block = ev.stage_args(b, 0, values.ctypes.data, 0x1000, 0x2000, 5, 60)
    fixed = np.frombuffer(block[:48], dtype=np.uint64)
    assert fixed[0] == values.ctypes.data and fixed[1] == 0x1000 and fixed[2] == 0x2000 and fixed[3] == 5 and fixed[4] == 60
    assert np.frombuffer(block[40:48], dtype=np.uint32).tolist() == [6, 4]          # extended_k, rot_scale
    assert (len(block) - 48) % 8 == 0
    with pytest.raises(Exception, match="no such stage"):
        ev.stage_args(b, 99, values.ctypes.data, 0, 0, 0, 64)
    kw2 = dict(kw, l0=None)
    with pytest.raises(Exception):
        ev.stage_args(ev.Builder().build(**kw2), 0, values.ctypes.data, 0, 0, 0, 64)


@pytest.mark.timeout(300)
def test_generator_fuzz_on_the_host():
    """tools/gen_host_fuzz.py for a short budget: the product circuits' programs, random circuits and random Evaluator programs
    under random generator options, each generated, compiled for the host and run against the oracle (long runs: DESIGN.md)"""
    import sys

    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_host_fuzz.py"), "25", "11"], capture_output=True, text=True,
                         timeout=280)
    assert out.returncode == 0 and "all equal to the oracle" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
