"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/halo2_hip.h declares, reports errors the documented way without a GPU, and the host-side
logic (device-pool env, sharding plan, host point fold) behaves.  No GPU compute is launched."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import halo2_gpu_specific_amd as h2
from halo2_gpu_specific_amd import parallel
from halo2_gpu_specific_amd._lib import SYMBOLS
from h2util import ROOT, Oracle, arr_to_points, to_mont


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "halo2_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(h2_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = h2.lib()
    names = declared_symbols()
    assert len(names) >= 30
    for name in names:
        assert hasattr(L, name), "libhalo2_hip.so does not export %s" % name
        assert name in SYMBOLS, "python binding table misses %s" % name
    assert sorted(SYMBOLS) == names


def test_every_entry_point_cites_the_reference():
    text = open(os.path.join(ROOT, "include", "halo2_hip.h")).read()
    for needle in ("arithmetic.rs:546", "arithmetic.rs:515", "arithmetic.rs:334", "arithmetic.rs:375", "arithmetic.rs:413",
                   "poly/domain.rs:270", "poly/domain.rs:328", "poly/domain.rs:354"):
        assert needle in text


def test_no_gpu_means_loud_error_not_fallback():
    L = h2.lib()
    if L.h2_device_count() > 0:
        pytest.skip("a GPU is visible")
    a = np.zeros((4, 4), dtype=np.uint64)
    w = np.zeros(4, dtype=np.uint64)
    rc = L.h2_ntt(a.ctypes.data, w.ctypes.data, 2)
    assert rc != 0 and L.h2_last_error()
    with pytest.raises(h2.H2Error):
        h2.arithmetic.best_fft(a, w, 2)


def test_argument_validation_without_gpu():
    L = h2.lib()
    assert L.h2_ntt(None, None, 3) == 1  # H2_ERR_INVALID
    out = np.zeros(12, dtype=np.uint64)
    assert L.h2_msm(None, None, 5, 254, out.ctypes.data) == 1
    # n == 0 / max_bits == 0 -> identity without touching a device (arithmetic.rs:346, :421)
    assert L.h2_msm(None, None, 0, 254, out.ctypes.data) == 0
    assert not out[8:].any()  # z == 0
    assert h2.arithmetic.best_multiexp(np.zeros((0, 4), np.uint64), np.zeros((0, 8), np.uint64))[8:].sum() == 0
    with pytest.raises(AssertionError):  # arithmetic.rs:466 assert_eq!(coeffs.len(), bases.len())
        h2.arithmetic.best_multiexp(np.zeros((2, 4), np.uint64), np.zeros((3, 8), np.uint64))
    with pytest.raises(AssertionError):  # arithmetic.rs:569 assert_eq!(n, 1 << log_n)
        h2.arithmetic.best_fft(np.zeros((5, 4), np.uint64), np.zeros(4, np.uint64), 3)


def test_missing_extension_fails_loudly(tmp_path):
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import halo2_gpu_specific_amd as h2\n"
        "from halo2_gpu_specific_amd import _lib\n"
        "_lib.lib_path = lambda: %r\n"
        "try:\n    h2._lib.lib()\nexcept h2.H2Error as e:\n    print('LOUD', e)\n"
    ) % (ROOT, str(tmp_path / "nope.so"))
    out = subprocess.check_output([sys.executable, "-c", code], text=True)
    assert out.startswith("LOUD") and "no CPU fallback" in out


def test_h2_lib_names_another_build_and_nothing_else(tmp_path):
    """H2_LIB (tools/gen_sanitize.sh: a sanitizer build of the same ABI) replaces the in-tree path; a missing file is the same
    loud error, and the package sets its HIP hardware-queue default without overriding a caller's"""
    code = (
        "import os, sys; sys.path.insert(0, %r)\n"
        "os.environ['H2_LIB'] = %r\n"
        "import halo2_gpu_specific_amd as h2\n"
        "print('QUEUES', os.environ['GPU_MAX_HW_QUEUES'])\n"
        "try:\n    h2.lib()\nexcept h2.H2Error as e:\n    print('LOUD', e)\n"
    ) % (ROOT, str(tmp_path / "elsewhere.so"))
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    out = subprocess.check_output([sys.executable, "-c", code], text=True, env=env)
    assert "QUEUES 8" in out and "LOUD" in out and "elsewhere.so" in out and "no CPU fallback" in out
    out = subprocess.check_output([sys.executable, "-c", code], text=True, env=dict(env, GPU_MAX_HW_QUEUES="3"))
    assert "QUEUES 3" in out


def test_library_sets_its_hardware_queue_default_when_loaded():
    """libhalo2_hip.so's load-time constructor: GPU_MAX_HW_QUEUES=8 for a host that reaches HIP only through the library (the
    runtime reads it at its first call), unless the caller has set it"""
    code = (
        "import ctypes, sys\n"
        "libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p; libc.getenv.argtypes = [ctypes.c_char_p]\n"
        "before = libc.getenv(b'GPU_MAX_HW_QUEUES')\n"
        "ctypes.CDLL(%r)\n"
        "print('BEFORE', before, 'AFTER', libc.getenv(b'GPU_MAX_HW_QUEUES'))\n"
    ) % h2.lib_path()
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    out = subprocess.check_output([sys.executable, "-c", code], text=True, env=env)
    assert "BEFORE None AFTER b'8'" in out, out
    out = subprocess.check_output([sys.executable, "-c", code], text=True, env=dict(env, GPU_MAX_HW_QUEUES="5"))
    assert "BEFORE b'5' AFTER b'5'" in out, out
    out = subprocess.check_output([sys.executable, "-c", code], text=True, env=dict(env, H2_NO_RUNTIME_DEFAULTS="1"))
    assert "BEFORE None AFTER None" in out, out          # the opt-out: the host's environment is left alone


def test_msm_shape_and_scratch():
    L = h2.lib()
    c, W, nb = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
    for n, bits in ((1 << 20, 254), (1 << 24, 254), (1 << 14, 16), (3, 254), (1 << 22, 64)):
        assert L.h2_msm_shape(n, bits, ctypes.byref(c), ctypes.byref(W), ctypes.byref(nb)) == 0
        assert W.value * c.value >= bits + 1  # room for the signed-digit carry
        assert nb.value == 1 << (c.value - 1)
        assert L.h2_msm_scratch_bytes(n, bits) > n * W.value * 8


def test_sharding_plan():
    cols = [parallel.shard_columns(11, 4, r) for r in range(4)]
    assert sorted(sum(cols, [])) == list(range(11))
    for n in (0, 1, 7, 1 << 20, (1 << 20) + 3):
        for world in (1, 2, 3, 8):
            spans = [parallel.msm_split_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
                assert a1 == b0 and a0 <= a1
            part = (n + world - 1) // world if n else 0
            assert all(hi - lo <= part for lo, hi in spans)


def test_host_point_fold_matches_oracle():
    oracle = Oracle.get()
    n = 96
    s, p = oracle.random_fr(31, n), oracle.random_g1(32, n)
    parts = np.stack([oracle.best_multiexp(s[i : i + 24], p[i : i + 24]) for i in range(0, n, 24)])
    ident = np.zeros((1, 12), dtype=np.uint64)
    ident[0, 4:8] = to_mont([1], 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47)[0]
    parts = np.concatenate([parts, ident, parts[:1], parts[:1]])  # identity + a repeated point (doubling branch)
    got = arr_to_points(oracle.to_affine(parallel.g1_sum(parts)))[0]
    want_j = oracle.best_multiexp(s, p)
    two = np.zeros(12, dtype=np.uint64)
    oracle.lib.oracle_g1_double(parts[0].ctypes.data, two.ctypes.data)
    full = np.zeros(12, dtype=np.uint64)
    oracle.lib.oracle_g1_add(want_j.ctypes.data, two.ctypes.data, full.ctypes.data)
    assert got == arr_to_points(oracle.to_affine(full))[0]
    assert parallel.g1_sum(np.zeros((0, 12), np.uint64))[8:].sum() == 0


def test_device_pool_env(tmp_path):
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import halo2_gpu_specific_amd as h2\n"
        "print(h2.lib().h2_device_count())\n"
    ) % ROOT
    env = dict(os.environ, HALO2_PROOFS_N_GPU="3")
    out = subprocess.check_output([sys.executable, "-c", code], text=True, env=env)
    # no GPU here: the pool is empty whatever the variable says; on a GPU box it would print 3
    assert out.strip() in ("0", "3")
