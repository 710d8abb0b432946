"""GPU parity of the MSM over a shifted-base table (h2_dev_bases_precompute: all digits of a scalar share one bucket
set, msm.hip) against the CPU oracle's `best_multiexp` (arithmetic.rs:20-108 restated) -- same group element for every
scalar distribution, bound, digit count, sub-range and batch shape the windowed pipeline is tested on."""
import ctypes
import os
import random

import numpy as np
import pytest

import halo2_gpu_specific_amd as h2
from h2util import R_MOD, arr_to_points, from_mont, to_mont

pytestmark = pytest.mark.gpu


def _affine(oracle, jac):
    return arr_to_points(oracle.to_affine(np.asarray(jac, dtype=np.uint64)))[0]


@pytest.fixture()
def small_tables():
    """tables are used from 2^15 scalars on; the tests also run them on small inputs"""
    os.environ["H2_MSM_TABLE_MIN_N"] = "1"
    yield
    del os.environ["H2_MSM_TABLE_MIN_N"]


class DevMsm:
    def __init__(self, pts):
        import torch

        self.torch, self.L = torch, h2.lib()
        self.n = len(pts)
        self.d_pts = torch.from_numpy(np.ascontiguousarray(pts).view(np.int64)).cuda()

    def precompute(self, digits=0):
        rc = self.L.h2_dev_bases_precompute(self.d_pts.data_ptr(), self.n, digits, None)
        assert rc == 0, self.L.h2_last_error()

    def forget(self):
        assert self.L.h2_dev_bases_forget(self.d_pts.data_ptr()) == 0

    def msm(self, scalars, bits=254, lo=0, m=None):
        m = self.n - lo if m is None else m
        d_s = self.torch.from_numpy(np.ascontiguousarray(scalars).view(np.int64)).cuda()
        nbytes = self.L.h2_msm_scratch_bytes(m, bits)
        scratch = self.torch.empty(max(nbytes, 256), dtype=self.torch.uint8, device="cuda")
        out = np.zeros(12, dtype=np.uint64)
        rc = self.L.h2_dev_msm(d_s.data_ptr(), self.d_pts.data_ptr() + 64 * lo, m, bits, scratch.data_ptr(), nbytes,
                               out.ctypes.data, None)
        assert rc == 0, self.L.h2_last_error()
        return out


def _columns(oracle, n, seed):
    rnd = random.Random(seed)
    rand = from_mont(oracle.random_fr(seed, n))
    hot = rnd.randrange(R_MOD)
    return {
        "uniform": (rand, 254),
        "64-bit": ([v & ((1 << 64) - 1) for v in rand], 64),
        "128-bit": ([v & ((1 << 128) - 1) for v in rand], 128),
        "200-bit bound, 254 claimed": ([v & ((1 << 200) - 1) for v in rand], 254),
        "7/8 dominant": ([hot if i >= n // 8 else rand[i] for i in range(n)], 254),
        "all one value": ([hot] * n, 254),
        "r - 1 and friends": ([R_MOD - 1 if i % 4 else rand[i] for i in range(n)], 254),
        "mostly zero": ([rand[i] if i % 17 == 0 else 0 for i in range(n)], 254),
        "digit extremes": ([((1 << 253) - 1, 1 << 252, (1 << 254) - 1 - (i % 3), 0x5555 << 200)[i % 4] % R_MOD for i in range(n)], 254),
    }


@pytest.mark.parametrize("n,digits", [(1000, 0), (4096, 11), (4099, 12), (1 << 13, 32), (1 << 15, 0), ((1 << 16) + 123, 0)])
def test_table_msm_vs_oracle(oracle, small_tables, n, digits):
    pts = oracle.random_g1(700 + n, n)
    pts[5] = 0            # identity bases stay identities on every level
    pts[n - 1] = 0
    pts[11] = pts[10]     # equal bases: P + P inside a bucket when their digits agree
    dev = DevMsm(pts)
    dev.precompute(digits)
    try:
        for name, (vals, bits) in _columns(oracle, n, n).items():
            vals = list(vals)
            vals[11] = vals[10]
            scalars = to_mont(vals)
            want = _affine(oracle, oracle.best_multiexp(scalars, pts))
            assert _affine(oracle, dev.msm(scalars, bits)) == want, name
    finally:
        dev.forget()


def test_table_sub_ranges_and_forget(oracle, small_tables):
    """a range-split MSM (gpu_multiexp_bound, arithmetic.rs:413-440) passes bases + lo: the table rows of that range
    are used; after h2_dev_bases_forget the windowed pipeline answers, with the same points"""
    n = 1 << 14
    pts = oracle.random_g1(801, n)
    scalars = oracle.random_fr(802, n)
    scalars[n // 4:n // 2] = scalars[3]     # a dominant value over whole 256-row blocks (their tabulated sums are used when the
    scalars[n // 2 + 77:] = scalars[3]      # view starts on a block boundary) and over ragged runs
    dev = DevMsm(pts)
    dev.precompute()
    got = {}
    for lo, m in ((0, n), (0, n // 2), (n // 2, n // 2), (1000, 3000), (n - 1, 1), (7, n - 7), (4096, 8192), (4097, 8000), (n // 4, n // 4)):
        want = _affine(oracle, oracle.best_multiexp(scalars[lo:lo + m], pts[lo:lo + m]))
        got[(lo, m)] = dev.msm(scalars[lo:lo + m], 254, lo, m)
        assert _affine(oracle, got[(lo, m)]) == want, (lo, m)
    need_with_table = dev.L.h2_msm_scratch_bytes(n, 254)
    dev.forget()
    assert dev.L.h2_msm_scratch_bytes(n, 254) <= need_with_table
    for (lo, m), before in got.items():
        assert _affine(oracle, dev.msm(scalars[lo:lo + m], 254, lo, m)) == _affine(oracle, before)
    dev.forget()          # forgetting twice is harmless


def test_table_batch_ex_mixes_tables_and_plain_bases(oracle, small_tables):
    """h2_dev_msm_batch_ex over columns whose bases have a table (full-width: table form; 16-bit: windowed form, the
    table saves nothing there) and columns over bases without one"""
    import torch

    L = h2.lib()
    n = 1 << 13
    tables = [oracle.random_g1(901, n), oracle.random_g1(902, n)]
    d_tab = [torch.from_numpy(t.view(np.int64)).cuda() for t in tables]
    assert L.h2_dev_bases_precompute(d_tab[0].data_ptr(), n, 0, None) == 0
    try:
        small = to_mont([(i * 2654435761) % 65536 for i in range(n)])
        z = oracle.random_fr(903, n)
        z[n // 8:] = z[5]
        cols = [oracle.random_fr(904, n), small, oracle.random_fr(905, n), z, z, oracle.random_fr(906, n)]
        which = [0, 0, 1, 0, 1, 0]
        bits = [254, 16, 254, 254, 254, 0]
        d_cols = [torch.from_numpy(np.ascontiguousarray(c).view(np.int64)).cuda() for c in cols]
        per = max((L.h2_msm_scratch_bytes(n, b) + 255) // 256 * 256 for b in bits)
        scratch = torch.empty(2 * per, dtype=torch.uint8, device="cuda")
        count = len(cols)
        sp = (ctypes.c_void_p * count)(*[t.data_ptr() for t in d_cols])
        bp = (ctypes.c_void_p * count)(*[d_tab[w].data_ptr() for w in which])
        bb = (ctypes.c_uint32 * count)(*bits)
        out = np.zeros((count, 12), dtype=np.uint64)
        assert L.h2_dev_msm_batch_ex(sp, bp, bb, count, n, scratch.data_ptr(), 2 * per, out.ctypes.data, None) == 0, L.h2_last_error()
        for j in range(count):
            want = (0, 0) if bits[j] == 0 else _affine(oracle, oracle.best_multiexp(cols[j], tables[which[j]]))
            assert _affine(oracle, out[j]) == want, j
        # the single-base batch entry point
        ptrs = (ctypes.c_void_p * 3)(*[d_cols[j].data_ptr() for j in (0, 3, 5)])
        out3 = np.zeros((3, 12), dtype=np.uint64)
        assert L.h2_dev_msm_batch(ptrs, 3, d_tab[0].data_ptr(), n, 254, scratch.data_ptr(), 2 * per, out3.ctypes.data, None) == 0
        for k, j in enumerate((0, 3, 5)):
            assert _affine(oracle, out3[k]) == _affine(oracle, oracle.best_multiexp(cols[j], tables[0])), j
    finally:
        assert L.h2_dev_bases_forget(d_tab[0].data_ptr()) == 0


def test_table_arguments(oracle):
    import torch

    L = h2.lib()
    pts = oracle.random_g1(77, 64)
    d = torch.from_numpy(pts.view(np.int64)).cuda()
    assert L.h2_dev_bases_precompute(d.data_ptr(), 64, 10, None) == 1       # digits out of range
    assert L.h2_dev_bases_precompute(d.data_ptr(), 64, 33, None) == 1
    assert L.h2_dev_bases_precompute(None, 64, 0, None) == 1
    assert L.h2_dev_bases_precompute(d.data_ptr(), 0, 0, None) == 0       # nothing to do
    assert L.h2_dev_bases_precompute_bytes(1 << 20, 13) == 13 * 64 << 20
    assert L.h2_dev_bases_forget(d.data_ptr()) == 0                       # never registered


@pytest.mark.timeout(600)
def test_table_msm_2p20(oracle):
    """BASELINE config 2's size over a table, uniform and proof-shaped columns"""
    n = 1 << 20
    pts = oracle.random_g1(2020, n)
    dev = DevMsm(pts)
    dev.precompute()
    try:
        s = oracle.random_fr(2022, n)
        assert _affine(oracle, dev.msm(s)) == _affine(oracle, oracle.best_multiexp(s, pts))
        z = oracle.random_fr(2021, n)
        z[n // 8:] = z[5]
        assert _affine(oracle, dev.msm(z)) == _affine(oracle, oracle.best_multiexp(z, pts))
    finally:
        dev.forget()


def test_table_is_dropped_when_the_bases_change(oracle, small_tables):
    """a table is keyed by the address of its bases; memory that comes back from the allocator at the same address
    with other points in it must not be committed against through the old table (sampled rows are compared with the
    table's copy of the bases before every use)"""
    n = 1 << 13
    old, new = oracle.random_g1(1001, n), oracle.random_g1(1002, n)
    scalars = oracle.random_fr(1003, n)
    dev = DevMsm(old)
    dev.precompute()
    assert _affine(oracle, dev.msm(scalars)) == _affine(oracle, oracle.best_multiexp(scalars, old))
    dev.d_pts.copy_(dev.torch.from_numpy(new.view(np.int64)).cuda())          # same address, other points
    dev.torch.cuda.synchronize()
    assert _affine(oracle, dev.msm(scalars)) == _affine(oracle, oracle.best_multiexp(scalars, new))
    assert _affine(oracle, dev.msm(scalars[:n // 2], 254, 0, n // 2)) == _affine(oracle, oracle.best_multiexp(scalars[:n // 2], new[:n // 2]))
    dev.precompute()                                                           # a fresh table serves the new points
    assert _affine(oracle, dev.msm(scalars)) == _affine(oracle, oracle.best_multiexp(scalars, new))
    dev.forget()
