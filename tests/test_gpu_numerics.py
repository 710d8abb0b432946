"""GPU parity of the adjacent numerics: eval_polynomial, batch_invert and the
multiopen linear combination, against the CPU oracle / big integers.  Bit-exact."""
import ctypes

import numpy as np
import pytest

import halo2_gpu_specific_amd as h2
from halo2_gpu_specific_amd import arithmetic as ar
from h2util import R_MOD, _ptr, fr_mont, from_mont, to_mont

pytestmark = pytest.mark.gpu


def _oracle_eval(oracle, poly, x):
    out = np.zeros(4, dtype=np.uint64)
    oracle.lib.oracle_eval_polynomial(_ptr(poly), len(poly), _ptr(x), _ptr(out))
    return out


@pytest.mark.parametrize("n", [0, 1, 2, 255, 256, 257, 4095, 4096, 4097, 100000, 1 << 20, (1 << 24) + 5])
def test_eval_polynomial(oracle, n):
    poly = oracle.random_fr(1000 + (n % 977), n) if n else np.zeros((0, 4), dtype=np.uint64)
    x = oracle.random_fr(2000, 1)[0]
    got = ar.eval_polynomial(poly, x)
    if n <= 4097:  # also against plain big integers
        xv, acc = from_mont(x)[0], 0
        for c in reversed(from_mont(poly) if n else []):
            acc = (acc * xv + c) % R_MOD
        assert from_mont(got)[0] == acc
    assert np.array_equal(got, _oracle_eval(oracle, poly, x))


def test_eval_polynomial_special_points(oracle):
    poly = oracle.random_fr(5, 5000)
    for v in (0, 1, R_MOD - 1):
        x = fr_mont(v)
        assert np.array_equal(ar.eval_polynomial(poly, x), _oracle_eval(oracle, poly, x))


@pytest.mark.parametrize("n", [1, 3, 64, 255, 1000, 16384, 100001, 1 << 20])
def test_batch_invert(oracle, n):
    a = oracle.random_fr(3000 + (n % 991), n)
    a[::7] = 0  # zeros stay zero (ff::BatchInvert skips them)
    if n > 2:
        a[1] = fr_mont(1)
        a[2] = fr_mont(R_MOD - 1)
    want = a.copy()
    oracle.lib.oracle_batch_invert(_ptr(want), n)
    got = ar.batch_invert(a.copy())
    assert np.array_equal(got, want)
    if n <= 1000:
        for x, y in zip(from_mont(a), from_mont(got)):
            assert (x * y) % R_MOD == (1 if x else 0)


def test_lincomb(oracle):
    import torch

    L = h2.lib()
    dev = torch.device("cuda", 0)
    size, count = 5000, 11
    polys = [oracle.random_fr(4000 + i, size) for i in range(count)]
    coeffs = oracle.random_fr(4100, count)
    d_polys = [torch.from_numpy(p.view(np.int64)).to(dev) for p in polys]
    res = torch.zeros((size, 4), dtype=torch.int64, device=dev)
    ptrs = (ctypes.c_void_p * count)(*[t.data_ptr() for t in d_polys])
    torch.cuda.synchronize()          # torch's stream filled `res`; the library's own stream (NULL) writes it next
    assert L.h2_dev_lincomb(res.data_ptr(), ptrs, coeffs.ctypes.data, count, size, None) == 0
    torch.cuda.synchronize()
    h2.lib().h2_synchronize()
    got = res.cpu().numpy().view(np.uint64)
    pv, cv = [from_mont(p) for p in polys], from_mont(coeffs)
    want = [sum(c * p[i] for c, p in zip(cv, pv)) % R_MOD for i in range(size)]
    assert from_mont(got) == want
    # in-place on the first input (poly_batch = poly_batch * v + poly, gwc/prover.rs:52)
    v = oracle.random_fr(4200, 1)[0]
    one = fr_mont(1)
    two = np.stack([v, one])
    ptrs2 = (ctypes.c_void_p * 2)(d_polys[0].data_ptr(), d_polys[1].data_ptr())
    assert L.h2_dev_lincomb(d_polys[0].data_ptr(), ptrs2, two.ctypes.data, 2, size, None) == 0
    h2.lib().h2_synchronize()
    vv = from_mont(v)[0]
    assert from_mont(d_polys[0].cpu().numpy().view(np.uint64)) == [(a * vv + b) % R_MOD for a, b in zip(pv[0], pv[1])]
    # aliasing a later input is refused
    ptrs3 = (ctypes.c_void_p * 2)(d_polys[1].data_ptr(), d_polys[0].data_ptr())
    assert L.h2_dev_lincomb(d_polys[0].data_ptr(), ptrs3, two.ctypes.data, 2, size, None) == 1


@pytest.mark.parametrize("n", [2, 3, 5, 1024, 1025, 1026, 4097, 100003, (1 << 20) + 2, (1 << 22) + 1])
def test_kate_division(oracle, n):
    a = oracle.random_fr(5000 + (n % 983), n)
    b = oracle.random_fr(5100, 1)[0]
    want = np.zeros((n - 1, 4), dtype=np.uint64)
    oracle.lib.oracle_kate_division(_ptr(a), n, _ptr(b), _ptr(want))
    got = ar.kate_division(a, b)
    assert np.array_equal(got, want)
    if n <= 1026:  # the recurrence itself, in big integers: q[i] = a[i+1] + b*q[i+1]
        av, bv, q, nxt = from_mont(a), from_mont(b)[0], [0] * (n - 1), 0
        for i in range(n - 2, -1, -1):
            nxt = (av[i + 1] + bv * nxt) % R_MOD
            q[i] = nxt
        assert from_mont(got) == q


@pytest.mark.parametrize("n", [1, 2, 3, 1024, 1025, 1026, 5000, (1 << 20) + 7])
def test_prefix_product(oracle, n):
    f = oracle.random_fr(6000 + (n % 971), max(n - 1, 1))[: n - 1]
    init = oracle.random_fr(6100, 1)[0]
    got = ar.prefix_product(f, init, n)
    z = [from_mont(init)[0]]
    for v in from_mont(f) if n > 1 else []:
        z.append(z[-1] * v % R_MOD)
    assert from_mont(got) == z


# ---- the permutation / logup building blocks of the device prover (csrc/poly.hip, scan.hip, logup.hip) ---------
def _dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()


def _host(t):
    return t.cpu().numpy().view(np.uint64)


@pytest.mark.parametrize("n", [1, 2, 5, 1024, 1025, 5000, 300001])
def test_prefix_sum(oracle, n):
    """z[0] = init, z[i] = z[i-1] + f[i-1] (logup/prover.rs:353-367) against Python integers"""
    L = h2.lib()
    f = oracle.random_fr(4100 + n % 13, n)
    init = 0x1234567 if n % 2 else 0
    d_f, d_z = _dev(f), _dev(np.zeros((n, 4), dtype=np.uint64))
    init_m = fr_mont(init)                  # (a temporary's address would dangle by the time the call runs)
    assert L.h2_dev_prefix_sum(d_f.data_ptr(), n, init_m.ctypes.data, d_z.data_ptr(), None) == 0
    got = from_mont(_host(d_z))
    fv = from_mont(f)
    acc, want = init, []
    for i in range(n):
        want.append(acc)
        acc = (acc + fv[i]) % R_MOD
    assert got == want


@pytest.mark.parametrize("usable,n", [(10, 16), (250, 256), (4090, 4096), (100000, 1 << 17)])
def test_logup_multiplicity(oracle, usable, n):
    """m[row] = how often the table row's value occurs among the first `usable` rows of the inputs, duplicates
    credited to the lowest row; rows past `usable` are ignored and left zero; a missing value is an error"""
    import random

    L = h2.lib()
    rnd = random.Random(usable)
    table = oracle.random_fr(77, n)
    for _ in range(max(1, usable // 8)):                       # duplicated table values
        table[rnd.randrange(usable)] = table[rnd.randrange(usable)]
    table[usable:] = table[0]                                  # rows past usable must not be credited
    inputs = []
    for j in range(3):
        idx = np.array([rnd.randrange(usable) if rnd.random() < 0.7 else 3 for _ in range(n)])   # row 3 is hot
        inputs.append(table[idx].copy())
    d_table, d_in = _dev(table), [_dev(a) for a in inputs]
    d_m = _dev(np.zeros((n, 4), dtype=np.uint64))
    nbytes = L.h2_logup_scratch_bytes(n)
    import torch

    scratch = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    ptrs = (ctypes.c_void_p * 3)(*[t.data_ptr() for t in d_in])
    rc = L.h2_dev_logup_multiplicity(d_table.data_ptr(), ptrs, 3, usable, n, d_m.data_ptr(), scratch.data_ptr(), nbytes, None)
    assert rc == 0, L.h2_last_error()
    first = {}
    keys = [tuple(int(x) for x in r) for r in table]
    for i in range(usable):
        first.setdefault(keys[i], i)
    want = [0] * n
    for a in inputs:
        for i in range(usable):
            want[first[tuple(int(x) for x in a[i])]] += 1
    assert from_mont(_host(d_m)) == want
    # one absent value
    inputs[1][usable // 2] = fr_mont(0xDEADBEEF)
    d_bad = _dev(inputs[1])
    ptrs = (ctypes.c_void_p * 1)(d_bad.data_ptr())
    rc = L.h2_dev_logup_multiplicity(d_table.data_ptr(), ptrs, 1, usable, n, d_m.data_ptr(), scratch.data_ptr(), nbytes, None)
    assert rc == 1 and b"missing from the table" in L.h2_last_error()
    assert L.h2_dev_logup_multiplicity(d_table.data_ptr(), ptrs, 1, usable, n, d_m.data_ptr(), scratch.data_ptr(), 16, None) == 1


@pytest.mark.parametrize("n", [8, 256, 2048, 2049, 70000])
def test_permutation_sigma_and_terms(oracle, n):
    """sigma[j] = DELTA^c * omega^r; num / den products of permutation/prover.rs:89-128, against Python integers"""
    import random

    import torch

    L = h2.lib()
    rnd = random.Random(n)
    DELTA = 0x09226B6E22C6F0CA64EC26AAD4C86E715B5F898E5E963F25870E56BBE533E9A2
    omega = from_mont(oracle.random_fr(9, 1))[0]
    mc = np.array([rnd.randrange(5) for _ in range(n)], dtype=np.uint32)
    mr = np.array([rnd.randrange(n) for _ in range(n)], dtype=np.uint32)
    d_mc, d_mr = torch.from_numpy(mc.view(np.int32)).cuda(), torch.from_numpy(mr.view(np.int32)).cuda()
    d_sig = _dev(np.zeros((n, 4), dtype=np.uint64))
    delta_m, omega_m = fr_mont(DELTA), fr_mont(omega)
    assert L.h2_dev_permutation_sigma(d_sig.data_ptr(), d_mc.data_ptr(), d_mr.data_ptr(), n, delta_m.ctypes.data,
                                      omega_m.ctypes.data, None) == 0
    assert L.h2_synchronize() == 0          # stream NULL = the library's own (non-blocking) stream
    sigma = from_mont(_host(d_sig))
    if n <= 2049:
        assert sigma == [pow(DELTA, int(c), R_MOD) * pow(omega, int(r), R_MOD) % R_MOD for c, r in zip(mc, mr)]
    beta, gamma = rnd.randrange(R_MOD), rnd.randrange(R_MOD)
    vals = [oracle.random_fr(20 + j, n) for j in range(2)]
    d_num, d_den = _dev(np.zeros((n, 4), dtype=np.uint64)), _dev(np.zeros((n, 4), dtype=np.uint64))
    want_num, want_den = [1] * n, [1] * n
    for j in range(2):
        dp = pow(DELTA, j + 3, R_MOD)
        d_v = _dev(vals[j])
        beta_m, gamma_m, dp_m = fr_mont(beta), fr_mont(gamma), fr_mont(dp)
        assert L.h2_dev_permutation_terms(d_num.data_ptr(), d_den.data_ptr(), d_v.data_ptr(), d_sig.data_ptr(), n,
                                          beta_m.ctypes.data, gamma_m.ctypes.data, dp_m.ctypes.data,
                                          omega_m.ctypes.data, 1 if j == 0 else 0, None) == 0
        v = from_mont(vals[j])
        w = 1
        for i in range(n):
            want_num[i] = want_num[i] * (dp * w % R_MOD * beta + gamma + v[i]) % R_MOD
            want_den[i] = want_den[i] * (beta * sigma[i] + gamma + v[i]) % R_MOD
            w = w * omega % R_MOD
    assert L.h2_synchronize() == 0
    assert from_mont(_host(d_num)) == want_num and from_mont(_host(d_den)) == want_den


def test_random_fr_matches_host_twin():
    """h2_dev_random_fr (ChaCha20 keystream -> 506 bits -> mod r, Montgomery) against rng.py's numpy twin"""
    from halo2_gpu_specific_amd.rng import ProverRng

    L = h2.lib()
    for key, n in ((bytes(32), 1), (bytes(range(32)), 1000), (b"\xff" * 32, 1 << 16)):
        d = _dev(np.zeros((n, 4), dtype=np.uint64))
        assert L.h2_dev_random_fr(key, n, d.data_ptr(), None) == 0 and L.h2_synchronize() == 0
        assert np.array_equal(_host(d), ProverRng.random_poly_limbs(key, n))


def test_eval_polynomial_batch(oracle):
    L = h2.lib()
    for n, count in ((1, 3), (300, 5), (5000, 7), (1 << 18, 4)):
        polys = [oracle.random_fr(700 + j, n) for j in range(count)]
        pts = oracle.random_fr(800 + n % 97, count)
        d = [_dev(p) for p in polys]
        ptrs = (ctypes.c_void_p * count)(*[t.data_ptr() for t in d])
        out = np.zeros((count, 4), dtype=np.uint64)
        assert L.h2_dev_eval_polynomial_batch(ptrs, count, n, pts.ctypes.data, out.ctypes.data, None) == 0
        for j in range(count):
            assert np.array_equal(out[j], _oracle_eval(oracle, polys[j], pts[j])), (n, j)
    assert L.h2_dev_eval_polynomial_batch(None, 0, 5, None, None, None) == 0


@pytest.mark.parametrize("n", [1, 63, 1000, 70001])
def test_max_scalar_bits_of_column_groups(oracle, n):
    """find_max_scalar_bits (plonk/prover.rs:237-254) for a group of canonical columns in one launch: the bit length of
    each column's largest value, 0 for an all-zero column; more than 16 columns take several launches"""
    import torch

    L = h2.lib()
    rows = np.arange(n, dtype=np.uint64)
    cols, want = [], []
    for bits in (0, 1, 7, 16, 31, 32, 33, 64, 65, 130, 200, 253, 254, 12, 1, 40, 90, 255):   # 18 columns
        top = min(bits, 254)
        vals = [0] * n
        if bits:
            for i in range(n):
                vals[i] = (int(rows[i]) * 0x9E3779B97F4A7C15F39CC0605CEDC8341082276BF3A27251 + 12345) % (1 << max(top - 1, 0) or 1)
            vals[(7 * n) // 11] = (1 << (top - 1)) | 1 if top > 1 else 1      # one value with the top bit set
        a = np.zeros((n, 4), dtype=np.uint64)
        for limb in range(4):
            a[:, limb] = np.array([(v >> (64 * limb)) & (2 ** 64 - 1) for v in vals], dtype=np.uint64)
        cols.append(_dev(a))
        want.append(max(v.bit_length() for v in vals))
    count = len(cols)
    ptrs = (ctypes.c_void_p * count)(*[c.data_ptr() for c in cols])
    out = (ctypes.c_uint32 * count)()
    words = torch.empty((count, 8), dtype=torch.int32, device="cuda")
    assert L.h2_dev_max_scalar_bits(ptrs, count, n, words.data_ptr(), out, None) == 0, L.h2_last_error()
    assert list(out) == want
    assert L.h2_dev_max_scalar_bits(ptrs, 0, n, None, None, None) == 0


def test_device_memory_and_stream_helpers_without_torch(oracle):
    """h2_dev_alloc / h2_dev_upload / h2_stream_* / h2_dev_download: a host without a HIP binding of its own (the Rust side of
    INTEGRATION.md) drives the device-resident API with these alone -- an NTT and its inverse on a library-allocated
    buffer and a library-created stream, pinned staging memory, against the oracle"""
    import ctypes

    import halo2_gpu_specific_amd as h2
    from h2util import R_MOD, fr_mont

    L = h2.lib()
    log_n = 14
    n = 1 << log_n
    root = 0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C
    omega = pow(root, 1 << (28 - log_n), R_MOD)
    x = oracle.random_fr(321, n)
    want = oracle.best_fft(x, fr_mont(omega), log_n)
    nbytes = 32 * n
    d_a, d_tmp, stream, pinned = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    assert L.h2_dev_alloc(nbytes, ctypes.byref(d_a)) == 0 and L.h2_dev_alloc(nbytes, ctypes.byref(d_tmp)) == 0
    assert L.h2_stream_create(ctypes.byref(stream)) == 0 and L.h2_host_alloc_pinned(nbytes, ctypes.byref(pinned)) == 0
    try:
        ctypes.memmove(pinned, x.ctypes.data, nbytes)
        assert L.h2_dev_upload(d_a, pinned, nbytes, stream) == 0
        w = fr_mont(omega)
        assert L.h2_dev_ntt(d_a, d_tmp, w.ctypes.data, log_n, stream) == 0, L.h2_last_error()
        got = np.zeros_like(x)
        assert L.h2_dev_download(got.ctypes.data, d_a, nbytes, stream) == 0
        assert np.array_equal(got, want)
        wi, ninv = fr_mont(pow(omega, -1, R_MOD)), fr_mont(pow(n, -1, R_MOD))
        assert L.h2_dev_intt(d_a, d_tmp, wi.ctypes.data, ninv.ctypes.data, log_n, stream) == 0
        assert L.h2_stream_synchronize(stream) == 0
        assert L.h2_dev_download(got.ctypes.data, d_a, nbytes, None) == 0
        assert np.array_equal(got, x)
    finally:
        assert L.h2_dev_free(d_a) == 0 and L.h2_dev_free(d_tmp) == 0 and L.h2_dev_free(None) == 0
        assert L.h2_stream_destroy(stream) == 0 and L.h2_host_free_pinned(pinned) == 0


def test_constant_operand_product_on_the_device_including_short_quotients(tmp_path):
    """fp_mul_const (csrc/field.hpp, round 6: the NTT's twiddle products) as compiled gfx950 code against host arithmetic:
    65 536 random operand triples (ANY 256-bit x) and 4 096 built so that the truncated quotient IS one short -- the branch
    the transforms take about once in 2^29 products: raw result = exact or exact + p, the value the passes use is below 2p and
    the same residue.  tests/fp_mul_const_check.hip, compiled here with hipcc (the schedule itself is executed line by line
    with Python integers in tests/test_fp_mul_schedule.py)."""
    import os
    import subprocess

    here = os.path.dirname(os.path.abspath(__file__))
    exe = str(tmp_path / "fp_mul_const_check")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "--offload-arch=gfx950", os.path.join(here, "fp_mul_const_check.hip"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln for ln in out.stdout.splitlines() if "fp_mul_const device vs host" in ln]
    assert len(lines) == 2 and all(": 0 mismatches, 0 results at or above 2p" in ln for ln in lines), out.stdout
