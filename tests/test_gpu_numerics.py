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
