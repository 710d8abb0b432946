"""The reference's own relational unit tests for this path, restated against the C oracle with
seeded inputs instead of OsRng.  CPU only."""
import numpy as np

from h2util import R_MOD, _ptr, arr_to_points, fr_mont, from_mont, to_mont


def _eval(oracle, poly, x):
    out = np.zeros(4, dtype=np.uint64)
    oracle.lib.oracle_eval_polynomial(_ptr(poly), len(poly), _ptr(x), _ptr(out))
    return out


def test_commit_lagrange(oracle):
    """poly/commitment.rs:481-495: commit(ifft(a)) == commit_lagrange(a), a[i] = i, k = 6"""
    k = 6
    n = 1 << k
    s = fr_mont(0xDEADBEEFCAFEBABE0123456789ABCDEF)
    g = np.zeros((n, 8), dtype=np.uint64)
    gl = np.zeros((n, 8), dtype=np.uint64)
    oracle.lib.oracle_unsafe_setup(k, _ptr(s), _ptr(g), _ptr(gl))
    d, _ = oracle.domain(1, k)
    a = to_mont(list(range(n)))
    b = oracle.ifft(a, d.fr("omega_inv"), k, d.fr("ifft_divisor"))  # lagrange_to_coeff
    lhs = oracle.best_multiexp(b, g)
    rhs = oracle.best_multiexp(a, gl)
    assert oracle.lib.oracle_g1_eq(_ptr(lhs), _ptr(rhs)) == 1
    assert arr_to_points(oracle.to_affine(lhs)) == arr_to_points(oracle.to_affine(rhs))


def test_rotate(oracle):
    """poly/domain.rs:551-589: p(omega x) <-> rotate-left-by-1 in the Lagrange basis, k = 3"""
    d, _ = oracle.domain(1, 3)
    poly = oracle.random_fr(41, 8)
    x = oracle.random_fr(42, 1)[0]
    coeffs = lambda v: oracle.ifft(v, d.fr("omega_inv"), 3, d.fr("ifft_divisor"))
    cur, nxt, prv = np.roll(poly, 0, axis=0), np.roll(poly, -1, axis=0), np.roll(poly, 1, axis=0)  # poly.rs:219-233
    p = coeffs(poly)
    xw = oracle.op2("oracle_fr_mul", x, d.fr("omega"))
    xwi = oracle.op2("oracle_fr_mul", x, d.fr("omega_inv"))
    assert np.array_equal(_eval(oracle, p, x), _eval(oracle, coeffs(cur), x))
    assert np.array_equal(_eval(oracle, p, xw), _eval(oracle, coeffs(nxt), x))
    assert np.array_equal(_eval(oracle, p, xwi), _eval(oracle, coeffs(prv), x))


def test_lagrange_interpolate(oracle):
    """arithmetic.rs:933-950"""
    points, evals = oracle.random_fr(51, 5), oracle.random_fr(52, 5)
    for m in range(1, 6):
        poly = np.zeros((m, 4), dtype=np.uint64)
        oracle.lib.oracle_lagrange_interpolate(_ptr(points), _ptr(evals), m, _ptr(poly))
        for i in range(m):
            assert np.array_equal(_eval(oracle, poly, points[i]), evals[i])


def test_l_i(oracle):
    """poly/domain.rs:592-619: l_i_range vs lagrange_interpolate of unit vectors, k = 3"""
    import ctypes

    d, _ = oracle.domain(1, 3)
    w = from_mont(d.fr("omega"))[0]
    points = to_mont([pow(w, i, R_MOD) for i in range(8)])
    ls = []
    for i in range(8):
        unit = to_mont([1 if j == i else 0 for j in range(8)])
        poly = np.zeros((8, 4), dtype=np.uint64)
        oracle.lib.oracle_lagrange_interpolate(_ptr(points), _ptr(unit), 8, _ptr(poly))
        ls.append(poly)
    x = oracle.random_fr(61, 1)[0]
    xn = fr_mont(pow(from_mont(x)[0], 8, R_MOD))
    rots = np.arange(-7, 8, dtype=np.int32)
    res = np.zeros((15, 4), dtype=np.uint64)
    oracle.lib.oracle_l_i_range(ctypes.byref(d), _ptr(x), _ptr(xn), _ptr(rots), 15, _ptr(res))
    for i in range(8):
        assert np.array_equal(_eval(oracle, ls[i], x), res[7 + i])
        assert np.array_equal(_eval(oracle, ls[(8 - i) % 8], x), res[7 - i])


def test_kate_division_and_batch_invert(oracle):
    n = 33
    a = oracle.random_fr(71, n)
    b = oracle.random_fr(72, 1)[0]
    av, bv = from_mont(a), from_mont(b)[0]
    r = 0
    for c in reversed(av):  # make a(b) == 0 so the division is exact (kate_division: no remainder)
        r = (r * bv + c) % R_MOD
    av[0] = (av[0] - r) % R_MOD
    a = to_mont(av)
    q = np.zeros((n - 1, 4), dtype=np.uint64)
    oracle.lib.oracle_kate_division(_ptr(a), n, _ptr(b), _ptr(q))
    qv = from_mont(q)
    # (X - b) * q == a
    prod = [0] * n
    for i, c in enumerate(qv):
        prod[i + 1] = (prod[i + 1] + c) % R_MOD
        prod[i] = (prod[i] - bv * c) % R_MOD
    assert prod == av
    v = oracle.random_fr(73, 17)
    v[3] = 0
    inv = v.copy()
    oracle.lib.oracle_batch_invert(_ptr(inv), len(inv))
    for x, y in zip(from_mont(v), from_mont(inv)):
        assert (x * y) % R_MOD == (1 if x else 0)
