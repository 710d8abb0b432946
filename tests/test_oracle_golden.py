"""Pin the C oracle (oracle/oracle.c) against the Python big-integer golden vectors
(tests/golden/*.json, produced by tests/golden/gen_golden.py).  CPU only."""
import numpy as np
import pytest

from h2util import (
    Q_MOD,
    R_MOD,
    arr_to_points,
    from_mont,
    golden_points,
    h2i,
    load_golden,
    points_to_arr,
    to_mont,
)


def test_constants():
    c = load_golden("constants.json")
    assert h2i(c["r"]) == R_MOD and h2i(c["q"]) == Q_MOD
    assert h2i(c["root_of_unity"]) == 0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C
    assert h2i(c["fr_inv64"]) == 0xC2E1F593EFFFFFFF and h2i(c["fq_inv64"]) == 0x87D20782E4866389


def test_field_kat(oracle):
    for case in load_golden("field_kat.json"):
        f, p = case["field"], (R_MOD if case["field"] == "fr" else Q_MOD)
        a, b = h2i(case["a"]), h2i(case["b"])
        am, bm = to_mont([a], p)[0], to_mont([b], p)[0]
        assert from_mont(am, p)[0] == a
        assert [int(x) for x in am] == [(h2i(case["mont_a"]) >> (64 * i)) & (2**64 - 1) for i in range(4)]
        assert from_mont(oracle.op2(f"oracle_{f}_mul", am, bm), p)[0] == h2i(case["mul"])
        assert from_mont(oracle.op2(f"oracle_{f}_add", am, bm), p)[0] == h2i(case["add"])
        assert from_mont(oracle.op2(f"oracle_{f}_sub", am, bm), p)[0] == h2i(case["sub"])
        assert from_mont(oracle.op1(f"oracle_{f}_inv", am), p)[0] == h2i(case["inv_a"])


def test_repr_batch(oracle):
    vals = [0, 1, R_MOD - 1, 12345678901234567890123456789]
    from h2util import _ptr, ints_to_arr

    a = ints_to_arr(vals)
    oracle.lib.oracle_from_repr_batch(_ptr(a), len(vals), 0)
    assert from_mont(a) == vals
    oracle.lib.oracle_to_repr_batch(_ptr(a), len(vals), 0)
    assert [int(x) for x in a[3]] == [(vals[3] >> (64 * i)) & (2**64 - 1) for i in range(4)]


@pytest.mark.parametrize("which", ["best_fft", "best_fft_st", "best_fft_1thread"])
def test_ntt_kat(oracle, which):
    for case in load_golden("ntt_kat.json"):
        log_n = case["log_n"]
        x = to_mont([h2i(v) for v in case["input"]])
        omega = to_mont([h2i(case["omega"])])[0]
        if which == "best_fft":
            out = oracle.best_fft(x, omega, log_n)
        elif which == "best_fft_1thread":
            out = oracle.best_fft(x, omega, log_n, threads=1)
        else:
            out = oracle.best_fft_st(x, omega, log_n)
        assert from_mont(out) == [h2i(v) for v in case["output"]], f"log_n={log_n}"
        # inverse: poly/domain.rs:400-410
        back = oracle.ifft(out, to_mont([h2i(case["omega_inv"])])[0], log_n, to_mont([h2i(case["n_inv"])])[0])
        assert np.array_equal(back, x)


@pytest.mark.parametrize("fn", ["best_multiexp", "multiexp_serial", "small_multiexp"])
def test_msm_kat(oracle, fn):
    for case in load_golden("msm_kat.json"):
        scalars = to_mont([h2i(s) for s in case["scalars"]]).reshape(-1, 4)
        pts = points_to_arr(golden_points(case["points"])).reshape(-1, 8)
        want = golden_points([case["result"]])[0]
        if fn == "best_multiexp":
            for threads in (1, 3, 8):
                got = arr_to_points(oracle.to_affine(oracle.best_multiexp(scalars, pts, threads=threads)))[0]
                assert got == want, (case["name"], threads)
        else:
            got = arr_to_points(oracle.to_affine(getattr(oracle, fn)(scalars, pts)))[0]
            assert got == want, case["name"]


def test_g1_kat(oracle):
    kat = load_golden("g1_kat.json")
    gen = points_to_arr([(1, 2)])[0]
    assert oracle.lib.oracle_g1_on_curve(gen.ctypes.data) == 1
    for case in kat:
        if "k" in case:
            got = arr_to_points(oracle.to_affine(oracle.g1_mul(gen, to_mont([h2i(case["k"])])[0])))[0]
            assert got == golden_points([case["kG"]])[0]
        else:
            p, q = points_to_arr(golden_points([case["p"]]))[0], points_to_arr(golden_points([case["q"]]))[0]
            one = to_mont([1], Q_MOD)[0]
            pj = np.concatenate([p, one])
            out = np.zeros(12, dtype=np.uint64)
            oracle.lib.oracle_g1_add_affine(pj.ctypes.data, q.ctypes.data, out.ctypes.data)
            assert arr_to_points(oracle.to_affine(out))[0] == golden_points([case["p_plus_q"]])[0]
            oracle.lib.oracle_g1_add_affine(pj.ctypes.data, p.ctypes.data, out.ctypes.data)  # P + P -> doubling branch
            assert arr_to_points(oracle.to_affine(out))[0] == golden_points([case["two_p"]])[0]
            oracle.lib.oracle_g1_double(pj.ctypes.data, out.ctypes.data)
            assert arr_to_points(oracle.to_affine(out))[0] == golden_points([case["two_p"]])[0]


def test_domain_kat(oracle):
    for case in load_golden("domain_kat.json"):
        d, t = oracle.domain(case["j"], case["k"])
        assert d.extended_k == case["extended_k"] and d.quotient_poly_degree == case["quotient_poly_degree"]
        for name in (
            "omega",
            "omega_inv",
            "extended_omega",
            "extended_omega_inv",
            "g_coset",
            "g_coset_inv",
            "ifft_divisor",
            "extended_ifft_divisor",
            "barycentric_weight",
        ):
            assert from_mont(d.fr(name))[0] == h2i(case[name]), (case["j"], case["k"], name)
        assert from_mont(t) == [h2i(v) for v in case["t_evaluations"]]


def test_coset_kat(oracle):
    for case in load_golden("coset_kat.json"):
        d, _ = oracle.domain(case["j"], case["k"])
        coeffs = to_mont([h2i(v) for v in case["coeffs"]])
        ext = oracle.coeff_to_extended(coeffs, d)
        assert from_mont(ext) == [h2i(v) for v in case["extended"]]
        # round trip: extended_to_coeff truncates to n*(j-1) >= n coefficients
        back = oracle.extended_to_coeff(ext, d)
        n = 1 << case["k"]
        assert len(back) == n * (case["j"] - 1)
        assert np.array_equal(back[:n], coeffs)
        assert not back[n:].any()


def test_setup_kat(oracle):
    case = load_golden("setup_kat.json")
    n = 1 << case["k"]
    g = np.zeros((n, 8), dtype=np.uint64)
    gl = np.zeros((n, 8), dtype=np.uint64)
    s = to_mont([h2i(case["s"])])[0]
    oracle.lib.oracle_unsafe_setup(case["k"], s.ctypes.data, g.ctypes.data, gl.ctypes.data)
    assert arr_to_points(g) == golden_points(case["g"])
    assert arr_to_points(gl) == golden_points(case["g_lagrange"])


def test_random_generators_are_deterministic_and_valid(oracle):
    a = oracle.random_fr(7, 64)
    b = oracle.random_fr(7, 64)
    assert np.array_equal(a, b) and len(set(from_mont(a))) == 64
    assert all(v < R_MOD for v in from_mont(a))
    p = oracle.random_g1(9, 32)
    assert np.array_equal(p, oracle.random_g1(9, 32))
    for x, y in arr_to_points(p):
        assert (y * y - x * x * x - 3) % Q_MOD == 0


def test_point_codec_against_the_big_integer_encoder(oracle):
    """oracle_points_{compress, decompress} (the checker of the device codec) against ref_plonk.point_to_bytes /
    point_from_bytes on random points, both parities, the identity, and invalid encodings"""
    import ref_plonk as rp
    from h2util import Q_MOD, arr_to_points, points_to_arr

    pts = oracle.random_g1(77, 600)
    ints = arr_to_points(pts)
    ints[5] = (0, 0)                                   # the identity
    pts = points_to_arr(ints)
    raw = oracle.points_compress(pts)
    want = [rp.point_to_bytes(None if P == (0, 0) else P) for P in ints]
    assert [bytes(r) for r in raw] == want
    assert {b[31] >> 7 for b in want} == {0, 1}        # both parities occur
    back, bad = oracle.points_decompress(raw)
    assert bad == 0 and np.array_equal(back, pts)
    for i in (0, 1, 2, 5, 599):
        P = rp.point_from_bytes(want[i])
        assert (P or (0, 0)) == ints[i]
    # not on the curve / non-canonical x / the flagged zero
    x = 1
    while pow((x ** 3 + 3) % Q_MOD, (Q_MOD - 1) // 2, Q_MOD) == 1:
        x += 1
    junk = np.zeros((3, 32), dtype=np.uint8)
    junk[0] = np.frombuffer(x.to_bytes(32, "little"), dtype=np.uint8)
    junk[1] = np.frombuffer((Q_MOD + 1).to_bytes(32, "little"), dtype=np.uint8)
    junk[2, 31] = 0x80
    out, bad = oracle.points_decompress(junk)
    assert bad == 3 and not out.any()
