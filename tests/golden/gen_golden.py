#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ with plain Python big integers.

This script is the *independent* pin for the C oracle (oracle/oracle.c): nothing
here shares code with it -- no limbs, no Montgomery form in the computation, naive
O(n^2) DFT, affine double-and-add.  The reference (Rust) holds no literal golden
vectors for this path and cannot be built or imported in this image (SURVEY.md
section 0 items 3-4), so the vectors pin the oracle to the mathematics and to the
public BN254 constants instead of to reference outputs.

Run:  python tests/golden/gen_golden.py        (rewrites tests/golden/*.json)

All field values are written as canonical integers in hex ("0x...").  The tests
convert to/from the 4 x u64 Montgomery layout themselves.
"""
import json
import os
import random

R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001  # Fr
Q_MOD = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47  # Fq
S = 28
ROOT_OF_UNITY = pow(7, (R_MOD - 1) >> S, R_MOD)
ZETA = 0x30644E72E131A029048B6E193FD84104CC37A73FEC2BC5E9B8CA0B2D36636F23
B = 3
HERE = os.path.dirname(os.path.abspath(__file__))


def hx(v):
    return hex(v)


# ---------------------------------------------------------------- curve (affine, ints)
def ec_add(p, q):
    if p is None:
        return q
    if q is None:
        return p
    x1, y1 = p
    x2, y2 = q
    if x1 == x2:
        if (y1 + y2) % Q_MOD == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, Q_MOD) % Q_MOD
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, Q_MOD) % Q_MOD
    x3 = (lam * lam - x1 - x2) % Q_MOD
    y3 = (lam * (x1 - x3) - y1) % Q_MOD
    return (x3, y3)


def ec_mul(k, p):
    acc = None
    k %= R_MOD
    for bit in bin(k)[2:] if k else "":
        acc = ec_add(acc, acc)
        if bit == "1":
            acc = ec_add(acc, p)
    return acc


def ec_neg(p):
    return None if p is None else (p[0], (-p[1]) % Q_MOD)


def pt_json(p):
    # identity is encoded as (0, 0), matching the assumed G1Affine layout
    return [hx(0), hx(0)] if p is None else [hx(p[0]), hx(p[1])]


def random_point(rng):
    while True:
        x = rng.randrange(Q_MOD)
        rhs = (x * x * x + B) % Q_MOD
        y = pow(rhs, (Q_MOD + 1) // 4, Q_MOD)
        if y * y % Q_MOD == rhs:
            if rng.getrandbits(1):
                y = (-y) % Q_MOD
            return (x, y)


# ---------------------------------------------------------------- generators
def gen_constants():
    out = {
        "r": hx(R_MOD),
        "q": hx(Q_MOD),
        "S": S,
        "root_of_unity": hx(ROOT_OF_UNITY),
        "zeta": hx(ZETA),
        "zeta_sq": hx(ZETA * ZETA % R_MOD),
        "delta": hx(pow(7, 1 << S, R_MOD)),
        "fr_R": hx((1 << 256) % R_MOD),
        "fr_R2": hx(pow(1 << 256, 2, R_MOD)),
        "fr_inv64": hx((-pow(R_MOD, -1, 1 << 64)) % (1 << 64)),
        "fq_R": hx((1 << 256) % Q_MOD),
        "fq_R2": hx(pow(1 << 256, 2, Q_MOD)),
        "fq_inv64": hx((-pow(Q_MOD, -1, 1 << 64)) % (1 << 64)),
        "generator": [hx(1), hx(2)],
    }
    assert pow(ROOT_OF_UNITY, 1 << S, R_MOD) == 1 and pow(ROOT_OF_UNITY, 1 << (S - 1), R_MOD) != 1
    assert pow(ZETA, 3, R_MOD) == 1 and ZETA != 1
    return out


def gen_field(rng):
    cases = []
    for name, p in (("fr", R_MOD), ("fq", Q_MOD)):
        specials = [0, 1, 2, p - 1, p - 2, (1 << 253), (1 << 64) - 1, 1 << 64]
        vals = specials + [rng.randrange(p) for _ in range(24)]
        for i in range(len(vals)):
            a, b = vals[i], vals[(i * 7 + 3) % len(vals)]
            cases.append(
                {
                    "field": name,
                    "a": hx(a),
                    "b": hx(b),
                    "add": hx((a + b) % p),
                    "sub": hx((a - b) % p),
                    "mul": hx(a * b % p),
                    "inv_a": hx(pow(a, -1, p) if a else 0),
                    "mont_a": hx(a * (1 << 256) % p),
                }
            )
    return cases


def naive_dft(x, omega):
    n = len(x)
    pw = [1] * n
    for i in range(1, n):
        pw[i] = pw[i - 1] * omega % R_MOD
    return [sum(x[j] * pw[(j * k) % n] for j in range(n)) % R_MOD for k in range(n)]


def gen_ntt(rng):
    cases = []
    for log_n in (1, 2, 3, 4, 5, 8, 10):
        n = 1 << log_n
        omega = pow(ROOT_OF_UNITY, 1 << (S - log_n), R_MOD)
        x = [rng.randrange(R_MOD) for _ in range(n)]
        if log_n == 3:
            x[0], x[1], x[2] = 0, 1, R_MOD - 1
        fwd = naive_dft(x, omega)
        n_inv = pow(n, -1, R_MOD)
        # inverse check of the generator itself
        back = [v * n_inv % R_MOD for v in naive_dft(fwd, pow(omega, -1, R_MOD))]
        assert back == x
        cases.append(
            {
                "log_n": log_n,
                "omega": hx(omega),
                "omega_inv": hx(pow(omega, -1, R_MOD)),
                "n_inv": hx(n_inv),
                "input": [hx(v) for v in x],
                "output": [hx(v) for v in fwd],
            }
        )
    return cases


def gen_msm(rng):
    cases = []
    G = (1, 2)

    def add_case(name, scalars, points):
        acc = None
        for s, p in zip(scalars, points):
            acc = ec_add(acc, ec_mul(s, p))
        cases.append(
            {
                "name": name,
                "scalars": [hx(s) for s in scalars],
                "points": [pt_json(p) for p in points],
                "result": pt_json(acc),
            }
        )

    add_case("empty", [], [])
    for n in (1, 2, 3, 4, 5, 31, 32, 33, 100, 257):
        pts = [random_point(rng) for _ in range(n)]
        sc = [rng.randrange(R_MOD) for _ in range(n)]
        add_case("random_%d" % n, sc, pts)
    # edge cases the domain offers (SURVEY.md 8(d) config 2, item iv)
    p0, p1 = random_point(rng), random_point(rng)
    add_case("scalars_0_1_rm1", [0, 1, R_MOD - 1, 0], [p0, p1, p0, p1])
    add_case("all_zero_scalars", [0] * 8, [random_point(rng) for _ in range(8)])
    add_case("duplicate_points", [rng.randrange(R_MOD) for _ in range(6)], [p0] * 6)
    add_case("duplicate_points_same_scalar", [5] * 6, [p0] * 6)
    add_case("p_and_neg_p", [7, 7, 3, 3], [p0, ec_neg(p0), p1, ec_neg(p1)])
    add_case("cancels_to_identity", [11, 11], [p1, ec_neg(p1)])
    add_case("identity_points", [5, 6, 7], [None, p0, None])
    add_case("generator_small", [1, 2, 3], [G, G, G])
    small = [rng.randrange(1 << 16) for _ in range(64)]
    add_case("small_16bit_scalars", small, [random_point(rng) for _ in range(64)])
    half_zero = [0 if i % 2 else rng.randrange(R_MOD) for i in range(40)]
    add_case("half_zero", half_zero, [random_point(rng) for _ in range(40)])
    return cases


def gen_g1(rng):
    G = (1, 2)
    out = []
    for k in (1, 2, 3, 4, 5, 0xFFFF, R_MOD - 1, rng.randrange(R_MOD), rng.randrange(R_MOD)):
        out.append({"k": hx(k), "kG": pt_json(ec_mul(k, G))})
    p, q2 = random_point(rng), random_point(rng)
    out.append({"p": pt_json(p), "q": pt_json(q2), "p_plus_q": pt_json(ec_add(p, q2)), "two_p": pt_json(ec_add(p, p))})
    return out


def gen_domain():
    out = []
    for j, k in ((1, 3), (2, 3), (3, 4), (4, 5), (5, 6), (3, 8), (9, 4), (3, 20), (5, 22), (3, 24)):
        qpd = j - 1
        n = 1 << k
        ek = k
        while (1 << ek) < n * qpd:
            ek += 1
        ext_omega = pow(ROOT_OF_UNITY, 1 << (S - ek), R_MOD)
        omega = pow(ext_omega, 1 << (ek - k), R_MOD)
        t_len = 1 << (ek - k)
        zn = pow(ZETA, n, R_MOD)
        step = pow(ext_omega, n, R_MOD)
        t_evals = [pow((zn * pow(step, i, R_MOD) - 1) % R_MOD, -1, R_MOD) for i in range(t_len)]
        out.append(
            {
                "j": j,
                "k": k,
                "extended_k": ek,
                "quotient_poly_degree": qpd,
                "omega": hx(omega),
                "omega_inv": hx(pow(omega, -1, R_MOD)),
                "extended_omega": hx(ext_omega),
                "extended_omega_inv": hx(pow(ext_omega, -1, R_MOD)),
                "g_coset": hx(ZETA),
                "g_coset_inv": hx(ZETA * ZETA % R_MOD),
                "ifft_divisor": hx(pow(n, -1, R_MOD)),
                "extended_ifft_divisor": hx(pow(1 << ek, -1, R_MOD)),
                "barycentric_weight": hx(pow(n, -1, R_MOD)),
                "t_evaluations": [hx(v) for v in t_evals],
            }
        )
    return out


def gen_coset(rng):
    """coeff_to_extended / extended_to_coeff (poly/domain.rs:270-350): evaluations of
    a(X) on zeta*<extended_omega>, computed here by direct Horner evaluation."""
    out = []
    for j, k in ((3, 3), (4, 4), (5, 5)):
        n = 1 << k
        ek = k
        while (1 << ek) < n * (j - 1):
            ek += 1
        ext_omega = pow(ROOT_OF_UNITY, 1 << (S - ek), R_MOD)
        coeffs = [rng.randrange(R_MOD) for _ in range(n)]
        evals = []
        for i in range(1 << ek):
            x = ZETA * pow(ext_omega, i, R_MOD) % R_MOD
            acc = 0
            for c in reversed(coeffs):
                acc = (acc * x + c) % R_MOD
            evals.append(acc)
        out.append({"j": j, "k": k, "extended_k": ek, "coeffs": [hx(c) for c in coeffs], "extended": [hx(v) for v in evals]})
    return out


def gen_setup(rng):
    """Params::unsafe_setup with injected s (poly/commitment.rs:56-124), k = 3."""
    k = 3
    n = 1 << k
    s = rng.randrange(R_MOD)
    G = (1, 2)
    g = [ec_mul(pow(s, i, R_MOD), G) for i in range(n)]
    root = pow(ROOT_OF_UNITY, 1 << (S - k), R_MOD)
    mult = (pow(s, n, R_MOD) - 1) * pow(n, -1, R_MOD) % R_MOD
    gl = []
    for i in range(n):
        rp = pow(root, i, R_MOD)
        sc = mult * rp % R_MOD * pow((s - rp) % R_MOD, -1, R_MOD) % R_MOD
        gl.append(ec_mul(sc, G))
    return {"k": k, "s": hx(s), "g": [pt_json(p) for p in g], "g_lagrange": [pt_json(p) for p in gl]}


def main():
    rng = random.Random(0x48414C4F32)  # "HALO2"
    files = {
        "constants.json": gen_constants(),
        "field_kat.json": gen_field(rng),
        "ntt_kat.json": gen_ntt(rng),
        "msm_kat.json": gen_msm(rng),
        "g1_kat.json": gen_g1(rng),
        "domain_kat.json": gen_domain(),
        "coset_kat.json": gen_coset(rng),
        "setup_kat.json": gen_setup(rng),
    }
    for name, obj in files.items():
        with open(os.path.join(HERE, name), "w") as f:
            json.dump(obj, f, indent=0, separators=(",", ":"))
            f.write("\n")
        print("wrote", name)


if __name__ == "__main__":
    main()
