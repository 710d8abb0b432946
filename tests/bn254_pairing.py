"""TEST INFRASTRUCTURE -- the BN254 (alt_bn128) optimal ate pairing in plain Python integers, for the verifier-side
acceptance check of SURVEY.md 8(f) N3: the opening equation of plonk/verifier.rs / poly/multiopen.rs:29-55 is
e(L, [s]G2) = e(R, G2).  Fq12 is represented as Fq[w] / (w^12 - 18 w^6 + 82) (so u = w^6 - 9 with u^2 = -1 and
xi = 9 + u = w^6), G2 lives on the sextic twist y^2 = x^3 + 3 / xi over Fq2 and is mapped into E(Fq12) by
(x, y) -> (x w^2, y w^3).  Slow (about a second per pairing) and only used by tests.

Self-checks (tests/test_plonk_host.py): generator on the twist and of order r, bilinearity, non-degeneracy.
"""
Q = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
ATE_LOOP_COUNT = 29793968203157093288  # 6x + 2, x = 4965661367192848881
LOG_ATE_LOOP_COUNT = 63

# ---- Fq2 = Fq[u] / (u^2 + 1), elements (c0, c1) -----------------------------------------------------------


def f2_add(a, b):
    return ((a[0] + b[0]) % Q, (a[1] + b[1]) % Q)


def f2_sub(a, b):
    return ((a[0] - b[0]) % Q, (a[1] - b[1]) % Q)


def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q)


def f2_inv(a):
    d = pow(a[0] * a[0] + a[1] * a[1], -1, Q)
    return (a[0] * d % Q, (-a[1]) * d % Q)


B2 = f2_mul((3, 0), f2_inv((9, 1)))  # 3 / (9 + u)

G2 = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
       11559732032986387107991004021392285783925812861821192530917403151452391805634),
      (8495653923123431417604973247489272438418190587263600148770280649306958101930,
       4082367875863433681332203403145435568316851327593401208105741076214120093531))


def g2_on_curve(P):
    if P is None:
        return True
    x, y = P
    return f2_mul(y, y) == f2_add(f2_mul(f2_mul(x, x), x), B2)


def g2_add(P, T):
    if P is None:
        return T
    if T is None:
        return P
    (x1, y1), (x2, y2) = P, T
    if x1 == x2:
        if f2_add(y1, y2) == (0, 0):
            return None
        lam = f2_mul(f2_mul((3, 0), f2_mul(x1, x1)), f2_inv(f2_add(y1, y1)))
    else:
        lam = f2_mul(f2_sub(y2, y1), f2_inv(f2_sub(x2, x1)))
    x3 = f2_sub(f2_sub(f2_mul(lam, lam), x1), x2)
    return (x3, f2_sub(f2_mul(lam, f2_sub(x1, x3)), y1))


def g2_mul(P, e):
    acc = None
    while e:
        if e & 1:
            acc = g2_add(acc, P)
        P = g2_add(P, P)
        e >>= 1
    return acc


# ---- Fq12 = Fq[w] / (w^12 - 18 w^6 + 82), elements = lists of 12 ints ---------------------------------------
ONE12 = [1] + [0] * 11


def f12_mul(a, b):
    t = [0] * 23
    for i, ai in enumerate(a):
        if ai:
            for j, bj in enumerate(b):
                t[i + j] += ai * bj
    for i in range(22, 11, -1):
        top = t[i]
        if top:
            t[i - 12] -= 82 * top
            t[i - 6] += 18 * top
    return [c % Q for c in t[:12]]


def f12_add(a, b):
    return [(x + y) % Q for x, y in zip(a, b)]


def f12_sub(a, b):
    return [(x - y) % Q for x, y in zip(a, b)]


def f12_scalar(a, c):
    return [x * c % Q for x in a]


def f12_pow(a, e):
    out = ONE12
    while e:
        if e & 1:
            out = f12_mul(out, a)
        a = f12_mul(a, a)
        e >>= 1
    return out


def _deg(p):
    d = len(p) - 1
    while d and p[d] == 0:
        d -= 1
    return d


def _poly_div(a, b):
    """rounded-down quotient of polynomials over Fq"""
    dega, degb = _deg(a), _deg(b)
    temp = list(a)
    out = [0] * len(a)
    binv = pow(b[degb], -1, Q)
    for i in range(dega - degb, -1, -1):
        out[i] = (out[i] + temp[degb + i] * binv) % Q
        for c in range(degb + 1):
            temp[c + i] = (temp[c + i] - out[i] * b[c]) % Q
    return out[: _deg(out) + 1]


def f12_inv(a):
    """extended Euclid against the modulus polynomial"""
    lm, hm = [1] + [0] * 12, [0] * 13
    low, high = list(a) + [0], [82, 0, 0, 0, 0, 0, Q - 18, 0, 0, 0, 0, 0, 1]
    while _deg(low):
        r = _poly_div(high, low)
        r += [0] * (13 - len(r))
        nm, new = list(hm), list(high)
        for i in range(13):
            for j in range(13 - i):
                nm[i + j] -= lm[i] * r[j]
                new[i + j] -= low[i] * r[j]
        nm = [x % Q for x in nm]
        new = [x % Q for x in new]
        lm, low, hm, high = nm, new, lm, low
    inv0 = pow(low[0], -1, Q)
    return [c * inv0 % Q for c in lm[:12]]


def f12_from_fq(c):
    return [c % Q] + [0] * 11


def twist(P):
    """G2 point on the twist -> point of E(Fq12): y^2 = x^3 + 3"""
    if P is None:
        return None
    (x0, x1), (y0, y1) = P
    nx = [(x0 - 9 * x1) % Q, 0, 0, 0, 0, 0, x1, 0, 0, 0, 0, 0]
    ny = [(y0 - 9 * y1) % Q, 0, 0, 0, 0, 0, y1, 0, 0, 0, 0, 0]
    w2 = [0, 0, 1] + [0] * 9
    w3 = [0, 0, 0, 1] + [0] * 8
    return (f12_mul(nx, w2), f12_mul(ny, w3))


def _e12_double(P):
    x, y = P
    lam = f12_mul(f12_scalar(f12_mul(x, x), 3), f12_inv(f12_scalar(y, 2)))
    nx = f12_sub(f12_mul(lam, lam), f12_scalar(x, 2))
    return (nx, f12_sub(f12_mul(lam, f12_sub(x, nx)), y))


def _e12_add(P, T):
    if P is None or T is None:
        return P if T is None else T
    (x1, y1), (x2, y2) = P, T
    if x1 == x2:
        return _e12_double(P) if y1 == y2 else None
    lam = f12_mul(f12_sub(y2, y1), f12_inv(f12_sub(x2, x1)))
    nx = f12_sub(f12_sub(f12_mul(lam, lam), x1), x2)
    return (nx, f12_sub(f12_mul(lam, f12_sub(x1, nx)), y1))


def _linefunc(P1, P2, T):
    (x1, y1), (x2, y2), (xt, yt) = P1, P2, T
    if x1 != x2:
        m = f12_mul(f12_sub(y2, y1), f12_inv(f12_sub(x2, x1)))
    elif y1 == y2:
        m = f12_mul(f12_scalar(f12_mul(x1, x1), 3), f12_inv(f12_scalar(y1, 2)))
    else:
        return f12_sub(xt, x1)
    return f12_sub(f12_mul(m, f12_sub(xt, x1)), f12_sub(yt, y1))


def miller_loop(Qt, P):
    """Qt: G2 point (on the twist), P: G1 affine point (x, y) of integers; no final exponentiation"""
    if Qt is None or P is None:
        return ONE12
    Q12 = twist(Qt)
    P12 = (f12_from_fq(P[0]), f12_from_fq(P[1]))
    Rp, f = Q12, ONE12
    for i in range(LOG_ATE_LOOP_COUNT, -1, -1):
        f = f12_mul(f12_mul(f, f), _linefunc(Rp, Rp, P12))
        Rp = _e12_double(Rp)
        if ATE_LOOP_COUNT & (1 << i):
            f = f12_mul(f, _linefunc(Rp, Q12, P12))
            Rp = _e12_add(Rp, Q12)
    Q1 = (f12_pow(Q12[0], Q), f12_pow(Q12[1], Q))
    nQ2 = (f12_pow(Q1[0], Q), [(-c) % Q for c in f12_pow(Q1[1], Q)])
    f = f12_mul(f, _linefunc(Rp, Q1, P12))
    Rp = _e12_add(Rp, Q1)
    f = f12_mul(f, _linefunc(Rp, nQ2, P12))
    return f


def final_exponentiation(f):
    return f12_pow(f, (Q ** 12 - 1) // R)


def pairing(Qt, P):
    return final_exponentiation(miller_loop(Qt, P))


def pairing_check(pairs):
    """prod e(P_i, Q_i) == 1 for [(G1 point, G2 point), ...] with one shared final exponentiation"""
    f = ONE12
    for P, Qt in pairs:
        f = f12_mul(f, miller_loop(Qt, P))
    return final_exponentiation(f) == ONE12
