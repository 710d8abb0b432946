"""TEST INFRASTRUCTURE -- a pure-Python (big-integer) restatement of the reference's proof flow for circuits made
of custom gates and the permutation argument, written independently of the product prover
(halo2-gpu-specific_amd/prover.py) so that the two can be compared byte for byte.  PARITY UNPINNED against the
Rust binary (no Rust toolchain here): the Debug-string verifying-key digest (plonk.rs:91-109) and the compressed
point flag bits (pairing_bn256@30b052f `to_bytes`) are stated conventions, see DESIGN.md.

Follows:  plonk/prover.rs:206-850 (create_proof_ext), plonk/permutation/prover.rs:47-165 (commit), :200-330
(evaluate/open), plonk/permutation/keygen.rs:112-143 (cycle -> mapping), :197-261 (sigma polynomials),
plonk/vanishing/prover.rs:40-160, poly/multiopen/shplonk.rs:58-135 (intermediate sets),
poly/multiopen/shplonk/prover.rs:89-225, poly/multiopen/shplonk/verifier.rs:23-103, plonk/verifier.rs:128-507,
plonk/permutation/verifier.rs:105-200, plonk/vanishing/verifier.rs:86-120, transcript.rs:15-300,
poly/domain.rs:44-149, :270-350, plonk/keygen.rs:395-425 (l0 / l_last / l_active_row).

Commitments use the trapdoor of `Params::unsafe_setup` (poly/commitment.rs:56-124): g[i] = [s^i]G and
g_lagrange[i] = [L_i(s)]G, so commit(p) = [p(s)]G -- one field evaluation and one scalar multiplication, with no
MSM and no SRS table.  That makes this file an independent check of the product's MSM + SRS plumbing too.
"""
import hashlib

R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
Q = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
ROOT_OF_UNITY = 0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C
DELTA = 0x09226B6E22C6F0CA64EC26AAD4C86E715B5F898E5E963F25870E56BBE533E9A2
ZETA = 0x30644E72E131A029048B6E193FD84104CC37A73FEC2BC5E9B8CA0B2D36636F23
S = 28
G1 = (1, 2)


def inv(a, p=R):
    return pow(a, -1, p)


# ---- BN254 G1, affine, identity = None ------------------------------------------------------------------
def g1_add(P, T):
    if P is None:
        return T
    if T is None:
        return P
    x1, y1 = P
    x2, y2 = T
    if x1 == x2:
        if (y1 + y2) % Q == 0:
            return None
        lam = 3 * x1 * x1 * inv(2 * y1, Q) % Q
    else:
        lam = (y2 - y1) * inv(x2 - x1, Q) % Q
    x3 = (lam * lam - x1 - x2) % Q
    return (x3, (lam * (x1 - x3) - y1) % Q)


def g1_neg(P):
    return None if P is None else (P[0], (-P[1]) % Q)


def g1_mul(P, e):
    e %= R
    acc = None
    while e:
        if e & 1:
            acc = g1_add(acc, P)
        P = g1_add(P, P)
        e >>= 1
    return acc


def point_to_bytes(P):
    """compressed point: x little-endian, bit 7 of the last byte = parity of y; identity = zeros"""
    if P is None:
        return bytes(32)
    b = bytearray(P[0].to_bytes(32, "little"))
    b[31] |= (P[1] & 1) << 7
    return bytes(b)


def point_from_bytes(b):
    if b == bytes(32):
        return None
    sign = b[31] >> 7
    x = int.from_bytes(bytes(b[:31]) + bytes([b[31] & 0x7F]), "little")
    assert x < Q
    y = pow((x * x * x + 3) % Q, (Q + 1) // 4, Q)
    assert y * y % Q == (x * x * x + 3) % Q, "not on the curve"
    if (y & 1) != sign:
        y = Q - y
    return (x, y)


# ---- transcript (transcript.rs) -----------------------------------------------------------------------
class Transcript:
    def __init__(self, proof=None):
        self.state = hashlib.blake2b(digest_size=64, person=b"Halo2-Transcript")
        self.out = bytearray()
        self.inp = proof
        self.pos = 0

    def squeeze(self):
        self.state.update(b"\x00")
        return int.from_bytes(self.state.copy().digest(), "little") % R

    def common_point(self, P):
        assert P is not None, "cannot write points at infinity to the transcript"
        self.state.update(b"\x01")
        self.state.update(P[0].to_bytes(32, "little"))
        self.state.update(P[1].to_bytes(32, "little"))

    def common_scalar(self, v):
        self.state.update(b"\x02")
        self.state.update(v.to_bytes(32, "little"))

    def write_point(self, P):
        self.common_point(P)
        self.out += point_to_bytes(P)

    def write_scalar(self, v):
        self.common_scalar(v)
        self.out += v.to_bytes(32, "little")

    def read_point(self):
        P = point_from_bytes(self.inp[self.pos:self.pos + 32])
        self.pos += 32
        self.common_point(P)
        return P

    def read_scalar(self):
        v = int.from_bytes(self.inp[self.pos:self.pos + 32], "little")
        assert v < R
        self.pos += 32
        self.common_scalar(v)
        return v


# ---- polynomials over Fr as lists of ints ----------------------------------------------------------------
def fft(a, omega):
    """natural order in / out, a[i] <- sum_j a[j] omega^(ij)"""
    n = len(a)
    if n == 1:
        return a[:]
    even = fft(a[0::2], omega * omega % R)
    odd = fft(a[1::2], omega * omega % R)
    out = [0] * n
    w = 1
    for i in range(n // 2):
        t = w * odd[i] % R
        out[i] = (even[i] + t) % R
        out[i + n // 2] = (even[i] - t) % R
        w = w * omega % R
    return out


def eval_poly(p, x):
    acc = 0
    for c in reversed(p):
        acc = (acc * x + c) % R
    return acc


def kate_division(a, b):
    """q(X) = a(X) / (X - b), remainder dropped (arithmetic.rs:754-773)"""
    q = [0] * (len(a) - 1)
    tmp = 0
    for i in range(len(a) - 1, 0, -1):
        tmp = (a[i] + tmp * b) % R
        q[i - 1] = tmp
    return q


def lagrange_interpolate(points, evals):
    """coefficients of the degree < len(points) interpolant (arithmetic.rs:849-903)"""
    n = len(points)
    out = [0] * n
    for j in range(n):
        num = [1]
        den = 1
        for m in range(n):
            if m == j:
                continue
            num = [((num[i - 1] if i > 0 else 0) - points[m] * (num[i] if i < len(num) else 0)) % R for i in range(len(num) + 1)]
            den = den * (points[j] - points[m]) % R
        c = evals[j] * inv(den) % R
        for i in range(n):
            out[i] = (out[i] + c * num[i]) % R
    return out


def vanishing_eval(roots, z):
    acc = 1
    for r in roots:
        acc = acc * (z - r) % R
    return acc


class Domain:
    """EvaluationDomain::new (poly/domain.rs:44-149)"""

    def __init__(self, k, degree):
        self.k, self.n = k, 1 << k
        self.quotient_poly_degree = degree - 1
        ek = k
        while (1 << ek) < self.n * self.quotient_poly_degree:
            ek += 1
        self.extended_k, self.extended_n = ek, 1 << ek
        self.extended_omega = pow(ROOT_OF_UNITY, 1 << (S - ek), R)
        self.omega = pow(self.extended_omega, 1 << (ek - k), R)
        self.omega_inv = inv(self.omega)
        self.extended_omega_inv = inv(self.extended_omega)
        self.n_inv = inv(self.n)
        self.extended_n_inv = inv(self.extended_n)
        self.g_coset, self.g_coset_inv = ZETA, ZETA * ZETA % R

    def rotate_omega(self, x, rot):
        return x * pow(self.omega if rot >= 0 else self.omega_inv, abs(rot), R) % R

    def lagrange_to_coeff(self, v):
        return [c * self.n_inv % R for c in fft(v, self.omega_inv)]

    def coeff_to_extended(self, c):
        """values at ZETA * extended_omega^j (poly/domain.rs:270-287, :382-398)"""
        z = [1, self.g_coset, self.g_coset_inv]
        a = [c[i] * z[i % 3] % R for i in range(len(c))] + [0] * (self.extended_n - len(c))
        return fft(a, self.extended_omega)

    def extended_to_coeff(self, v):
        a = fft(v, self.extended_omega_inv)
        z = [1, self.g_coset_inv, self.g_coset]
        a = [a[i] * self.extended_n_inv % R * z[i % 3] % R for i in range(len(a))]
        return a[: self.n * self.quotient_poly_degree]

    def l_i_range(self, x, xn, rots):
        """evaluations of the Lagrange basis polynomials l_rot(x) (poly/domain.rs l_i_range)"""
        out = []
        for rot in rots:
            w = self.rotate_omega(1, rot)
            out.append((xn - 1) * w % R * self.n_inv % R * inv((x - w) % R) % R)
        return out


# ---- circuit description -------------------------------------------------------------------------------
class MiniPlonk:
    """the commented-out circuit of examples/simple-example-2.rs:177-288: a*sa + b*sb + a*b*sm - c*sc"""
    num_advice, num_fixed = 3, 4                 # advice a, b, c; fixed sm, sa, sb, sc (allocation order :193-196)
    advice_queries = [(0, 0), (1, 0), (2, 0)]
    fixed_queries = [(1, 0), (2, 0), (3, 0), (0, 0)]   # query order inside the gate: sa, sb, sc, sm
    perm_columns = [("advice", 0), ("advice", 1), ("advice", 2)]
    degree = 3
    blinding_factors = 5                          # circuit.rs:1919-1944 with one query per advice column
    name = "mini-plonk"

    @staticmethod
    def gates(adv, fix):
        a, b, c = adv(0, 0), adv(1, 0), adv(2, 0)
        sm, sa, sb, sc = fix(0, 0), fix(1, 0), fix(2, 0), fix(3, 0)
        return [(a * sa + b * sb + a * b * sm - c * sc) % R]

    @staticmethod
    def synthesize(k, a=5):
        """rows 2i (mul) and 2i+1 (add) for i < 2^(k-4); copies a0 = a1 and b1 = c0 (:229-251)"""
        n = 1 << k
        adv = [[0] * n for _ in range(3)]
        fixed = [[0] * n for _ in range(4)]
        copies = []
        a2 = a * a % R
        for i in range(1 << (k - 4)):
            r0, r1 = 2 * i, 2 * i + 1
            adv[0][r0], adv[1][r0], adv[2][r0] = a, a, a2
            fixed[0][r0], fixed[3][r0] = 1, 1                      # sm = 1, sc = 1
            adv[0][r1], adv[1][r1], adv[2][r1] = a, a2, (a + a2) % R
            fixed[1][r1], fixed[2][r1], fixed[3][r1] = 1, 1, 1     # sa = sb = sc = 1
            copies.append(((0, r0), (0, r1)))
            copies.append(((1, r1), (2, r0)))
        return adv, fixed, copies


class RotGate(MiniPlonk):
    """a second shape that exercises rotations and a 2-column-per-set permutation (degree 4):
    gate 0: s0 * (a(X) + b(X) - c(X)); gate 1: s1 * (a(wX) - c(X)) * (b(w^-1 X) + s0)"""
    num_advice, num_fixed = 3, 2
    advice_queries = [(0, 0), (1, 0), (2, 0), (0, 1), (1, -1)]
    fixed_queries = [(0, 0), (1, 0)]
    perm_columns = [("advice", 0), ("advice", 1), ("advice", 2), ("fixed", 1)]
    degree = 4
    blinding_factors = 5
    name = "rot-gate"

    @staticmethod
    def gates(adv, fix):
        s0, s1 = fix(0, 0), fix(1, 0)
        return [s0 * (adv(0, 0) + adv(1, 0) - adv(2, 0)) % R,
                s1 * (adv(0, 1) - adv(2, 0)) % R * (adv(1, -1) + s0) % R]

    @staticmethod
    def synthesize(k, a=7):
        n = 1 << k
        usable = n - 6
        adv = [[0] * n for _ in range(3)]
        fixed = [[0] * n for _ in range(2)]
        copies = []
        x, y = a, a + 1
        for r in range(usable - 1):
            adv[0][r], adv[1][r], adv[2][r] = x, y, (x + y) % R
            fixed[0][r] = 1
            if r > 0:
                fixed[1][r - 1] = 1 if r % 3 else 0           # a(next row) = c(this row) on some rows
                copies.append(((0, r), (2, r - 1)))
            x, y = (x + y) % R, (y * 3 + 1) % R
        # tie a fixed-column cell into a cycle with an advice cell holding the same value (1)
        adv[1][usable - 1] = 1
        copies.append(((3, 0), (1, usable - 1)))
        return adv, fixed, copies


def permutation_mapping(ncols, n, copies):
    """cycles -> mapping: each cycle sorted by (column, row), every cell maps to its successor
    (permutation/keygen.rs:112-143); the merge order does not matter"""
    parent = {}

    def find(x):
        while parent.setdefault(x, x) != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    for l, r in copies:
        parent[find(l)] = find(r)
    classes = {}
    for x in list(parent):
        classes.setdefault(find(x), []).append(x)
    mapping = [[(c, j) for j in range(n)] for c in range(ncols)]
    for cyc in classes.values():
        cyc.sort()
        for i, cell in enumerate(cyc):
            mapping[cell[0]][cell[1]] = cyc[(i + 1) % len(cyc)]
    return mapping


def vk_digest(cs, k, fixed_commitments, perm_commitments):
    """stand-in for `format!("{:?}", vk.pinned())` (plonk.rs:91-109): same framing (u64 length, Blake2b-512 with
    the Halo2-Verify-Key personalisation, from_bytes_wide), our own canonical text"""
    s = "halo2-hip-vk circuit=%s k=%d advice=%d fixed=%d degree=%d fixed_commitments=%s permutation_commitments=%s" % (
        cs.name, k, cs.num_advice, cs.num_fixed, cs.degree,
        ",".join(point_to_bytes(p).hex() for p in fixed_commitments),
        ",".join(point_to_bytes(p).hex() for p in perm_commitments))
    h = hashlib.blake2b(digest_size=64, person=b"Halo2-Verify-Key")
    h.update(len(s).to_bytes(8, "little"))
    h.update(s.encode())
    return int.from_bytes(h.digest(), "little") % R


class Keys:
    pass


def keygen(cs, k, s, fixed, copies):
    """keygen_vk + keygen_pk (plonk/keygen.rs) with trapdoor commitments"""
    dom = Domain(k, cs.degree)
    n = dom.n
    pk = Keys()
    pk.cs, pk.dom, pk.s = cs, dom, s
    pk.fixed_values = fixed
    pk.fixed_polys = [dom.lagrange_to_coeff(f) for f in fixed]
    mapping = permutation_mapping(len(cs.perm_columns), n, copies)
    omegas = [pow(dom.omega, j, R) for j in range(n)]
    pk.sigma_values = [[pow(DELTA, mapping[i][j][0], R) * omegas[mapping[i][j][1]] % R for j in range(n)]
                       for i in range(len(cs.perm_columns))]
    pk.sigma_polys = [dom.lagrange_to_coeff(v) for v in pk.sigma_values]
    bf = cs.blinding_factors
    l0 = [0] * n
    l0[0] = 1
    l_last = [0] * n
    l_last[n - bf - 1] = 1
    l_blind = [0] * n
    for i in range(n - bf, n):
        l_blind[i] = 1
    pk.l0 = dom.coeff_to_extended(dom.lagrange_to_coeff(l0))
    pk.l_last = dom.coeff_to_extended(dom.lagrange_to_coeff(l_last))
    lb = dom.coeff_to_extended(dom.lagrange_to_coeff(l_blind))
    pk.l_active_row = [(1 - pk.l_last[i] - lb[i]) % R for i in range(dom.extended_n)]
    pk.fixed_commitments = [g1_mul(G1, eval_poly(p, s)) for p in pk.fixed_polys]
    pk.perm_commitments = [g1_mul(G1, eval_poly(p, s)) for p in pk.sigma_polys]
    pk.transcript_repr = vk_digest(cs, k, pk.fixed_commitments, pk.perm_commitments)
    return pk


def commit(pk, coeffs):
    return g1_mul(G1, eval_poly(coeffs, pk.s))


def column_values(kind, idx, advice, fixed):
    return {"advice": advice, "fixed": fixed}[kind][idx]


def intermediate_sets(queries):
    """queries: list of (commitment key, rotation, point, eval).  poly/multiopen/shplonk.rs:58-135."""
    rot_point = {}
    for _, rot, point, _ in queries:
        assert rot_point.setdefault(rot, point) == point
    super_point_set = [rot_point[r] for r in sorted(rot_point)]
    order, rotsets = [], {}
    for key, rot, _, _ in queries:
        if key not in rotsets:
            rotsets[key] = set()
            order.append(key)
        rotsets[key].add(rot)
    groups = {}
    for key in order:
        groups.setdefault(tuple(sorted(rotsets[key])), []).append(key)
    evals = {(key, rot): ev for key, rot, _, ev in queries}
    out = []
    for rots in sorted(groups):
        out.append({"points": [rot_point[r] for r in rots],
                    "commitments": [(key, [evals[(key, r)] for r in rots]) for key in groups[rots]]})
    return out, super_point_set


def poly_sub_low(p, low):
    out = p[:]
    for i, c in enumerate(low):
        out[i] = (out[i] - c) % R
    return out


def fold(polys, ch, n):
    acc = [0] * n
    for p in polys:
        acc = [(acc[i] * ch + (p[i] if i < len(p) else 0)) % R for i in range(n)]
    return acc


def create_proof(pk, advice_in, rng, use_gwc=False):
    """plonk/prover.rs:206-850 (create_proof_ext); one circuit instance, no instance columns, no lookups/shuffles"""
    cs, dom = pk.cs, pk.dom
    n, bf = dom.n, cs.blinding_factors
    t = Transcript()
    t.common_scalar(pk.transcript_repr)
    # advice: blinding rows (prover.rs:281-289), commitments
    advice = [col[:] for col in advice_in]
    for col in advice:
        for r in range(n - (bf + 1), n):
            col[r] = rng.u16()
    advice_polys = [dom.lagrange_to_coeff(col) for col in advice]
    for p in advice_polys:
        t.write_point(commit(pk, p))
    theta = t.squeeze()  # noqa: F841 - drawn even without lookups (prover.rs:318)
    beta = t.squeeze()
    gamma = t.squeeze()
    # permutation grand products (permutation/prover.rs:47-165)
    chunk = cs.degree - 2
    cols = cs.perm_columns
    zs, last_z = [], 1
    for si in range(0, len(cols), chunk):
        mv = [1] * n
        for ci in range(si, min(si + chunk, len(cols))):
            vals = column_values(*cols[ci], advice, pk.fixed_values)
            for i in range(n):
                mv[i] = mv[i] * (beta * pk.sigma_values[ci][i] + gamma + vals[i]) % R
        mv = [inv(v) for v in mv]
        dw = pow(DELTA, si, R)
        for ci in range(si, min(si + chunk, len(cols))):
            vals = column_values(*cols[ci], advice, pk.fixed_values)
            for i in range(n):
                mv[i] = mv[i] * (dw * beta + gamma + vals[i]) % R
                dw = dw * dom.omega % R
            dw = dw * DELTA % R
        z = [last_z]
        for i in range(n - 1):
            z.append(z[i] * mv[i] % R)
        for i in range(n - bf, n):
            z[i] = rng.fr()
        last_z = z[n - (bf + 1)]
        zs.append(z)
    z_polys = [dom.lagrange_to_coeff(z) for z in zs]
    for p in z_polys:
        t.write_point(commit(pk, p))
    # vanishing argument: random polynomial (vanishing/prover.rs:40-67)
    random_poly = rng.random_poly(n)
    t.write_point(commit(pk, random_poly))
    y = t.squeeze()
    # h(X) on the extended coset
    adv_c = [dom.coeff_to_extended(p) for p in advice_polys]
    fix_c = [dom.coeff_to_extended(p) for p in pk.fixed_polys]
    sig_c = [dom.coeff_to_extended(p) for p in pk.sigma_polys]
    z_c = [dom.coeff_to_extended(p) for p in z_polys]
    en = dom.extended_n
    scale = en // n
    last_rot = -(bf + 1)
    h = [0] * en
    point = ZETA
    for j in range(en):
        adv = lambda c, r: adv_c[c][(j + r * scale) % en]  # noqa: E731
        fix = lambda c, r: fix_c[c][(j + r * scale) % en]  # noqa: E731
        exprs = list(cs.gates(adv, fix))
        l0, ll, la = pk.l0[j], pk.l_last[j], pk.l_active_row[j]
        exprs.append(l0 * (1 - z_c[0][j]) % R)
        exprs.append(ll * (z_c[-1][j] * z_c[-1][j] - z_c[-1][j]) % R)
        for i in range(1, len(z_c)):
            exprs.append(l0 * (z_c[i][j] - z_c[i - 1][(j + last_rot * scale) % en]) % R)
        for i in range(len(z_c)):
            left = z_c[i][(j + scale) % en]
            right = z_c[i][j]
            cur = beta * point % R * pow(DELTA, i * chunk, R) % R
            for ci in range(i * chunk, min((i + 1) * chunk, len(cols))):
                kind, idx = cols[ci]
                v = (adv_c if kind == "advice" else fix_c)[idx][j]
                left = left * (v + beta * sig_c[ci][j] + gamma) % R
                right = right * (v + cur + gamma) % R
                cur = cur * DELTA % R
            exprs.append(la * (left - right) % R)
        acc = 0
        for e in exprs:
            acc = (acc * y + e) % R
        h[j] = acc * inv((pow(point, n, R) - 1) % R) % R
        point = point * dom.extended_omega % R
    h_coeffs = dom.extended_to_coeff(h)
    pieces = [h_coeffs[i * n:(i + 1) * n] for i in range(dom.quotient_poly_degree)]
    for p in pieces:
        t.write_point(commit(pk, p))
    x = t.squeeze()
    xn = pow(x, n, R)
    # evaluations (prover.rs:700-790)
    for c, rot in cs.advice_queries:
        t.write_scalar(eval_poly(advice_polys[c], dom.rotate_omega(x, rot)))
    for c, rot in cs.fixed_queries:
        t.write_scalar(eval_poly(pk.fixed_polys[c], dom.rotate_omega(x, rot)))
    h_poly = fold(reversed(pieces), xn, n)
    t.write_scalar(eval_poly(random_poly, x))
    for p in pk.sigma_polys:
        t.write_scalar(eval_poly(p, x))
    x_next, x_last = dom.rotate_omega(x, 1), dom.rotate_omega(x, last_rot)
    for i, p in enumerate(z_polys):
        t.write_scalar(eval_poly(p, x))
        t.write_scalar(eval_poly(p, x_next))
        if i + 1 < len(z_polys):
            t.write_scalar(eval_poly(p, x_last))
    # multiopen query list (prover.rs:792-840)
    polys = {}
    queries = []

    def q(key, poly, rot):
        polys[key] = poly
        pt = dom.rotate_omega(x, rot)
        queries.append((key, rot, pt, eval_poly(poly, pt)))

    for c, rot in cs.advice_queries:
        q(("advice", c), advice_polys[c], rot)
    for i, p in enumerate(z_polys):
        q(("z", i), p, 0)
        q(("z", i), p, 1)
    for i in reversed(range(len(z_polys) - 1)):
        q(("z", i), z_polys[i], last_rot)
    for c, rot in cs.fixed_queries:
        q(("fixed", c), pk.fixed_polys[c], rot)
    for i, p in enumerate(pk.sigma_polys):
        q(("sigma", i), p, 0)
    q(("h",), h_poly, 0)
    q(("random",), random_poly, 0)
    (gwc_prove if use_gwc else shplonk_prove)(pk, t, queries, polys, n)
    return bytes(t.out)


def gwc_sets(queries):
    """poly/multiopen/gwc.rs:36-60: queries grouped by rotation, groups in rotation order"""
    groups = {}
    for qu in queries:
        groups.setdefault(qu[1], []).append(qu)
    return [groups[r] for r in sorted(groups)]


def gwc_prove(pk, t, queries, polys, n):
    """poly/multiopen/gwc/prover.rs:20-175"""
    v = t.squeeze()
    for group in gwc_sets(queries):
        z = group[0][2]
        batch = fold([polys[key] for key, _, _, _ in group], v, n)
        batch[0] = (batch[0] - eval_poly(batch, z)) % R
        t.write_point(commit(pk, kate_division(batch, z)))


def shplonk_prove(pk, t, queries, polys, n):
    """poly/multiopen/shplonk/prover.rs:89-225"""
    y = t.squeeze()
    rsets, super_points = intermediate_sets(queries)
    for rs in rsets:
        rs["low"] = [lagrange_interpolate(rs["points"], evals) for _, evals in rs["commitments"]]
    v = t.squeeze()
    quotients = []
    for rs in rsets:
        nums = [poly_sub_low(polys[key], low) for (key, _), low in zip(rs["commitments"], rs["low"])]
        n_x = fold(nums, y, n)
        for pt in rs["points"]:
            n_x = kate_division(n_x, pt)
        quotients.append(n_x + [0] * (n - len(n_x)))
    h_x = fold(quotients, v, n)
    t.write_point(commit(pk, h_x))
    u = t.squeeze()
    zt_eval = vanishing_eval(super_points, u)
    lin, z_diffs = [], []
    for rs in rsets:
        diffs = [p for p in super_points if p not in rs["points"]]
        z_i = vanishing_eval(diffs, u)
        inner = [poly_sub_low(polys[key], [eval_poly(low, u)]) for (key, _), low in zip(rs["commitments"], rs["low"])]
        l_x = fold(inner, y, n)
        lin.append([c * z_i % R for c in l_x])
        z_diffs.append(z_i)
    l_x = fold(lin, v, n)
    l_x = [(l_x[i] - h_x[i] * zt_eval) % R for i in range(n)]
    assert eval_poly(l_x, u) == 0
    hq = kate_division(l_x, u)
    zi = inv(z_diffs[0])
    hq = [c * zi % R for c in hq]
    t.write_point(commit(pk, hq))


# ---- verifier (plonk/verifier.rs) with the trapdoor standing in for the pairing ---------------------------
def opening_check(pk, left, right, pairing):
    """the decision of the `PairMSM` (poly/multiopen.rs:29-55, plonk/verifier.rs:496-507):
    e(left, [s]G2) * e(-right, G2) == 1.  pairing=False: the same statement through the setup trapdoor,
    [s]left == right; pairing=True: the real BN254 pairing on ParamsVerifier's s_g2 (bn254_pairing.py)."""
    if not pairing:
        return g1_mul(left, pk.s) == right
    import bn254_pairing as bp

    if getattr(pk, "s_g2", None) is None:
        pk.s_g2 = bp.g2_mul(bp.G2, pk.s)       # Params::unsafe_setup's additional_data (poly/commitment.rs:113-116)
    return bp.pairing_check([(left, pk.s_g2), (g1_neg(right), bp.G2)])


def verify_proof(pk, proof, use_gwc=False, pairing=False):
    """True iff the proof is accepted.  e(L, [s]G2) == e(Rgt, G2) is checked as [s]L == Rgt (s is known in the
    unsafe test setup), everything else follows plonk/verifier.rs:128-507."""
    cs, dom = pk.cs, pk.dom
    n, bf = dom.n, cs.blinding_factors
    t = Transcript(proof)
    t.common_scalar(pk.transcript_repr)
    advice_commitments = [t.read_point() for _ in range(cs.num_advice)]
    t.squeeze()  # theta
    beta = t.squeeze()
    gamma = t.squeeze()
    chunk = cs.degree - 2
    nsets = (len(cs.perm_columns) + chunk - 1) // chunk
    z_commitments = [t.read_point() for _ in range(nsets)]
    random_commitment = t.read_point()
    y = t.squeeze()
    h_commitments = [t.read_point() for _ in range(dom.quotient_poly_degree)]
    x = t.squeeze()
    advice_evals = [t.read_scalar() for _ in cs.advice_queries]
    fixed_evals = [t.read_scalar() for _ in cs.fixed_queries]
    random_eval = t.read_scalar()
    sigma_evals = [t.read_scalar() for _ in cs.perm_columns]
    z_evals = []
    for i in range(nsets):
        e = {"cur": t.read_scalar(), "next": t.read_scalar()}
        if i + 1 < nsets:
            e["last"] = t.read_scalar()
        z_evals.append(e)
    xn = pow(x, n, R)
    last_rot = -(bf + 1)
    l_evals = dom.l_i_range(x, xn, range(last_rot, 1))
    l_last, l_blind, l_0 = l_evals[0], sum(l_evals[1:1 + bf]) % R, l_evals[1 + bf]
    adv = lambda c, r: advice_evals[cs.advice_queries.index((c, r))]  # noqa: E731
    fix = lambda c, r: fixed_evals[cs.fixed_queries.index((c, r))]  # noqa: E731
    exprs = list(cs.gates(adv, fix))
    exprs.append(l_0 * (1 - z_evals[0]["cur"]) % R)
    exprs.append((z_evals[-1]["cur"] ** 2 - z_evals[-1]["cur"]) * l_last % R)
    for i in range(1, nsets):
        exprs.append((z_evals[i]["cur"] - z_evals[i - 1]["last"]) * l_0 % R)
    for i in range(nsets):
        left, right = z_evals[i]["next"], z_evals[i]["cur"]
        cur = beta * x % R * pow(DELTA, i * chunk, R) % R
        for ci in range(i * chunk, min((i + 1) * chunk, len(cs.perm_columns))):
            kind, idx = cs.perm_columns[ci]
            ev = adv(idx, 0) if kind == "advice" else fix(idx, 0)
            left = left * (ev + beta * sigma_evals[ci] + gamma) % R
            right = right * (ev + cur + gamma) % R
            cur = cur * DELTA % R
        exprs.append((left - right) * (1 - (l_last + l_blind)) % R)
    expected_h = 0
    for e in exprs:
        expected_h = (expected_h * y + e) % R
    expected_h = expected_h * inv((xn - 1) % R) % R
    h_commitment = None
    for c in reversed(h_commitments):
        h_commitment = g1_add(g1_mul(h_commitment, xn), c)
    # queries in the verifier's order (verifier.rs:385-470)
    commitments, queries = {}, []

    def q(key, com, rot, ev):
        commitments[key] = com
        queries.append((key, rot, dom.rotate_omega(x, rot), ev))

    for (c, rot), ev in zip(cs.advice_queries, advice_evals):
        q(("advice", c), advice_commitments[c], rot, ev)
    for i in range(nsets):
        q(("z", i), z_commitments[i], 0, z_evals[i]["cur"])
        q(("z", i), z_commitments[i], 1, z_evals[i]["next"])
    for i in reversed(range(nsets - 1)):
        q(("z", i), z_commitments[i], last_rot, z_evals[i]["last"])
    for (c, rot), ev in zip(cs.fixed_queries, fixed_evals):
        q(("fixed", c), pk.fixed_commitments[c], rot, ev)
    for i, ev in enumerate(sigma_evals):
        q(("sigma", i), pk.perm_commitments[i], 0, ev)
    q(("h",), h_commitment, 0, expected_h)
    q(("random",), random_commitment, 0, random_eval)
    if use_gwc:
        return gwc_verify(pk, t, proof, queries, commitments, pairing)
    # shplonk/verifier.rs:23-103
    rsets, super_points = intermediate_sets(queries)
    sy = t.squeeze()
    sv = t.squeeze()
    h1 = t.read_point()
    u = t.squeeze()
    h2 = t.read_point()
    assert t.pos == len(proof), "trailing bytes in the proof"
    outer, r_outer = None, 0
    z_0 = z_0_diff_inv = None
    for i, rs in enumerate(rsets):
        diffs = [p for p in super_points if p not in rs["points"]]
        z_diff = vanishing_eval(diffs, u)
        if i == 0:
            z_0 = vanishing_eval(rs["points"], u)
            z_0_diff_inv = inv(z_diff)
            z_diff = 1
        else:
            z_diff = z_diff * z_0_diff_inv % R
        inner, r_inner = None, 0
        for key, evals in rs["commitments"]:
            r_eval = eval_poly(lagrange_interpolate(rs["points"], evals), u)
            r_inner = (sy * r_inner + r_eval) % R
            inner = g1_add(g1_mul(inner, sy), commitments[key])
        r_outer = (sv * r_outer + r_inner * z_diff) % R
        outer = g1_add(g1_mul(outer, sv), g1_mul(inner, z_diff))
    right = g1_add(outer, g1_mul(G1, -r_outer))
    right = g1_add(right, g1_mul(h1, -z_0))
    right = g1_add(right, g1_mul(h2, u))
    return opening_check(pk, h2, right, pairing)


def gwc_verify(pk, t, proof, queries, commitments, pairing=False):
    """poly/multiopen/gwc/verifier.rs:16-95"""
    v = t.squeeze()
    u = t.squeeze()
    commitment_multi, eval_multi, witness, witness_with_aux = None, 0, None, None
    for group in gwc_sets(queries):
        z = group[0][2]
        wi = t.read_point()
        witness_with_aux = g1_add(g1_mul(witness_with_aux, u), g1_mul(wi, z))
        witness = g1_add(g1_mul(witness, u), wi)
        commitment_multi = g1_mul(commitment_multi, u)
        eval_multi = eval_multi * u % R
        cb, eb = None, 0
        for key, _, _, ev in group:
            cb = g1_add(g1_mul(cb, v), commitments[key])
            eb = (eb * v + ev) % R
        commitment_multi = g1_add(commitment_multi, cb)
        eval_multi = (eval_multi + eb) % R
    assert t.pos == len(proof), "trailing bytes in the proof"
    right = g1_add(g1_add(witness_with_aux, commitment_multi), g1_mul(G1, -eval_multi))
    return opening_check(pk, witness, right, pairing)
