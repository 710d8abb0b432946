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


# ---- write_cs (helpers.rs:406-456) from the circuit classes below ---------------------------------------------------
# The classes describe their gates / lookups / shuffles as closures over cell VALUES.  Calling a closure with the
# symbolic cells below records the Expression tree halo2's operator overloading would build for the same formula
# (plonk/circuit.rs: `a + b` -> Sum, `a - b` -> Sum(a, Negated(b)), `a * b` -> Product, `a * F` -> Scaled, `-a` -> Negated);
# the serialisation of that tree follows Expression::store (helpers.rs:687-757) and write_cs field by field.  Nothing of
# the product package is used: tests/test_plonk_host.py compares these bytes with formats.cs_store of the product-side
# description of the same circuit -- two independent derivations of the verifying-key digest's preimage.
class _Sym:
    def __init__(self, node):
        self.node = node

    @staticmethod
    def wrap(o):
        return o if isinstance(o, _Sym) else _Sym(("const", o % R))

    def __add__(self, o):
        return _Sym(("sum", self.node, _Sym.wrap(o).node))

    __radd__ = __add__

    def __sub__(self, o):
        return _Sym(("sum", self.node, ("neg", _Sym.wrap(o).node)))

    def __rsub__(self, o):                       # Expression::Constant(o) - self
        return _Sym(("sum", _Sym.wrap(o).node, ("neg", self.node)))

    def __mul__(self, o):
        if isinstance(o, int):
            return _Sym(("scaled", self.node, o % R))
        return _Sym(("prod", self.node, o.node))

    __rmul__ = __mul__

    def __neg__(self):
        return _Sym(("neg", self.node))

    def __mod__(self, _):                        # the closures reduce modulo r as they go: the tree does not change
        return self


_ANY_CODE = {"advice": 0, "fixed": 1, "instance": 2}                   # plonk/circuit.rs:79-86
_EXPR_CODE = {"const": 0, "fixed": 1, "advice": 2, "instance": 3, "neg": 4, "sum": 5, "prod": 6, "scaled": 7}   # helpers.rs:590-599


def _le32(v):
    return (v & 0xFFFFFFFF).to_bytes(4, "little")


def _sym_cells(cs):
    def cell(kind):
        return lambda column, rotation: _Sym((kind, column, rotation))

    return cell("advice"), cell("fixed"), cell("instance")


def _store_expression(cs, node, out):
    kind = node[0]
    out.append(_le32(_EXPR_CODE[kind]))
    if kind == "const":
        out.append(node[1].to_bytes(32, "little"))
    elif kind in ("fixed", "advice", "instance"):
        queries = _cs_get(cs, kind + "_queries", [])
        out += [_le32(queries.index((node[1], node[2]))), _le32(node[1]), _le32(node[2])]
    elif kind == "neg":
        _store_expression(cs, node[1], out)
    elif kind in ("sum", "prod"):
        _store_expression(cs, node[1], out)
        _store_expression(cs, node[2], out)
    else:
        _store_expression(cs, node[1], out)
        out.append(node[2].to_bytes(32, "little"))


def _store_expressions(cs, syms, out):
    out.append(_le32(len(syms)))
    for e in syms:
        _store_expression(cs, _Sym.wrap(e).node, out)


def _first_use_cells(node, cells):
    """(kind, column, rotation) of every query of an expression tree, each once, in depth-first order: this build's
    convention for Gate::queried_cells (the reference records the order of the `meta.query_*` calls of the gate's
    closure, which an expression tree does not carry)"""
    if node[0] in ("fixed", "advice", "instance"):
        if node not in cells:
            cells.append(node)
    elif node[0] != "const":
        for child in node[1:]:
            if isinstance(child, tuple):
                _first_use_cells(child, cells)


def write_cs(cs):
    adv, fix, inst = _sym_cells(cs)
    nadv = cs.num_advice
    per_column = [sum(1 for c, _ in cs.advice_queries if c == i) for i in range(nadv)]
    out = [_le32(nadv), _le32(_cs_get(cs, "num_instance", 0)), _le32(0), _le32(cs.num_fixed), _le32(nadv)]
    out += [_le32(v) for v in per_column]
    out += [_le32(0), _le32(0)]                                  # selector_map, constants: empty after compilation
    for queries in (cs.advice_queries, _cs_get(cs, "instance_queries", []), cs.fixed_queries):
        out.append(_le32(len(queries)))
        for column, rotation in queries:
            out += [_le32(column), _le32(rotation)]
    out.append(_le32(len(cs.perm_columns)))
    for kind, index in cs.perm_columns:
        out += [_le32(index), _le32(_ANY_CODE[kind])]
    lookups = _cs_get(cs, "lookups", [])
    out.append(_le32(len(lookups)))
    for lk in lookups:
        out.append(_le32(len(lk["input_sets"])))
        for st in lk["input_sets"]:
            out.append(_le32(len(st)))
            for inputs in st:
                _store_expressions(cs, inputs(adv, fix, inst), out)
        _store_expressions(cs, lk["table"](adv, fix, inst), out)
    shuffles = _cs_get(cs, "shuffles", [])
    out.append(_le32(len(shuffles)))
    for group in shuffles:
        out.append(_le32(len(group)))
        for inputs, shuffle in group:
            _store_expressions(cs, inputs(adv, fix, inst), out)
            _store_expressions(cs, shuffle(adv, fix, inst), out)
    checks = _cs_get(cs, "range_checks", [])
    out.append(_le32(len(checks)))
    for origin, sort, vmin, vmax, step in checks:
        out += [_le32(origin), _le32(sort), _le32(vmin), _le32(vmax), _le32(step)]
    out.append(_le32(0))                                         # named_advices
    polys = _gates(cs, adv, fix, inst)
    groups = _cs_get(cs, "gate_groups", None) or [1] * len(polys)        # polynomials per create_gate call
    assert sum(groups) == len(polys)
    out.append(_le32(len(groups)))
    at = 0
    for count in groups:
        mine = [_Sym.wrap(p).node for p in polys[at:at + count]]
        at += count
        out.append(_le32(len(mine)))
        cells = []
        for node in mine:
            _store_expression(cs, node, out)
            _first_use_cells(node, cells)
        out.append(_le32(len(cells)))
        for kind, column, rotation in cells:
            out += [_le32(column), _le32(_ANY_CODE[kind]), _le32(rotation)]
    return b"".join(out)


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

    @classmethod
    def cs_bytes(cls):
        """write_cs bytes of this circuit (helpers.rs:406-456), derived from THIS class alone (`write_cs` below): hashed
        into the verifying-key digest"""
        return write_cs(cls)

    @staticmethod
    def gates(adv, fix):
        a, b, c = adv(0, 0), adv(1, 0), adv(2, 0)
        sm, sa, sb, sc = fix(0, 0), fix(1, 0), fix(2, 0), fix(3, 0)
        return [(a * sa + b * sb + a * b * sm + c * sc * (-1)) % R]        # the formula of simple-example-2.rs:201

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


class LookupShuffle:
    """instance column + two logup lookups (one with two input sets, one of them holding two inputs, and a
    duplicated table row) + one shuffle group of two units + an advice/instance copy constraint.
    advice: a b c d e g h g2 h2 p p2 w;  fixed: q t0 t1 u qi;  instance: one column."""
    num_advice, num_fixed, num_instance = 12, 5, 1
    instance_queries = [(0, 0)]
    advice_queries = [(11, 0)] + [(c, 0) for c in range(11)]      # enable_equality(w) first, then a..p2
    fixed_queries = [(0, 0), (4, 0), (1, 0), (2, 0), (3, 0)]       # q, qi (gates), then t0, t1, u (lookups)
    perm_columns = [("advice", 11), ("instance", 0)]
    degree = 6
    blinding_factors = 5
    name = "lookup-shuffle"
    cs_bytes = classmethod(MiniPlonk.cs_bytes.__func__)

    @staticmethod
    def gates(adv, fix, inst):
        q, qi = fix(0, 0), fix(4, 0)
        a, b, w = adv(0, 0), adv(1, 0), adv(11, 0)
        return [q * (a * a + 1 - b) % R, qi * (w - inst(0, 0)) % R]

    lookups = [
        {"table": lambda adv, fix, inst: [fix(1, 0), fix(2, 0)],
         "input_sets": [[lambda adv, fix, inst: [fix(0, 0) * adv(0, 0) % R, fix(0, 0) * adv(1, 0) % R],
                         lambda adv, fix, inst: [adv(2, 0), adv(3, 0)]],
                        [lambda adv, fix, inst: [fix(0, 0) * adv(2, 0) % R, fix(0, 0) * adv(3, 0) % R]]]},
        {"table": lambda adv, fix, inst: [fix(3, 0)],
         "input_sets": [[lambda adv, fix, inst: [adv(4, 0)]]]},
    ]
    shuffles = [[(lambda adv, fix, inst: [adv(5, 0), adv(6, 0)], lambda adv, fix, inst: [adv(7, 0), adv(8, 0)]),
                 (lambda adv, fix, inst: [adv(9, 0)], lambda adv, fix, inst: [adv(10, 0)])]]

    @staticmethod
    def synthesize(k):
        n = 1 << k
        usable = n - 6
        adv = [[0] * n for _ in range(12)]
        fixed = [[0] * n for _ in range(5)]
        t0, t1, u = fixed[1], fixed[2], fixed[3]
        for i in range(1, usable):
            t0[i], t1[i] = i, i * i + 1
        t0[usable - 1], t1[usable - 1] = t0[3], t1[3]          # a duplicated table row
        for i in range(usable):
            u[i] = 7 * i + 3
        for i in range(usable):
            if 1 <= i <= usable // 3:
                fixed[0][i] = 1
                r = 1 + (3 * i) % (usable - 2)
                adv[0][i], adv[1][i] = t0[r], t1[r]
            r = (5 * i + 2) % (usable - 1)
            adv[2][i], adv[3][i] = t0[r], t1[r]
            adv[4][i] = u[(i * i) % usable]
            adv[5][i], adv[6][i] = i + 1, (1000 - i) % R
            adv[9][i] = i * i + 5
        for i in range(usable):
            adv[7][i], adv[8][i] = adv[5][usable - 1 - i], adv[6][usable - 1 - i]
            adv[10][i] = adv[9][(i + 3) % usable]
        adv[11][0] = 42
        fixed[4][0] = 1
        copies = [((0, 0), (1, 0))]
        instances = [[42, 7]]
        return adv, fixed, copies, instances


def wide_class(quads):
    """the big-integer twin of halo2-gpu-specific_amd.circuits.wide: 4 * quads advice columns (a, b, c, d per quad),
    fixed q and t; q * (a b c - d) per quad; quads / 2 logup lookups of two columns each into t; equality on columns 0, 1"""

    class Wide:
        num_advice, num_fixed = 4 * quads, 2
        advice_queries = [(c, 0) for c in range(4 * quads)]
        fixed_queries = [(0, 0), (1, 0)]
        perm_columns = [("advice", 0), ("advice", 1)]
        degree = 5
        blinding_factors = 5
        name = "wide-%d" % quads
        cs_bytes = classmethod(MiniPlonk.cs_bytes.__func__)

        @staticmethod
        def gates(adv, fix):
            q = fix(0, 0)
            return [q * (adv(4 * i, 0) * adv(4 * i + 1, 0) % R * adv(4 * i + 2, 0) + adv(4 * i + 3, 0) * (-1)) % R for i in range(quads)]

        gate_groups = [quads]                 # one create_gate call with a polynomial per quad

        lookups = [{"table": lambda adv, fix, inst: [fix(1, 0)],
                    "input_sets": [[lambda adv, fix, inst, l=l: [adv(8 * l, 0)], lambda adv, fix, inst, l=l: [adv(8 * l + 4, 0)]]]}
                   for l in range(quads // 2)]

        @staticmethod
        def synthesize(k):
            """same witness as circuits.wide_synthesize, as Python integers"""
            n = 1 << k
            usable = n - 6
            T = min(usable, 1 << 16)
            adv = [[0] * n for _ in range(4 * quads)]
            fixed = [[0] * n for _ in range(2)]
            for r in range(usable):
                fixed[0][r] = 1
            for i in range(T):
                fixed[1][i] = i
            mask = (1 << 64) - 1
            for qd in range(quads):
                for r in range(usable):
                    v = [(((r * 2654435761 + 40503 * (4 * qd + j) + 7) & mask) >> 5) % T for j in range(3)]
                    if qd == 0:
                        v[1] = 3 % T if r == 0 else (((((r - 1) * 2654435761 + 7) & mask) >> 5) % T)
                    for j in range(3):
                        adv[4 * qd + j][r] = v[j]
                    adv[4 * qd + 3][r] = v[0] * v[1] * v[2]
            m = min(usable - 1, 1 << 16)
            copies = [((0, r), (1, r + 1)) for r in range(m)]
            return adv, fixed, copies

    return Wide


class LookupApi:
    """the big-integer twin of circuits.lookup_api (examples/lookup_api.rs): advice input_0..2; fixed s_0, s_1, table.
    The lookups are listed as the reference's chunking pass leaves them (BTreeMap order of the table identifiers): the
    `lookup_any` first, then the three lookups into the table column packed into one input set"""
    num_advice, num_fixed = 3, 3
    advice_queries = [(0, 0), (1, 0), (2, 0)]
    fixed_queries = [(0, 0), (2, 0), (1, 0)]
    perm_columns = []
    degree = 6
    blinding_factors = 5
    name = "lookup-api"
    cs_bytes = classmethod(MiniPlonk.cs_bytes.__func__)

    @staticmethod
    def gates(adv, fix):
        return [fix(0, 0) * (adv(0, 0) * 1 - adv(1, 0)) % R]          # examples/lookup_api.rs:66-71

    lookups = [
        {"table": lambda adv, fix, inst: [fix(0, 0) * adv(1, 0) % R, fix(1, 0) * adv(2, 0) % R],
         "input_sets": [[lambda adv, fix, inst: [fix(0, 0) * adv(0, 0) % R, fix(1, 0) * adv(0, 0) % R]]]},
        {"table": lambda adv, fix, inst: [fix(2, 0)],
         "input_sets": [[lambda adv, fix, inst: [adv(0, 0)], lambda adv, fix, inst: [adv(1, 0) * 2 % R],
                         lambda adv, fix, inst: [adv(2, 0)]]]},
    ]

    @staticmethod
    def synthesize(k):
        n = 1 << k
        adv = [[0] * n for _ in range(3)]
        fixed = [[0] * n for _ in range(3)]
        adv[0][0], adv[1][0], fixed[0][0] = 1, 1, 1
        adv[0][1], adv[2][1], fixed[1][1] = 3, 3, 1
        for i in range(9):
            fixed[2][i] = i
        return adv, fixed, []


class ShuffleApiGroup:
    """the big-integer twin of circuits.shuffle_api_group (examples/shuffle_api_group.rs): advice in_0..4, sh_0..4; fixed
    s_in0, s_in1, s_sh0, s_sh1; the four shuffles grouped as the chunking pass groups them at degree 5:
    [shuffle1, shuffle2], [shuffle3], [shuffle4]"""
    num_advice, num_fixed = 10, 4
    advice_queries = [(0, 0), (1, 0), (5, 0), (6, 0), (2, 0), (7, 0), (3, 0), (8, 0), (4, 0), (9, 0)]
    fixed_queries = [(0, 0), (2, 0), (1, 0), (3, 0)]
    perm_columns = []
    degree = 5
    blinding_factors = 5
    name = "shuffle-api-group"
    cs_bytes = classmethod(MiniPlonk.cs_bytes.__func__)

    @staticmethod
    def gates(adv, fix):
        return [fix(0, 0) * (adv(0, 0) - adv(1, 0)) % R]

    shuffles = [
        [(lambda adv, fix, inst: [adv(0, 0), adv(1, 0)], lambda adv, fix, inst: [adv(5, 0), adv(6, 0)]),
         (lambda adv, fix, inst: [adv(2, 0)], lambda adv, fix, inst: [adv(7, 0)])],
        [(lambda adv, fix, inst: [adv(3, 0) * fix(0, 0) % R], lambda adv, fix, inst: [adv(8, 0) * fix(2, 0) % R])],
        [(lambda adv, fix, inst: [adv(4, 0) * fix(0, 0) % R * fix(1, 0) % R],
          lambda adv, fix, inst: [adv(9, 0) * fix(2, 0) % R * fix(3, 0) % R])],
    ]

    @staticmethod
    def synthesize(k, input0=(1, 2, 4, 1), input1=(4, 1, 1, 2)):
        n = 1 << k
        adv = [[0] * n for _ in range(10)]
        fixed = [[0] * n for _ in range(4)]
        for i, (a, b) in enumerate(zip(input0, input1)):
            for c in range(5):
                adv[c][i], adv[5 + c][i] = a, b
            for f in fixed:
                f[i] = 1
        return adv, fixed, []


class LookupApiSet:
    """the big-integer twin of circuits.lookup_api_set (examples/lookup_api_set.rs): six advice inputs; fixed s_0, s_1,
    table; ONE logup argument into the table column whose six traced lookups the chunking pass packs into four input
    sets: {input_0}, {2 input_1, input_2}, {10 input_3, input_4}, {input_5}"""
    num_advice, num_fixed = 6, 3
    advice_queries = [(c, 0) for c in range(6)]
    fixed_queries = [(0, 0), (2, 0)]                 # s_0 (gate), table; s_1 is allocated and never queried
    perm_columns = []
    degree = 4
    blinding_factors = 5
    name = "lookup-api-set"
    cs_bytes = classmethod(MiniPlonk.cs_bytes.__func__)

    @staticmethod
    def gates(adv, fix):
        return [fix(0, 0) * (adv(0, 0) * 1 - adv(1, 0)) % R]          # examples/lookup_api_set.rs:58-63

    lookups = [
        {"table": lambda adv, fix, inst: [fix(2, 0)],
         "input_sets": [[lambda adv, fix, inst: [adv(0, 0)]],
                        [lambda adv, fix, inst: [adv(1, 0) * 2 % R], lambda adv, fix, inst: [adv(2, 0)]],
                        [lambda adv, fix, inst: [adv(3, 0) * 10 % R], lambda adv, fix, inst: [adv(4, 0)]],
                        [lambda adv, fix, inst: [adv(5, 0)]]]},
    ]

    @staticmethod
    def synthesize(k):
        n = 1 << k
        adv = [[0] * n for _ in range(6)]
        fixed = [[0] * n for _ in range(3)]
        for a in adv:
            a[0], a[1] = 1, 3
        fixed[0][0], fixed[1][1] = 1, 1
        for i in range(100):
            fixed[2][i] = i
        return adv, fixed, []


class ShuffleApi:
    """the big-integer twin of circuits.shuffle_api (examples/shuffle_api.rs): advice input_0, input_1, shuffle_0,
    shuffle_1; fixed s_input, s_shuffle; one shuffle unit of two (input, shuffle) pairs"""
    num_advice, num_fixed = 4, 2
    advice_queries = [(0, 0), (1, 0), (2, 0), (3, 0)]
    fixed_queries = [(0, 0), (1, 0)]
    perm_columns = []
    degree = 4
    blinding_factors = 5
    name = "shuffle-api"
    cs_bytes = classmethod(MiniPlonk.cs_bytes.__func__)

    @staticmethod
    def gates(adv, fix):
        return [fix(0, 0) * (adv(0, 0) * 10 - adv(1, 0)) % R]         # examples/shuffle_api.rs:59-64

    shuffles = [[(lambda adv, fix, inst: [fix(0, 0) * adv(0, 0) % R, fix(0, 0) * adv(1, 0) % R],
                  lambda adv, fix, inst: [fix(1, 0) * adv(2, 0) % R, fix(1, 0) * adv(3, 0) % R])]]

    @staticmethod
    def synthesize(k, input0=(1, 2, 4, 1), shuffle0=(4, 1, 1, 2)):
        n = 1 << k
        adv = [[0] * n for _ in range(4)]
        fixed = [[0] * n for _ in range(2)]
        for i, (a, b) in enumerate(zip(input0, shuffle0)):
            adv[0][i], adv[1][i], adv[2][i], adv[3][i] = a, 10 * a, b, 10 * b
            fixed[0][i] = fixed[1][i] = 1
        return adv, fixed, []


def _const_like(x, v):
    """Expression::Constant(v) next to the cell `x`: an integer among integers, a constant node among traced cells"""
    return _Sym(("const", v % R)) if isinstance(x, _Sym) else v % R


def shuffle_gates_class(width=4, theta=111, beta=222):
    """the big-integer twin of circuits.shuffle_gates (examples/shuffle.rs): the shuffle written as three gates over a
    running-product advice column; fixed q_shuffle, q_first, q_last; advice original[W], shuffled[W], z"""

    class ShuffleGates:
        num_advice, num_fixed = 2 * width + 1, 3
        advice_queries = [(2 * width, 0)] + [(c, 0) for c in range(2 * width)] + [(2 * width, 1)]
        fixed_queries = [(1, 0), (2, 0), (0, 0)]
        perm_columns = []
        degree = 3
        blinding_factors = 5
        name = "shuffle-gates-%d-%d-%d" % (width, theta, beta)
        cs_bytes = classmethod(MiniPlonk.cs_bytes.__func__)

        @staticmethod
        def gates(adv, fix):
            z, z_next = adv(2 * width, 0), adv(2 * width, 1)
            one, th, be = _const_like(z, 1), _const_like(z, theta), _const_like(z, beta)

            def compress(first):
                acc = adv(first, 0)
                for c in range(first + 1, first + width):
                    acc = (acc * th + adv(c, 0)) % R
                return acc

            return [fix(1, 0) * (one - z) % R, fix(2, 0) * (one - z) % R,
                    fix(0, 0) * (z * (compress(0) + be) - z_next * (compress(width) + be)) % R]

        @staticmethod
        def synthesize(k, height=32, seed=0x5348554646):
            """the same seeded walk as circuits.shuffle_gates_witness, restated here"""
            import random

            n = 1 << k
            rnd = random.Random(seed)
            original = [[rnd.randrange(R) for _ in range(height)] for _ in range(width)]
            order = list(range(height))
            for row in range(height - 1, 0, -1):
                other = rnd.getrandbits(32) % row
                order[row], order[other] = order[other], order[row]
            shuffled = [[col[i] for i in order] for col in original]
            fold = lambda cols, i: sum(col[i] * pow(theta, len(cols) - 1 - j, R) for j, col in enumerate(cols)) % R  # noqa: E731
            z = [1]
            for i in range(height):
                z.append(z[-1] * (fold(original, i) + beta) * inv((fold(shuffled, i) + beta) % R) % R)
            pad = lambda col: col + [0] * (n - len(col))  # noqa: E731
            return ([pad(c) for c in original] + [pad(c) for c in shuffled] + [pad(z)],
                    [pad([1] * height), pad([1]), pad([0] * height + [1])], [])

    return ShuffleGates


def range_check_class(vmin, vmax, step):
    """the big-integer twin of halo2-gpu-specific_amd.circuits.range_check (examples/range-check.rs; the gate and the
    shuffle of `advice_column_range`, plonk/circuit.rs:1769-1826): advice origin (0) and sort (1); fixed l_0, l_active,
    l_last_active; no permutation columns"""

    class RangeCheck:
        num_advice, num_fixed = 2, 3
        advice_queries = [(1, 0), (1, 1), (0, 0)]
        fixed_queries = [(0, 0), (2, 0), (1, 0)]
        perm_columns = []
        degree = max(3, step + 2)
        blinding_factors = 5
        name = "range-check-%d-%d-%d" % (vmin, vmax, step)
        cs_bytes = classmethod(MiniPlonk.cs_bytes.__func__)

        @staticmethod
        def gates(adv, fix):
            cur, nxt = adv(1, 0), adv(1, 1)
            prod = None
            for i in range(step + 1):
                term = (nxt - cur - (step - i)) % R
                prod = term if prod is None else prod * term % R
            return [fix(0, 0) * (vmin - cur) % R, fix(2, 0) * (vmax - cur) % R, (fix(1, 0) - fix(2, 0)) * prod % R]

        range_checks = [(0, 1, vmin, vmax, step)]
        gate_groups = [3]                     # one create_gate call with three polynomials (circuit.rs:1793-1818)

        shuffles = [[(lambda adv, fix, inst: [adv(0, 0)], lambda adv, fix, inst: [adv(1, 0)])]]

        @staticmethod
        def complete(k, origin):
            """plant the range from the last usable row upwards, sort (plonk/prover.rs:1699-1783)"""
            n = 1 << k
            usable = n - 6
            origin = list(origin)
            planted, cur = [], vmin
            while cur < vmax:
                planted.append(cur)
                cur = min(cur + step, vmax)
            planted.append(vmax)
            row = usable - 1
            for v in planted:
                origin[row] = v
                row -= 1
            assert row >= 0
            return [origin, sorted(origin[:usable]) + [0] * (n - usable)]

    return RangeCheck


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


def vk_digest(cs, dom, fixed_commitments, perm_commitments):
    """stand-in for `format!("{:?}", vk.pinned())` (plonk.rs:91-109): same framing (u64 length, Blake2b-512 with
    the Halo2-Verify-Key personalisation, from_bytes_wide) over the domain, both moduli, the canonical write_cs bytes
    of the constraint system (`write_cs` above, derived from the circuit class of this file alone: the digest's preimage is
    not taken from the product) and the commitments"""
    cs_bytes = cs.cs_bytes()
    body = [b"halo2-hip-vk-v2", dom.k.to_bytes(4, "little"), dom.extended_k.to_bytes(4, "little"),
            dom.omega.to_bytes(32, "little"), R.to_bytes(32, "little"), Q.to_bytes(32, "little"),
            len(cs_bytes).to_bytes(4, "little"), cs_bytes]
    for group in (fixed_commitments, perm_commitments):
        body.append(len(group).to_bytes(4, "little"))
        body += [point_to_bytes(p) for p in group]
    body = b"".join(body)
    h = hashlib.blake2b(digest_size=64, person=b"Halo2-Verify-Key")
    h.update(len(body).to_bytes(8, "little"))
    h.update(body)
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
    pk.transcript_repr = vk_digest(cs, dom, pk.fixed_commitments, pk.perm_commitments)
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


def _cs_get(cs, name, default):
    return getattr(cs, name, default)


def _gates(cs, adv, fix, inst):
    try:
        return list(cs.gates(adv, fix, inst))
    except TypeError:
        return list(cs.gates(adv, fix))


def _compress(values, theta):
    """evaluate_with_theta / the verifier's compress_expressions: fold(0, acc * theta + v)"""
    acc = 0
    for v in values:
        acc = (acc * theta + v) % R
    return acc


def _argument_expressions(cs, adv, fix, inst, theta, beta, gamma, l0, ll, la, point_beta, perm, lookups, shuffles, sig,
                          perm_vals):
    """everything after the gates in the order of plonk/evaluation.rs:1017-1219 == plonk/verifier.rs:300-383.
    perm: list of (z, z_next, z_prev_last); lookups: list of (m, [(z, z_next, z_prev_last)]); shuffles: list of
    (z, z_next); sig[i], perm_vals[i]: sigma_i / the permuted column's value at the point."""
    chunk = cs.degree - 2
    cols = cs.perm_columns
    exprs = []
    if perm:
        exprs.append(l0 * (1 - perm[0][0]) % R)
        exprs.append(ll * (perm[-1][0] * perm[-1][0] - perm[-1][0]) % R)
        for i in range(1, len(perm)):
            exprs.append(l0 * (perm[i][0] - perm[i][2]) % R)
        for i in range(len(perm)):
            left, right = perm[i][1], perm[i][0]
            cur = point_beta * pow(DELTA, i * chunk, R) % R
            for ci in range(i * chunk, min((i + 1) * chunk, len(cols))):
                left = left * (perm_vals[ci] + beta * sig[ci] + gamma) % R
                right = right * (perm_vals[ci] + cur + gamma) % R
                cur = cur * DELTA % R
            exprs.append(la * (left - right) % R)
    for lk, (m, zsets) in zip(_cs_get(cs, "lookups", []), lookups):
        tau = (_compress(lk["table"](adv, fix, inst), theta) + beta) % R
        phis = [[(_compress(e(adv, fix, inst), theta) + beta) % R for e in st] for st in lk["input_sets"]]

        def prod_sum(phi):
            prod = 1
            for v in phi:
                prod = prod * v % R
            sm = 0
            for i in range(len(phi)):
                term = 1
                for j, v in enumerate(phi):
                    if j != i:
                        term = term * v % R
                sm = (sm + term) % R
            return prod, sm

        exprs.append(l0 * zsets[0][0] % R)
        exprs.append(ll * zsets[-1][0] % R)
        prod, sm = prod_sum(phis[0])
        exprs.append(la * (((zsets[0][1] - zsets[0][0]) * tau + m) * prod - tau * sm) % R)
        for i in range(1, len(zsets)):
            exprs.append(l0 * (zsets[i][0] - zsets[i][2]) % R)
        for i in range(1, len(zsets)):
            prod, sm = prod_sum(phis[i])
            exprs.append(la * ((zsets[i][1] - zsets[i][0]) * prod - sm) % R)
    for group, (z, z_next) in zip(_cs_get(cs, "shuffles", []), shuffles):
        a = b = 1
        for i, (inp, shf) in enumerate(group):
            ch = pow(beta, i + 1, R)
            a = a * (_compress(inp(adv, fix, inst), theta) + ch) % R
            b = b * (_compress(shf(adv, fix, inst), theta) + ch) % R
        exprs.append(l0 * (1 - z) % R)
        exprs.append(ll * (z * z - z) % R)
        exprs.append(la * (z_next * b - z * a) % R)
    return exprs


def _is_multi(advice_in):
    """several circuit instances: advice_in is a list of column lists (a column is a list of ints)"""
    return len(advice_in) > 0 and len(advice_in[0]) > 0 and isinstance(advice_in[0][0], (list, tuple))


def create_proof(pk, advice_in, rng, use_gwc=False, instances=()):
    """plonk/prover.rs:206-850 (create_proof_ext).  One circuit instance: advice_in = its columns, instances = its
    instance columns; several (`circuits: &[ConcreteCircuit]`): lists of those, every phase circuit by circuit."""
    cs, dom = pk.cs, pk.dom
    n, bf = dom.n, cs.blinding_factors
    usable = n - (bf + 1)
    multi = _is_multi(advice_in)
    advice_sets = advice_in if multi else [advice_in]
    instance_sets = instances if multi else [instances]
    assert len(advice_sets) == len(instance_sets)
    t = Transcript()
    t.common_scalar(pk.transcript_repr)
    circuits = []
    # instance columns (prover.rs:85-162): zero-padded, committed, hashed but not written
    for inst in instance_sets:
        assert len(inst) == _cs_get(cs, "num_instance", 0)
        C = {"instance": []}
        for vals in inst:
            assert len(vals) <= usable, "InstanceTooLarge"
            C["instance"].append(list(vals) + [0] * (n - len(vals)))
        C["instance_polys"] = [dom.lagrange_to_coeff(col) for col in C["instance"]]
        for p in C["instance_polys"]:
            t.common_point(commit(pk, p))
        circuits.append(C)
    # advice: blinding rows (prover.rs:281-289), commitments
    for C, adv_in in zip(circuits, advice_sets):
        C["advice"] = [col[:] for col in adv_in]
        for col in C["advice"]:
            for r in range(usable, n):
                col[r] = rng.u16()
        C["advice_polys"] = [dom.lagrange_to_coeff(col) for col in C["advice"]]
        for p in C["advice_polys"]:
            t.write_point(commit(pk, p))
    theta = t.squeeze()

    def row_access(C, i):
        return (lambda c, r: C["advice"][c][(i + r) % n], lambda c, r: pk.fixed_values[c][(i + r) % n],
                lambda c, r: C["instance"][c][(i + r) % n])

    # lookups: compressed inputs / table and the multiplicities (logup/prover.rs:63-240)
    for C in circuits:
        C["lk"] = []
        for lk in _cs_get(cs, "lookups", []):
            table = [_compress(lk["table"](*row_access(C, i)), theta) for i in range(n)]
            inputs = [[[_compress(e(*row_access(C, i)), theta) for i in range(n)] for e in st] for st in lk["input_sets"]]
            first = {}
            for i in range(usable):
                first.setdefault(table[i], i)          # a duplicated table value is credited to its first row
            m = [0] * n
            for st in inputs:
                for col in st:
                    for i in range(usable):
                        m[first[col[i]]] += 1          # KeyError = "logup binary_search_by_key should hit"
            for i in range(usable, n):
                m[i] = rng.u16()
            C["lk"].append({"table": table, "inputs": inputs, "m": m, "m_poly": dom.lagrange_to_coeff(m)})
        # shuffles: compressed expressions (shuffle/prover.rs:40-80)
        C["sh"] = []
        for group in _cs_get(cs, "shuffles", []):
            C["sh"].append([([_compress(inp(*row_access(C, i)), theta) for i in range(n)],
                             [_compress(shf(*row_access(C, i)), theta) for i in range(n)]) for inp, shf in group])
    for C in circuits:
        for st in C["lk"]:
            t.write_point(commit(pk, st["m_poly"]))
    beta = t.squeeze()
    gamma = t.squeeze()
    # permutation grand products (permutation/prover.rs:47-165), circuit by circuit
    chunk = cs.degree - 2
    cols = cs.perm_columns
    for C in circuits:
        colvals = {"advice": C["advice"], "fixed": pk.fixed_values, "instance": C["instance"]}
        zs, last_z = [], 1
        for si in range(0, len(cols), chunk):
            mv = [1] * n
            for ci in range(si, min(si + chunk, len(cols))):
                vals = colvals[cols[ci][0]][cols[ci][1]]
                for i in range(n):
                    mv[i] = mv[i] * (beta * pk.sigma_values[ci][i] + gamma + vals[i]) % R
            mv = [inv(v) for v in mv]
            dw = pow(DELTA, si, R)
            for ci in range(si, min(si + chunk, len(cols))):
                vals = colvals[cols[ci][0]][cols[ci][1]]
                for i in range(n):
                    mv[i] = mv[i] * (dw * beta + gamma + vals[i]) % R
                    dw = dw * dom.omega % R
                dw = dw * DELTA % R
            z = [last_z]
            for i in range(n - 1):
                z.append(z[i] * mv[i] % R)
            for i in range(n - bf, n):
                z[i] = rng.fr()
            last_z = z[usable]
            zs.append(z)
        C["z_polys"] = [dom.lagrange_to_coeff(z) for z in zs]
    # lookup grand sums (logup/prover.rs:243-415, blinding prover.rs:446-465)
    for C in circuits:
        for st in C["lk"]:
            st["z_polys"] = []
            last = 0
            for si, cols_in in enumerate(st["inputs"]):
                g = [0] * n
                for col in cols_in:
                    for i in range(n):
                        g[i] = (g[i] + inv((beta + col[i]) % R)) % R
                if si == 0:
                    for i in range(n):
                        g[i] = (g[i] - inv((beta + st["table"][i]) % R) * st["m"][i]) % R
                z = [last]
                for i in range(usable):
                    z.append((z[i] + g[i]) % R)
                last = z[usable]
                z += [rng.fr() for _ in range(bf)]
                assert len(z) == n
                st["z_polys"].append(dom.lagrange_to_coeff(z))
            assert last == 0, "the lookup does not hold"
    # shuffle products (shuffle/prover.rs:82-150, blinding prover.rs:512-530)
    for C in circuits:
        C["sh_polys"] = []
        for group in C["sh"]:
            prod = [1] * n
            for i, (_, shf) in enumerate(group):
                ch = pow(beta, i + 1, R)
                for r in range(n):
                    prod[r] = prod[r] * (ch + shf[r]) % R
            prod = [inv(v) for v in prod]
            for i, (inp, _) in enumerate(group):
                ch = pow(beta, i + 1, R)
                for r in range(n):
                    prod[r] = prod[r] * (ch + inp[r]) % R
            z = [1]
            for i in range(usable):
                z.append(z[i] * prod[i] % R)
            assert z[usable] == 1, "the shuffle does not hold"
            z += [rng.fr() for _ in range(bf)]
            C["sh_polys"].append(dom.lagrange_to_coeff(z))
    for C in circuits:
        for p in C["z_polys"]:
            t.write_point(commit(pk, p))
    for C in circuits:
        for st in C["lk"]:
            for p in st["z_polys"]:
                t.write_point(commit(pk, p))
    for C in circuits:
        for p in C["sh_polys"]:
            t.write_point(commit(pk, p))
    # vanishing argument: random polynomial (vanishing/prover.rs:40-67)
    random_poly = rng.random_poly(n)
    t.write_point(commit(pk, random_poly))
    y = t.squeeze()
    # h(X) on the extended coset: ONE Horner fold in y over the expressions of every circuit (evaluation.rs:839-1100)
    ext = dom.coeff_to_extended
    fix_c = [ext(p) for p in pk.fixed_polys]
    sig_c = [ext(p) for p in pk.sigma_polys]
    for C in circuits:
        C["adv_c"] = [ext(p) for p in C["advice_polys"]]
        C["ins_c"] = [ext(p) for p in C["instance_polys"]]
        C["z_c"] = [ext(p) for p in C["z_polys"]]
        C["lk_c"] = [(ext(st["m_poly"]), [ext(p) for p in st["z_polys"]]) for st in C["lk"]]
        C["sh_c"] = [ext(p) for p in C["sh_polys"]]
    en = dom.extended_n
    scale = en // n
    last_rot = -(bf + 1)
    h = [0] * en
    point = ZETA
    for j in range(en):
        jn, jl = (j + scale) % en, (j + last_rot * scale) % en
        acc = 0
        for C in circuits:
            adv_c, ins_c, z_c = C["adv_c"], C["ins_c"], C["z_c"]
            colc = {"advice": adv_c, "fixed": fix_c, "instance": ins_c}
            adv = lambda c, r: adv_c[c][(j + r * scale) % en]  # noqa: E731
            fix = lambda c, r: fix_c[c][(j + r * scale) % en]  # noqa: E731
            ins = lambda c, r: ins_c[c][(j + r * scale) % en]  # noqa: E731
            exprs = _gates(cs, adv, fix, ins)
            exprs += _argument_expressions(
                cs, adv, fix, ins, theta, beta, gamma, pk.l0[j], pk.l_last[j], pk.l_active_row[j], beta * point % R,
                [(z_c[i][j], z_c[i][jn], z_c[i - 1][jl] if i else None) for i in range(len(z_c))],
                [(mc[j], [(zc[i][j], zc[i][jn], zc[i - 1][jl] if i else None) for i in range(len(zc))]) for mc, zc in C["lk_c"]],
                [(zc[j], zc[jn]) for zc in C["sh_c"]],
                [c[j] for c in sig_c], [colc[kd][ix][j] for kd, ix in cs.perm_columns])
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
    for C in circuits:
        for c, rot in _cs_get(cs, "instance_queries", []):
            t.write_scalar(eval_poly(C["instance_polys"][c], dom.rotate_omega(x, rot)))
    for C in circuits:
        for c, rot in cs.advice_queries:
            t.write_scalar(eval_poly(C["advice_polys"][c], dom.rotate_omega(x, rot)))
    for c, rot in cs.fixed_queries:
        t.write_scalar(eval_poly(pk.fixed_polys[c], dom.rotate_omega(x, rot)))
    h_poly = fold(reversed(pieces), xn, n)
    t.write_scalar(eval_poly(random_poly, x))
    for p in pk.sigma_polys:
        t.write_scalar(eval_poly(p, x))
    x_next, x_last = dom.rotate_omega(x, 1), dom.rotate_omega(x, last_rot)

    def write_set_evals(polys_):
        for i, p in enumerate(polys_):
            t.write_scalar(eval_poly(p, x))
            t.write_scalar(eval_poly(p, x_next))
            if i + 1 < len(polys_):
                t.write_scalar(eval_poly(p, x_last))

    for C in circuits:
        write_set_evals(C["z_polys"])
    for C in circuits:
        for st in C["lk"]:
            t.write_scalar(eval_poly(st["m_poly"], x))
            write_set_evals(st["z_polys"])
    for C in circuits:
        for p in C["sh_polys"]:
            t.write_scalar(eval_poly(p, x))
            t.write_scalar(eval_poly(p, x_next))
    # multiopen query list (prover.rs:792-840)
    polys = {}
    queries = []

    def q(key, poly, rot):
        polys[key] = poly
        pt = dom.rotate_omega(x, rot)
        queries.append((key, rot, pt, eval_poly(poly, pt)))

    def open_sets(name, polys_):
        for i, p in enumerate(polys_):
            q((name, i), p, 0)
            q((name, i), p, 1)
        for i in reversed(range(len(polys_) - 1)):
            q((name, i), polys_[i], last_rot)

    for ci, C in enumerate(circuits):
        for c, rot in _cs_get(cs, "instance_queries", []):
            q(("instance", ci, c), C["instance_polys"][c], rot)
        for c, rot in cs.advice_queries:
            q(("advice", ci, c), C["advice_polys"][c], rot)
        open_sets("z%d" % ci, C["z_polys"])
        for li, st in enumerate(C["lk"]):
            q(("lookup_m", ci, li), st["m_poly"], 0)
            open_sets("lookup_z%d_%d" % (ci, li), st["z_polys"])
        for i, p in enumerate(C["sh_polys"]):
            q(("shuffle_z", ci, i), p, 0)
            q(("shuffle_z", ci, i), p, 1)
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


def verify_proof(pk, proof, use_gwc=False, pairing=False, instances=(), circuits=None):
    """True iff the proof is accepted (plonk/verifier.rs:128-507).  `circuits` = number of circuit instances in the proof
    (then `instances` is a list with one list of instance columns per circuit); None = one circuit, `instances` its columns."""
    cs, dom = pk.cs, pk.dom
    n, bf = dom.n, cs.blinding_factors
    instance_sets = [instances] if circuits is None else list(instances)
    ncirc = len(instance_sets)
    assert circuits is None or circuits == ncirc
    t = Transcript(proof)
    t.common_scalar(pk.transcript_repr)
    instance_commitments = []
    for inst in instance_sets:
        assert len(inst) == _cs_get(cs, "num_instance", 0)
        coms = []
        for vals in inst:
            assert len(vals) <= n - (bf + 1)
            coms.append(commit(pk, dom.lagrange_to_coeff(list(vals) + [0] * (n - len(vals)))))
            t.common_point(coms[-1])
        instance_commitments.append(coms)
    advice_commitments = [[t.read_point() for _ in range(cs.num_advice)] for _ in range(ncirc)]
    theta = t.squeeze()
    lookups_cs, shuffles_cs = _cs_get(cs, "lookups", []), _cs_get(cs, "shuffles", [])
    m_commitments = [[t.read_point() for _ in lookups_cs] for _ in range(ncirc)]
    beta = t.squeeze()
    gamma = t.squeeze()
    chunk = cs.degree - 2
    nsets = (len(cs.perm_columns) + chunk - 1) // chunk
    z_commitments = [[t.read_point() for _ in range(nsets)] for _ in range(ncirc)]
    lk_z_commitments = [[[t.read_point() for _ in lk["input_sets"]] for lk in lookups_cs] for _ in range(ncirc)]
    sh_commitments = [[t.read_point() for _ in shuffles_cs] for _ in range(ncirc)]
    random_commitment = t.read_point()
    y = t.squeeze()
    h_commitments = [t.read_point() for _ in range(dom.quotient_poly_degree)]
    x = t.squeeze()
    instance_queries = _cs_get(cs, "instance_queries", [])
    instance_evals = [[t.read_scalar() for _ in instance_queries] for _ in range(ncirc)]
    advice_evals = [[t.read_scalar() for _ in cs.advice_queries] for _ in range(ncirc)]
    fixed_evals = [t.read_scalar() for _ in cs.fixed_queries]
    random_eval = t.read_scalar()
    sigma_evals = [t.read_scalar() for _ in cs.perm_columns]

    def read_set_evals(count):
        out = []
        for i in range(count):
            e = {"cur": t.read_scalar(), "next": t.read_scalar()}
            if i + 1 < count:
                e["last"] = t.read_scalar()
            out.append(e)
        return out

    z_evals = [read_set_evals(nsets) for _ in range(ncirc)]
    lk_evals = []
    for _ in range(ncirc):
        per = []
        for lk in lookups_cs:
            m_eval = t.read_scalar()
            per.append((m_eval, read_set_evals(len(lk["input_sets"]))))
        lk_evals.append(per)
    sh_evals = [[(t.read_scalar(), t.read_scalar()) for _ in shuffles_cs] for _ in range(ncirc)]
    xn = pow(x, n, R)
    last_rot = -(bf + 1)
    l_evals = dom.l_i_range(x, xn, range(last_rot, 1))
    l_last, l_blind, l_0 = l_evals[0], sum(l_evals[1:1 + bf]) % R, l_evals[1 + bf]
    fix = lambda c, r: fixed_evals[cs.fixed_queries.index((c, r))]  # noqa: E731

    def triples(evs):
        return [(evs[i]["cur"], evs[i]["next"], evs[i - 1]["last"] if i else None) for i in range(len(evs))]

    expected_h = 0
    for ci in range(ncirc):
        adv = lambda c, r, ci=ci: advice_evals[ci][cs.advice_queries.index((c, r))]  # noqa: E731
        ins = lambda c, r, ci=ci: instance_evals[ci][instance_queries.index((c, r))]  # noqa: E731
        getters = {"advice": adv, "fixed": fix, "instance": ins}
        exprs = _gates(cs, adv, fix, ins)
        exprs += _argument_expressions(
            cs, adv, fix, ins, theta, beta, gamma, l_0, l_last, (1 - (l_last + l_blind)) % R, beta * x % R,
            triples(z_evals[ci]), [(m_eval, triples(evs)) for m_eval, evs in lk_evals[ci]], sh_evals[ci],
            sigma_evals, [getters[kd](ix, 0) for kd, ix in cs.perm_columns])
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

    def open_sets(name, coms, evs):
        for i in range(len(coms)):
            q((name, i), coms[i], 0, evs[i]["cur"])
            q((name, i), coms[i], 1, evs[i]["next"])
        for i in reversed(range(len(coms) - 1)):
            q((name, i), coms[i], last_rot, evs[i]["last"])

    for ci in range(ncirc):
        for (c, rot), ev in zip(instance_queries, instance_evals[ci]):
            q(("instance", ci, c), instance_commitments[ci][c], rot, ev)
        for (c, rot), ev in zip(cs.advice_queries, advice_evals[ci]):
            q(("advice", ci, c), advice_commitments[ci][c], rot, ev)
        open_sets("z%d" % ci, z_commitments[ci], z_evals[ci])
        for li, (m_eval, evs) in enumerate(lk_evals[ci]):
            q(("lookup_m", ci, li), m_commitments[ci][li], 0, m_eval)
            open_sets("lookup_z%d_%d" % (ci, li), lk_z_commitments[ci][li], evs)
        for i, (cur, nxt) in enumerate(sh_evals[ci]):
            q(("shuffle_z", ci, i), sh_commitments[ci][i], 0, cur)
            q(("shuffle_z", ci, i), sh_commitments[ci][i], 1, nxt)
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
