"""The generated multiplier schedules (tools/gen_fp_mul.py -> csrc/fp_mul_gen.hpp), executed line by line with Python
integers: every schedule -- product, square, a * b + c * d under one reduction -- must give the Montgomery result for
random and edge inputs below 2^254, and a multiply-add emitted as "cannot carry" must indeed never carry.  (The device
runs the same lines as v_mad_u64_u32 / v_addc_co_u32; tools/mulbench checks the compiled code against the host product.)
"""
import importlib.util
import os
import random
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("gen_fp_mul", os.path.join(ROOT, "tools", "gen_fp_mul.py"))
gen = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gen)

M32, M64 = (1 << 32) - 1, (1 << 64) - 1
MAD = re.compile(r"H2_MAD_(FREE|SET|ACC|WRAP)_[VS]\(([^,]+), ([^)]+)\);")


def limbs(v):
    return [(v >> (32 * i)) & M32 for i in range(8)]


def run(lines, p, ops, raw=False):
    """ops: {"a": int, "b": int, ...}; returns the schedule's result after the final conditional subtraction (`raw`: before
    it -- the wide schedule returns its value unreduced)"""
    mod = limbs(p)
    inv = (-pow(p, -1, 1 << 32)) & M32
    env = {k: limbs(v) for k, v in ops.items()}
    if "a" in ops:  # operands of the squaring schedule
        two_a = limbs(2 * ops["a"])
        for k in range(2, 8):
            env["d%d" % k] = two_a[k]
        for j in range(7):
            env["f%d" % j] = (limbs(ops["a"])[j + 1] << 1) & M32
    m = {}
    lo, hi, r = 0, 0, [0] * 8

    def val(tok):
        tok = tok.strip()
        g = re.fullmatch(r"([abcd])\.l\[(\d)\]", tok)
        if g:
            return env[g.group(1)][int(g.group(2))]
        g = re.fullmatch(r"P::MOD\[(\d)\]", tok)
        if g:
            return mod[int(g.group(1))]
        g = re.fullmatch(r"P::NMOD\[(\d)\]", tok)
        if g:
            return limbs((1 << 256) - p)[int(g.group(1))]
        if tok in m:
            return m[tok]
        return env[tok]

    for ln in lines:
        ln = ln.strip()
        g = MAD.match(ln)
        if g:
            s = lo + val(g.group(2)) * val(g.group(3))
            carry, lo = s >> 64, s & M64
            assert carry <= 1
            if g.group(1) == "FREE":
                assert carry == 0, "a multiply-add emitted as carry-free carried: " + ln
            elif g.group(1) == "WRAP":
                pass                       # the top column of a result taken mod 2^256: its carries are not wanted
            elif g.group(1) == "SET":
                hi = carry
            else:
                hi = (hi + carry) & M32
            continue
        g = re.match(r"const uint32_t (m\d) = \(uint32_t\)lo \* P::INV;", ln)
        if g:
            m[g.group(1)] = ((lo & M32) * inv) & M32
            continue
        g = re.match(r"r\.l\[(\d)\] = \(uint32_t\)lo;", ln)
        if g:
            r[int(g.group(1))] = lo & M32
            continue
        g = re.match(r"const uint32_t (q\d) = \(uint32_t\)lo;", ln)
        if g:
            m[g.group(1)] = lo & M32
            continue
        if ln.startswith("H2_RESET"):
            lo, hi = 0, 0
            continue
        if ln.startswith("H2_SHIFT1"):
            lo = (lo >> 32) | (hi << 32)
            continue
        if ln.startswith("H2_SHIFT0"):
            lo >>= 32
            continue
        assert ln.startswith("//") or not ln, "unparsed schedule line: " + ln
    out = sum(x << (32 * i) for i, x in enumerate(r))
    if raw == "const":
        return out, sum(m["q%d" % i] << (32 * i) for i in range(8))
    assert out < 2 * p, "schedule result not below 2p"
    if raw:
        return out
    return out - p if out >= p else out


def run_lowered(lowered, p, ops, raw=False):
    """the same schedule after gen.lower (what fp_mul_gen.hpp holds: inline-assembly blocks whose carries are consumed two
    instructions behind their multiply-adds, out of three rotating carry registers): executed instruction by instruction,
    with the hazard distance and the life of every carry register checked"""
    mod = limbs(p)
    inv = (-pow(p, -1, 1 << 32)) & M32
    env = {k: limbs(v) for k, v in ops.items()}
    if "a" in ops:
        two_a = limbs(2 * ops["a"])
        for k in range(2, 8):
            env["d%d" % k] = two_a[k]
        for j in range(7):
            env["f%d" % j] = (limbs(ops["a"])[j + 1] << 1) & M32
    m = {}
    lo, hi, r = 0, 0, [0] * 8

    def val(tok):
        g = re.fullmatch(r"([abcd])\.l\[(\d)\]", tok)
        if g:
            return env[g.group(1)][int(g.group(2))]
        g = re.fullmatch(r"P::MOD\[(\d)\]", tok)
        if g:
            return mod[int(g.group(1))]
        g = re.fullmatch(r"P::NMOD\[(\d)\]", tok)
        if g:
            return limbs((1 << 256) - p)[int(g.group(1))]
        return m[tok] if tok in m else env[tok]

    for item in lowered:
        if item[0] == "asm":
            cy, written, clock = [None] * 3, [None] * 3, 0       # carry registers: value, clock of the write
            # (the top column of the constant-operand product: a bare multiply-add whose carry is not wanted -- its blocks
            #  consume no carry at all, which is what tells them apart here)
            wrap_ok = raw == "const" and not any(ins[0] in ("set", "acc") for ins in item[1])
            for ins in item[1]:
                if ins[0] == "mad":
                    assert wrap_ok or cy[ins[4]] is None or cy[ins[4]] == 0 or written[ins[4]] is None, "a pending carry was overwritten"
                    sm = lo + val(ins[1]) * val(ins[2])
                    cy[ins[4]], lo, written[ins[4]] = sm >> 64, sm & M64, clock
                    clock += 1
                elif ins[0] == "nop":
                    clock += ins[1]
                else:
                    assert clock - written[ins[1]] - 1 >= gen.WAIT, "carry read too early"
                    hi = cy[ins[1]] if ins[0] == "set" else (hi + cy[ins[1]]) & M32
                    cy[ins[1]], written[ins[1]] = None, None
                    clock += 1
            assert wrap_ok or all(c in (None, 0) for c in cy), "a carry was dropped at the end of a block"
            continue
        ln = item[1]
        g = re.match(r"const uint32_t (m\d) = \(uint32_t\)lo \* P::INV;", ln)
        if g:
            m[g.group(1)] = ((lo & M32) * inv) & M32
        elif re.match(r"r\.l\[(\d)\] = \(uint32_t\)lo;", ln):
            r[int(ln[4])] = lo & M32
        elif re.match(r"const uint32_t (q\d) = \(uint32_t\)lo;", ln):
            m[ln[15:17]] = lo & M32
        elif ln.startswith("lo = 0;"):
            lo, hi = 0, 0
        elif ln.startswith("lo = (lo >> 32) |"):
            lo = (lo >> 32) | (hi << 32)
        elif ln.startswith("lo >>= 32"):
            lo >>= 32
        else:
            assert ln.startswith("//") or not ln, "unparsed line: " + ln
    out = sum(x << (32 * i) for i, x in enumerate(r))
    if raw == "const":
        return out, sum(m["q%d" % i] << (32 * i) for i in range(8))
    if raw:
        return out
    return out - p if out >= p else out


def operands(p, rng, count):
    edge = [0, 1, p - 1, p, (1 << 254) - 1, (1 << 253), p - 2, (1 << 254) - (1 << 224)]
    for x in edge:
        for y in edge:
            yield x, y
    for _ in range(count):
        yield rng.randrange(1 << 254), rng.randrange(1 << 254)


@pytest.mark.parametrize("name", sorted(gen.FIELDS))
def test_product_and_square_schedules(name):
    p = gen.FIELDS[name]
    rinv = pow(1 << 256, -1, p)
    rng = random.Random(1)
    mul, _ = gen.schedule(p)
    sqr, _ = gen.schedule(p, square=True)
    for a, b in operands(p, rng, 300):
        assert run(mul, p, {"a": a, "b": b}) == a * b * rinv % p
        assert run(sqr, p, {"a": a}) == a * a * rinv % p


@pytest.mark.parametrize("name", sorted(gen.FIELDS))
def test_two_products_one_reduction_schedule(name):
    p = gen.FIELDS[name]
    rinv = pow(1 << 256, -1, p)
    rng = random.Random(2)
    dual, stats = gen.schedule(p, dual=True)
    assert stats["free"] + stats["set"] + stats["acc"] == 128 + 64 - 8 + 8  # 128 operand + 56 reduction + 8 m_i p_0 terms
    pairs = list(operands(p, rng, 200))
    for i, (a, b) in enumerate(pairs):
        c, d = pairs[(7 * i + 3) % len(pairs)]
        assert run(dual, p, {"a": a, "b": b, "c": c, "d": d}) == (a * b + c * d) * rinv % p


@pytest.mark.parametrize("name", sorted(gen.FIELDS))
def test_wide_operand_schedule_of_the_lazy_ntt_domain(name):
    """fp_mul_wide: the first operand is ANY 256-bit value (the NTT stage loops keep values below 4p), the second a
    canonical residue; the unreduced result is below 2p and congruent to a * b / 2^256"""
    p = gen.FIELDS[name]
    rinv = pow(1 << 256, -1, p)
    rng = random.Random(3)
    wide, stats = gen.schedule(p, wide=True)
    assert stats["free"] + stats["set"] + stats["acc"] == 128      # 64 operand + 56 reduction + 8 m_i p_0 terms
    firsts = [0, 1, p, 2 * p - 1, 2 * p, 4 * p - 1, (1 << 256) - 1, (1 << 255), 3 * p + 12345] + [rng.randrange(1 << 256) for _ in range(300)]
    seconds = [0, 1, p - 1, p - 2, (1 << 253), (p + 1) // 2] + [rng.randrange(p) for _ in range(40)]
    for i, a in enumerate(firsts):
        for b in (seconds if i < 9 else seconds[i % 7::7]):
            out = run(wide, p, {"a": a, "b": b}, raw=True)
            assert out < 2 * p and out % p == a * b * rinv % p, (hex(a), hex(b))


@pytest.mark.parametrize("name", sorted(gen.FIELDS))
def test_lowered_blocks_compute_what_the_schedules_do(name):
    """gen.lower: every variant's assembly blocks against the line-by-line schedule"""
    p = gen.FIELDS[name]
    rng = random.Random(4)
    variants = {"mul": {}, "square": {"square": True}, "dual": {"dual": True}, "wide": {"wide": True}}
    for vname, kw in variants.items():
        lines, _ = gen.schedule(p, **kw)
        lowered = gen.lower(lines)
        nops = sum(1 for it in lowered if it[0] == "asm" for ins in it[1] if ins[0] == "nop")
        mads = sum(1 for it in lowered if it[0] == "asm" for ins in it[1] if ins[0] == "mad")
        assert nops <= 40 and mads in (128, 100, 192), (vname, nops, mads)
        pairs = list(operands(p, rng, 60))
        for i, (a, b) in enumerate(pairs):
            if vname == "square":
                ops = {"a": a}
            elif vname == "dual":
                c, d = pairs[(5 * i + 1) % len(pairs)]
                ops = {"a": a, "b": b, "c": c, "d": d}
            elif vname == "wide":
                ops = {"a": (a << 2 | 3) & ((1 << 256) - 1), "b": b % p}
            else:
                ops = {"a": a, "b": b}
            assert run_lowered(lowered, p, ops, raw=True) == run(lines, p, ops, raw=True), (vname, i)


@pytest.mark.parametrize("name", sorted(gen.FIELDS))
def test_constant_operand_schedule_of_the_ntt_twiddles(name):
    """fp_mul_const: x * w for a tabulated constant w < p with its quotient w' = floor(w 2^256 / p) and ANY 256-bit x, 115
    multiply-adds: q from the anti-diagonals >= 6 of x w' is the exact quotient floor(x w' / 2^256) or one less, the result
    x w - q p is congruent to x w, below 2p (1 + 2^-30), and at or above 2p only when q came out short (then the top limb says
    so, which is the test fp_mul_const makes before its rare subtraction).  Logical schedule and lowered blocks alike."""
    p = gen.FIELDS[name]
    rng = random.Random(5)
    lines, stats = gen.schedule_const(p)
    assert stats["free"] + stats["set"] + stats["acc"] + stats["wrap"] == 43 + 36 + 36
    lowered = gen.lower(lines)
    top2p = (2 * p) >> 224
    xs = [0, 1, p, 2 * p - 1, 2 * p, 4 * p - 1, (1 << 256) - 1, (1 << 255), 3 * p + 12345, (1 << 224) - 1, 1 << 224] + [rng.randrange(1 << 256) for _ in range(300)]
    ws = [0, 1, 2, p - 1, p - 2, (p + 1) // 2, 1 << 253, (1 << 224) - 1, 1 << 224] + [rng.randrange(p) for _ in range(60)]
    short = 0
    for i, x in enumerate(xs):
        for w in (ws if i < 11 else ws[i % 7::7]):
            wq = (w << 256) // p
            out, q = run(lines, p, {"a": x, "b": w, "c": wq}, raw="const")
            assert run_lowered(lowered, p, {"a": x, "b": w, "c": wq}, raw="const") == (out, q)
            exact = (x * wq) >> 256
            assert q in (exact, exact - 1) and out == x * w - q * p, (hex(x), hex(w))
            assert out % p == x * w % p and out < 2 * p + (p >> 29)
            if out >= 2 * p:
                assert q == exact - 1 and (out >> 224) >= top2p
            short += q != exact
    # operands built so that x w' mod 2^256 is tiny -- the case in which the truncated quotient IS one short
    hits = 0
    for _ in range(200):
        w = rng.randrange(1, p)
        wq = (w << 256) // p
        if wq % 2 == 0:
            continue
        x = (rng.randrange(1 << 200) * pow(wq, -1, 1 << 256)) % (1 << 256)          # x w' = (something < 2^200) mod 2^256
        out, q = run(lines, p, {"a": x, "b": w, "c": wq}, raw="const")
        assert run_lowered(lowered, p, {"a": x, "b": w, "c": wq}, raw="const") == (out, q)
        exact = (x * wq) >> 256
        assert q in (exact, exact - 1) and out == x * w - q * p and out < 2 * p + (p >> 29)
        if q != exact:
            hits += 1
            fixed = out - 2 * p if (out >> 224) >= top2p and out >= 2 * p else out    # what fp_mul_const returns
            assert fixed < 2 * p and fixed % p == x * w % p
    assert hits > 20, "the short-quotient case was not exercised"
