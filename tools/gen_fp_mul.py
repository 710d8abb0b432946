#!/usr/bin/env python3
"""tools/gen_fp_mul.py -- emits halo2-gpu-specific_amd/csrc/fp_mul_gen.hpp: the device Montgomery product of field.hpp
as a straight-line schedule per field.

A 32 x 32 -> 64 bit product added to a 64-bit accumulator can carry, and gfx950 has no multiply-add with a carry
*input*: the product-scanning multiplier therefore follows every v_mad_u64_u32 with a v_addc_co_u32 into a third
accumulator word (126 of them per product in round 1: ~27 % of the issue slots).  Most of those carries cannot happen:

  * a column starts from the small carry-over of the previous one (< 2^37);
  * a reduction term m_j * p_k is below 2^32 * p_k and the modulus limbs p_k are constants, several of them small;
  * inputs are field elements (< 2^254), so a_7 and b_7 are below 2^30 and the products that involve them below 2^62.

For each column this script sorts the terms by their exact upper bound and takes the longest prefix whose sum, carry-over
included, stays below 2^64: those terms are accumulated first with a bare v_mad_u64_u32 (`H2_MAD_FREE`); the first term
that can carry sets the third word from the carry bit (`H2_MAD_SET`, no zero initialisation), the rest are the usual pair
(`H2_MAD_ACC`).  The bounds are recomputed here with Python integers and asserted, column by column.

`schedule` returns that LOGICAL schedule as lines (`H2_MAD_FREE_V(a.l[0], b.l[1]);` ...: what tests/test_fp_mul_schedule.py
executes with Python integers); `lower` + `emit` turn it into the inline-assembly blocks the header holds (see `lower`).

Precondition of the generated code (same as the reference's `Fr` / `Fq` invariants): both inputs < 2^254.
"""
import os

FIELDS = {
    "FrParams": 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001,
    "FqParams": 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47,
}
W = (1 << 32) - 1
TOP = (1 << 30) - 1  # limb 7 of an input < 2^254


def schedule(p, square=False, dual=False, wide=False):
    """wide: the first operand may be ANY 256-bit value (the lazily reduced values of the NTT stage loops, below 4p), the
    second a canonical residue; the result (a b + m p) / 2^256 < (2^256 p + 2^256 p) / 2^256 = 2p is NOT reduced further.
    dual: a * b + c * d under ONE reduction (twice the operand products per column, the same 64 reduction products).
    square: a * a with every cross product a_j a_k (j < k) taken once against the DOUBLED operand -- row j multiplies
    a_j by the limbs of 2 * (a >> 32 (j + 1)) << 32 (j + 1): d_k = limb k of 2a for k >= j + 2 and f_j = d_(j+1) with
    bit 0 cleared (that bit is the top bit of a_j, which belongs to 2 a_j, not to the tail) -- 8 squares + 28 cross
    products instead of 64."""
    mod = [(p >> (32 * i)) & W for i in range(8)]
    amax = [W] * 7 + [TOP]
    awide = [W] * 8 if wide else amax  # limbs of the first operand
    dmax = [W] * 7 + [2 * TOP + 1]  # limbs of 2a: 2a < 2^255
    lines, stats = [], {"free": 0, "set": 0, "acc": 0}
    carry_in = 0  # exact upper bound of the accumulator at the start of the column
    for i in range(16):
        terms = []  # (bound, code x, code y)
        for j in range(max(0, i - 7), min(i, 7) + 1):
            k = i - j
            if not square:
                terms.append((awide[j] * amax[k], "a.l[%d]" % j, "b.l[%d]" % k, "v"))
                if dual:
                    terms.append((amax[j] * amax[k], "c.l[%d]" % j, "d.l[%d]" % k, "v"))
            elif k == j:
                terms.append((amax[j] * amax[j], "a.l[%d]" % j, "a.l[%d]" % j, "v"))
            elif k == j + 1:
                terms.append((amax[j] * (dmax[k] & ~1), "a.l[%d]" % j, "f%d" % j, "v"))
            elif k > j + 1:
                terms.append((amax[j] * dmax[k], "a.l[%d]" % j, "d%d" % k, "v"))
        jm_hi = min(i - 1, 7) if i <= 7 else 7
        for j in range(max(0, i - 7), jm_hi + 1):
            if i - j >= 1:
                terms.append((W * mod[i - j], "m%d" % j, "P::MOD[%d]" % (i - j), "s"))
        terms.sort(key=lambda t: t[0])
        total, nfree = carry_in, 0
        for t in terms:
            if total + t[0] < (1 << 64):
                total += t[0]
                nfree += 1
            else:
                break
        lines.append("    // column %d: %d terms, %d cannot carry" % (i, len(terms) + (1 if i <= 7 else 0), nfree))
        column_max = carry_in + sum(t[0] for t in terms)
        carried = False
        for k, (_, x, y, cls) in enumerate(terms):
            kind = "FREE" if k < nfree else ("ACC" if carried else "SET")
            if kind == "SET":
                carried = True
            stats[kind.lower()] += 1
            lines.append("    H2_MAD_%s_%s(%s, %s);" % (kind, cls.upper(), x, y))
        if i <= 7:
            lines.append("    const uint32_t m%d = (uint32_t)lo * P::INV;" % i)
            kind = "ACC" if carried else "SET"
            carried = True
            stats[kind.lower()] += 1
            lines.append("    H2_MAD_%s_S(m%d, P::MOD[0]);  // low word becomes 0" % (kind, i))
            column_max += W * mod[0]
        else:
            lines.append("    r.l[%d] = (uint32_t)lo;" % (i - 8))
        if i < 15:
            lines.append("    H2_SHIFT%d();" % (1 if carried else 0))
        assert column_max < (1 << 96)
        carry_in = column_max >> 32
        assert carry_in < (1 << 40)
    # a, b < p  =>  result < 2p: the word above r.l[7] is zero
    return lines, stats


def schedule_const(p):
    """The product by a CONSTANT (a twiddle factor of the NTT): next to the plain value w < p the table holds its quotient
    w' = floor(w * 2^256 / p), and for ANY 256-bit x

        q = floor(x w' / 2^256)   -- only the anti-diagonals i + j >= 6 of x w' are summed (43 of 64 products): the dropped
                                     ones are below 7 * 2^224, so q is the exact quotient or one less
        r = (x w + q (2^256 - p)) mod 2^256 = x w - q p   -- the low halves of two products (36 + 36; the top column's carries
                                     fall off the end)

    115 multiply-adds and no `m = t n'` steps, against 64 + 64 + 8 for the Montgomery form.  x w - q_exact p =
    (f p + x e) / 2^256 with f = x w' mod 2^256 and e = w 2^256 mod p, below 2p; the truncated q is one short only when
    f < 7 * 2^224, and then r = (f p + x e) / 2^256 + p < 2p + 7 * 2^-32 p.  So r < 2p (1 + 2^-30) always and r >= 2p almost
    never: the caller tests the top limb and subtracts 2p in that case (field.hpp fp_mul_const).  The data stay in Montgomery
    form (x = X R): x w = (X w) R for a PLAIN w, nothing to convert.  Operands: a = x, b = w, c = w'."""
    nmod = [(((1 << 256) - p) >> (32 * i)) & W for i in range(8)]
    wmax = [W] * 7 + [(p - 1) >> 224]
    lines, stats = [], {"free": 0, "set": 0, "acc": 0, "wrap": 0}

    def column(i, terms, carry_in, last):
        """-> (exact upper bound of the carry-over, whether the third word was written)"""
        terms = sorted(terms, key=lambda t: t[0])
        total, nfree = carry_in, 0
        for t in terms:
            if total + t[0] < (1 << 64):
                total += t[0]
                nfree += 1
            else:
                break
        lines.append("    // column %d: %d terms, %d cannot carry%s" % (i, len(terms), nfree, " (carries out of the top column are not wanted)" if last else ""))
        carried = False
        for k, (_, x, y, cls) in enumerate(terms):
            kind = "WRAP" if (last and k >= nfree) else ("FREE" if k < nfree else ("ACC" if carried else "SET"))
            if kind == "SET":
                carried = True
            stats[kind.lower()] += 1
            lines.append("    H2_MAD_%s_%s(%s, %s);" % (kind, cls, x, y))
        column_max = carry_in + sum(t[0] for t in terms)
        assert column_max < (1 << 96)
        return column_max >> 32, carried

    # ---- q: anti-diagonals 6 .. 14 of a * c; limbs 8 .. 15 of the sum are q0 .. q7
    lines.append("    // ---- q = floor(a c / 2^256), anti-diagonals 6..14 only (exact or one short)")
    carry_in = 0
    for i in range(6, 15):
        terms = [(W * W, "a.l[%d]" % j, "c.l[%d]" % (i - j), "V") for j in range(max(0, i - 7), min(i, 7) + 1)]
        carry_in, carried = column(i, terms, carry_in, False)
        if i >= 8:
            lines.append("    const uint32_t q%d = (uint32_t)lo;" % (i - 8))
        lines.append("    H2_SHIFT%d();" % (1 if carried else 0))
    assert carry_in < (1 << 32)          # limb 15 of the product
    lines.append("    const uint32_t q7 = (uint32_t)lo;")
    lines.append("    H2_RESET();")
    # ---- r: columns 0 .. 7 of a * b + q * (2^256 - p)
    lines.append("    // ---- r = (a b + q (2^256 - p)) mod 2^256")
    carry_in = 0
    for i in range(8):
        terms = []
        for j in range(i + 1):
            terms.append((W * wmax[i - j], "a.l[%d]" % j, "b.l[%d]" % (i - j), "V"))
            terms.append((W * nmod[i - j], "q%d" % j, "P::NMOD[%d]" % (i - j), "S"))
        carry_in, carried = column(i, terms, carry_in, i == 7)
        lines.append("    r.l[%d] = (uint32_t)lo;" % i)
        if i < 7:
            lines.append("    H2_SHIFT%d();" % (1 if carried else 0))
    return lines, stats


# ---------------------------------------------------------------------------------------------------------------
# Lowering of a logical schedule (the macro lines above, which tests/test_fp_mul_schedule.py executes with Python integers)
# to inline-assembly BLOCKS.  One asm statement per instruction made the compiler put an `s_nop` behind almost every
# multiply-add (it cannot see inside an asm statement and pads each one against the gfx940 "VALU writes an SGPR -> VALU
# reads it" hazard): a third of all instructions of the arithmetic kernels were `s_nop`s -- invisible at four waves per SIMD,
# where another wave issues meanwhile, but a wave that runs alone (the chains of dependent point additions in k_reduce /
# k_finish, small transforms) issues one instruction per ~4.6 cycles whatever it is.  Inside a block the schedule keeps the
# hazard distance itself: the carry of multiply-add k (an SGPR pair, three of them in rotation) is consumed -- v_cndmask for
# SET, v_addc_co for ACC -- after multiply-add k + 2, i.e. WAIT instructions later; only the tail of a block needs `s_nop`s.
# ---------------------------------------------------------------------------------------------------------------
WAIT = 2            # wait states between a VALU that writes an SGPR and a VALU that reads it (LLVM: VALUWriteSGPRVALURead)
MAX_OPERANDS = 30   # of one asm statement (clang's limit)
import re as _re

_MAD = _re.compile(r"H2_MAD_(FREE|SET|ACC|WRAP)_([VS])\(([^,]+), ([^)]+)\);")


def lower(lines):
    """-> list of ("c", text) / ("asm", [instr]) with instr = ("mad", x, y, ycls, reg) / ("set", reg) / ("acc", reg) /
    ("nop", wait states)"""
    out, run = [], []

    def flush():
        """the multiply-adds collected so far become asm blocks of at most MAX_OPERANDS operands"""
        nonlocal run
        while run:
            ops, take = set(), 0
            for kind, cls, x, y in run:
                new = ops | {x, y}
                if len(new) + 5 > MAX_OPERANDS:          # + lo, hi, three carry registers
                    break
                ops, take = new, take + 1
            assert take > 0
            out.append(("asm", _schedule_block(run[:take])))
            run = run[take:]

    for ln in lines:
        t = ln.strip()
        g = _MAD.match(t)
        if g:
            run.append((g.group(1), g.group(2), g.group(3).strip(), g.group(4).strip()))
            continue
        flush()
        if t.startswith("//") or not t:
            out.append(("c", t))
        elif t.startswith("H2_SHIFT1"):
            out.append(("c", "lo = (lo >> 32) | ((uint64_t)hi << 32);"))
        elif t.startswith("H2_SHIFT0"):
            out.append(("c", "lo >>= 32;"))
        elif t.startswith("H2_RESET"):
            out.append(("c", "lo = 0;"))
        else:
            out.append(("c", t))
    flush()
    return out


def _schedule_block(terms):
    instrs, pending = [], []          # pending: (index of the producing instruction, kind, reg)
    for k, (kind, cls, x, y) in enumerate(terms):
        reg = k % 3
        instrs.append(("mad", x, y, cls, reg))
        if kind not in ("FREE", "WRAP"):
            pending.append((len(instrs) - 1, kind, reg))
        # consume what is WAIT instructions old (and must go before its register comes round again)
        while pending and len(instrs) - 1 - pending[0][0] >= WAIT:
            _, kd, rg = pending.pop(0)
            instrs.append(("set" if kd == "SET" else "acc", rg))
    while pending:
        at, kd, rg = pending.pop(0)
        gap = len(instrs) - 1 - at
        if gap < WAIT:
            instrs.append(("nop", WAIT - gap))
        instrs.append(("set" if kd == "SET" else "acc", rg))
    # a register is never rewritten while its carry is pending, and every carry is consumed WAIT or more instructions
    # (or wait states of s_nop) after its multiply-add
    live, nmad, clock = {}, 0, 0
    for ins in instrs:
        if ins[0] == "mad":
            assert ins[4] not in live, "carry register reused before its carry was consumed"
            if terms[nmad][0] not in ("FREE", "WRAP"):
                live[ins[4]] = clock
            nmad += 1
            clock += 1
        elif ins[0] == "nop":
            clock += ins[1]
        else:
            assert clock - live.pop(ins[1]) - 1 >= WAIT, "carry consumed too early"
            clock += 1
    assert not live and nmad == len(terms)
    return instrs


def emit(lowered, indent="    "):
    """C++ text of a lowered schedule"""
    text = []
    for item in lowered:
        if item[0] == "c":
            text.append(indent + item[1] if item[1] else "")
            continue
        names, inputs, body = {}, [], []

        def operand(expr, cls):
            key = (expr, cls)
            if key not in names:
                names[key] = "x%d" % len(names)
                inputs.append('[%s] "%s"(%s)' % (names[key], "s" if cls == "S" else "v", expr))
            return "%%[%s]" % names[key]

        for ins in item[1]:
            if ins[0] == "mad":
                body.append("v_mad_u64_u32 %%[lo], %%[cy%d], %s, %s, %%[lo]" % (ins[4], operand(ins[1], "V"), operand(ins[2], ins[3])))
            elif ins[0] == "set":
                body.append("v_cndmask_b32 %%[hi], 0, 1, %%[cy%d]" % ins[1])
            elif ins[0] == "acc":
                body.append("v_addc_co_u32 %%[hi], %%[cy%d], %%[hi], 0, %%[cy%d]" % (ins[1], ins[1]))
            else:
                body.append("s_nop %d" % (ins[1] - 1))
        assert len(inputs) + 5 <= MAX_OPERANDS
        # the third word: not an operand of a block without carries; write-only (early clobber: it is written while inputs
        # are still to be read) when the block's first carry SETS it -- the register allocator can then put it where the
        # upper half of the NEXT column's accumulator will be, which saves a move per column; read-write otherwise
        carries = [ins[0] for ins in item[1] if ins[0] in ("set", "acc")]
        outs = ['[lo] "+v"(lo)']
        if carries:
            outs.append('[hi] "=&v"(hi)' if carries[0] == "set" else '[hi] "+v"(hi)')
        used = sorted({ins[4] for ins in item[1] if ins[0] == "mad"})
        outs += ['[cy%d] "=&s"(cy%d)' % (r, r) for r in used]
        text.append(indent + "asm(" + ("\n" + indent + "    ").join('"%s\\n\\t"' % b for b in body))
        text.append(indent + "    : " + ", ".join(outs))
        text.append(indent + "    : " + ", ".join(inputs) + ");")
    return text


def main():
    out = ["// fp_mul_gen.hpp -- GENERATED by tools/gen_fp_mul.py (do not edit): carry-aware product-scanning schedules of the",
           "// device Montgomery product for BN254 Fr and Fq.  Included by field.hpp inside namespace h2.",
           "// A multiply-add whose column sum provably fits 64 bits so far is bare; the first term of a column that can carry sets the",
           "// third word from its carry (v_cndmask), the later ones add theirs (v_addc_co).  One asm statement per run of",
           "// multiply-adds: the carries sit in three rotating SGPR pairs and are consumed two instructions behind their producers",
           "// (the gfx940 'VALU writes an SGPR -> VALU reads it' distance), so the compiler has nothing to pad (see `lower`).", ""]
    for name, p in FIELDS.items():
        lines, stats = schedule(p)
        out.append("// %s: %d bare multiply-adds, %d carry-setting, %d carry-accumulating (round 1: 2 + 126)"
                   % (name, stats["free"], stats["set"], stats["acc"]))
        out.append("template <>")
        out.append("__device__ __forceinline__ Fp<%s> fp_mul_dev<%s>(const Fp<%s>& a, const Fp<%s>& b) {" % ((name,) * 4))
        out.append("    using P = %s;" % name)
        out.append("    Fp<P> r;")
        out.append("    uint64_t lo = 0, cy0, cy1, cy2;")
        out.append("    uint32_t hi = 0;")
        out += emit(lower(lines))
        out.append("    (void)cy0; (void)cy1; (void)cy2; (void)hi;")
        out.append("    return fp_reduce_once(r);")
        out.append("}")
        out.append("")
        print(name, stats)
        # the square: 36 operand products instead of 64 (see schedule)
        lines, stats = schedule(p, square=True)
        out.append("// %s square: %d bare multiply-adds, %d carry-setting, %d carry-accumulating" % (name, stats["free"], stats["set"], stats["acc"]))
        out.append("template <>")
        out.append("__device__ __forceinline__ Fp<%s> fp_sqr_dev<%s>(const Fp<%s>& a) {" % ((name,) * 3))
        out.append("    using P = %s;" % name)
        out.append("    Fp<P> r;")
        out.append("    uint64_t lo = 0, cy0, cy1, cy2;")
        out.append("    uint32_t hi = 0;")
        out.append("    // limbs of 2a (a < 2^254) and, per row, the first tail limb with the bit that belongs to 2 a_j cleared")
        for k in range(2, 8):
            out.append("    const uint32_t d%d = __builtin_amdgcn_alignbit(a.l[%d], a.l[%d], 31);" % (k, k, k - 1))
        for j in range(7):
            out.append("    const uint32_t f%d = a.l[%d] << 1;" % (j, j + 1))
        out += emit(lower(lines))
        out.append("    (void)cy0; (void)cy1; (void)cy2; (void)hi;")
        out.append("    return fp_reduce_once(r);")
        out.append("}")
        out.append("")
        print(name, "square", stats)
        # a * b + c * d under one reduction: 128 operand products + 64 reduction products instead of 2 x (64 + 64)
        lines, stats = schedule(p, dual=True)
        out.append("// %s a * b + c * d: %d bare multiply-adds, %d carry-setting, %d carry-accumulating" % (name, stats["free"], stats["set"], stats["acc"]))
        out.append("template <>")
        out.append("__device__ __forceinline__ Fp<%s> fp_mul2_dev<%s>(const Fp<%s>& a, const Fp<%s>& b, const Fp<%s>& c, const Fp<%s>& d) {" % ((name,) * 6))
        out.append("    using P = %s;" % name)
        out.append("    Fp<P> r;")
        out.append("    uint64_t lo = 0, cy0, cy1, cy2;")
        out.append("    uint32_t hi = 0;")
        out += emit(lower(lines))
        out.append("    (void)cy0; (void)cy1; (void)cy2; (void)hi;")
        out.append("    return fp_reduce_once(r);  // inputs < 2^254: (a b + c d) / 2^256 + p < 2^253 + p < 2 p")
        out.append("}")
        out.append("")
        print(name, "dual", stats)
        # (any 256-bit value) * (canonical residue) -> a value below 2p, no final subtraction: the NTT stage loops
        lines, stats = schedule(p, wide=True)
        out.append("// %s wide: first operand < 2^256, second < 2^254; result < 2p, NOT reduced: %d bare multiply-adds, %d carry-setting, %d carry-accumulating"
                   % (name, stats["free"], stats["set"], stats["acc"]))
        out.append("template <>")
        out.append("__device__ __forceinline__ Fp<%s> fp_mul_wide_dev<%s>(const Fp<%s>& a, const Fp<%s>& b) {" % ((name,) * 4))
        out.append("    using P = %s;" % name)
        out.append("    Fp<P> r;")
        out.append("    uint64_t lo = 0, cy0, cy1, cy2;")
        out.append("    uint32_t hi = 0;")
        out += emit(lower(lines))
        out.append("    (void)cy0; (void)cy1; (void)cy2; (void)hi;")
        out.append("    return r;  // (a b + m p) / 2^256 < 2p: the word above r.l[7] is zero")
        out.append("}")
        out.append("")
        print(name, "wide", stats)
        # (any 256-bit value) * (a tabulated constant with its quotient) -> below 2p (1 + 2^-30): the NTT's twiddle products
        lines, stats = schedule_const(p)
        out.append("// %s by a constant: a < 2^256, b = w < p plain, c = floor(w 2^256 / p); result = a w - q p < 2p (1 + 2^-30), NOT reduced:"
                   % name)
        out.append("// %d bare multiply-adds, %d carry-setting, %d carry-accumulating, %d whose carry falls off the top column"
                   % (stats["free"], stats["set"], stats["acc"], stats["wrap"]))
        out.append("template <>")
        out.append("__device__ __forceinline__ Fp<%s> fp_mul_const_dev<%s>(const Fp<%s>& a, const Fp<%s>& b, const Fp<%s>& c) {" % ((name,) * 5))
        out.append("    using P = %s;" % name)
        out.append("    Fp<P> r;")
        out.append("    uint64_t lo = 0, cy0, cy1, cy2;")
        out.append("    uint32_t hi = 0;")
        out += emit(lower(lines))
        out.append("    (void)cy0; (void)cy1; (void)cy2; (void)hi;")
        out.append("    return r;")
        out.append("}")
        out.append("")
        print(name, "const", stats)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "halo2-gpu-specific_amd", "csrc", "fp_mul_gen.hpp")
    with open(path, "w") as f:
        f.write("\n".join(out))


if __name__ == "__main__":
    main()
