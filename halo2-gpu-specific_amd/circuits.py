"""The benchmark circuit of BASELINE configs 1, 4 and 5: the mini-PLONK of examples/simple-example-2.rs:177-288
(commented out in the reference tree; the only create_proof-running circuit without selectors -- SURVEY.md 8(d))."""
import numpy as np

from .circuit import ConstraintSystem


def mini_plonk():
    """`configure` (:182-222): advice a, b, c with equality; fixed sm, sa, sb, sc; one degree-3 gate"""
    cs = ConstraintSystem("mini-plonk")
    a, b, c = cs.advice_column(), cs.advice_column(), cs.advice_column()
    for col in (a, b, c):
        cs.enable_equality(col)
    sm, sa, sb, sc = cs.fixed_column(), cs.fixed_column(), cs.fixed_column(), cs.fixed_column()
    qa, qb, qc = cs.query_advice(a), cs.query_advice(b), cs.query_advice(c)
    qsa, qsb, qsc, qsm = cs.query_fixed(sa), cs.query_fixed(sb), cs.query_fixed(sc), cs.query_fixed(sm)
    cs.create_gate("mini plonk", [qa * qsa + qb * qsb + qa * qb * qsm + (qc * qsc) * (-1)])
    return cs


def mini_plonk_synthesize(k, a=5, alloc=None):
    """`synthesize` (:224-254): 2^(k-4) times { raw_multiply(a, a, a^2); raw_add(a, a^2, a + a^2); copy a0 = a1;
    copy b1 = c0 }, one region per row.  Returns (advice[3], fixed[4], copies) with canonical (n, 4) u64 columns and
    copies as (m, 4) = (left column position, left row, right column position, right row).  `alloc(count, n)`
    supplies the advice columns (e.g. prover.Device.pinned_columns, so that the witness is DMA-able)."""
    n = 1 << k
    pairs = 1 << (k - 4)
    assert a * a + a < (1 << 64)
    adv = alloc(3, n) if alloc else [np.zeros((n, 4), dtype=np.uint64) for _ in range(3)]
    fixed = [np.zeros((n, 4), dtype=np.uint64) for _ in range(4)]
    r0 = np.arange(pairs) * 2
    r1 = r0 + 1
    a2 = a * a
    adv[0][r0, 0], adv[1][r0, 0], adv[2][r0, 0] = a, a, a2
    fixed[0][r0, 0] = 1           # sm
    fixed[3][r0, 0] = 1           # sc
    adv[0][r1, 0], adv[1][r1, 0], adv[2][r1, 0] = a, a2, a + a2
    fixed[1][r1, 0] = 1           # sa
    fixed[2][r1, 0] = 1           # sb
    fixed[3][r1, 0] = 1           # sc
    z, o, t = np.zeros(pairs, dtype=np.int64), np.ones(pairs, dtype=np.int64), np.full(pairs, 2, dtype=np.int64)
    copies = np.concatenate([np.stack([z, r0, z, r1], axis=1), np.stack([o, r1, t, r0], axis=1)])
    return adv, fixed, copies


def wide(quads=16):
    """A wide, lookup-bearing circuit in the shape of the zkWasm circuits this fork exists for (many advice columns,
    degree 5, range lookups; examples/range-check.rs:104-137 and examples/lookup_api.rs are its small relatives):
    4 * quads advice columns (a, b, c, d per quad), fixed q (row selector) and t (range table); one gate with a
    polynomial q * (a * b * c - d) per quad (degree 4); quads / 2 logup lookups into t, each with one input set of
    two columns (the a's of two neighbouring quads: degree 2 + 1 + 1 + 1 = 5); equality on the first two columns."""
    cs = ConstraintSystem("wide-%d" % quads)
    adv = [cs.advice_column() for _ in range(4 * quads)]
    q, t = cs.fixed_column(), cs.fixed_column()
    cs.enable_equality(adv[0])
    cs.enable_equality(adv[1])
    qq = cs.query_fixed(q)
    cells = [cs.query_advice(col) for col in adv]
    cs.create_gate("mul3", [qq * (cells[4 * i] * cells[4 * i + 1] * cells[4 * i + 2] + cells[4 * i + 3] * (-1))
                            for i in range(quads)])
    tt = cs.query_fixed(t)
    for l in range(quads // 2):
        cs.lookup_any("range%d" % l, [tt], [[[cells[8 * l]], [cells[8 * l + 4]]]])
    cs.set_minimum_degree(5)            # as benches/plonk.rs:181; what two inputs per lookup set need
    return cs


def wide_synthesize(k, quads=16, alloc=None, compact=False):
    """Witness of `wide`: t[i] = i for i < T = min(usable rows, 2^16); a, b, c < T from multiplicative hashes of (row,
    column), d = a b c on the usable rows; b of quad 0 is a of quad 0 one row up, tied by copy constraints on the first
    min(usable - 1, 2^16) rows.  Returns (advice[4 * quads], fixed[2], copies) in the layout of mini_plonk_synthesize.
    `compact`: the advice columns as 1-D u64 arrays (every value is below 2^48): prover.create_proof_ext uploads 8 bytes
    per cell; `alloc(count, n, compact=True)` then provides them."""
    n = 1 << k
    usable = n - 6
    T = min(usable, 1 << 16)
    ncols = 4 * quads
    if compact:
        adv = alloc(ncols, n, compact=True) if alloc else [np.zeros(n, dtype=np.uint64) for _ in range(ncols)]
    else:
        adv = alloc(ncols, n) if alloc else [np.zeros((n, 4), dtype=np.uint64) for _ in range(ncols)]
    cell = (lambda c: c[:usable]) if compact else (lambda c: c[:usable, 0])
    fixed = [np.zeros((n, 4), dtype=np.uint64) for _ in range(2)]
    fixed[0][:usable, 0] = 1
    fixed[1][:T, 0] = np.arange(T, dtype=np.uint64)
    rows = np.arange(usable, dtype=np.uint64)
    for qd in range(quads):
        vals = []
        for j in range(3):
            col = 4 * qd + j
            v = ((rows * np.uint64(2654435761) + np.uint64(40503 * col + 7)) >> np.uint64(5)) % np.uint64(T)
            vals.append(v)
        if qd == 0:
            vals[1] = np.concatenate([np.array([3 % T], dtype=np.uint64), vals[0][:-1]])
        for j in range(3):
            cell(adv[4 * qd + j])[:] = vals[j]
        cell(adv[4 * qd + 3])[:] = vals[0] * vals[1] * vals[2]             # < 2^48
    m = min(usable - 1, 1 << 16)
    r = np.arange(m, dtype=np.int64)
    copies = np.stack([np.zeros(m, dtype=np.int64), r, np.ones(m, dtype=np.int64), r + 1], axis=1)
    return adv, fixed, copies


def range_check(vmin=0, vmax=0xFFFF, step=2):
    """`configure` of examples/range-check.rs:38-60: fixed l_0, l_active, l_last_active and one `advice_column_range`
    over 0 ..= 65535 with step 2 (a second advice column, a degree-4 gate and a shuffle come with it)"""
    cs = ConstraintSystem("range-check-%d-%d-%d" % (vmin, vmax, step))
    l_0, l_active, l_last_active = cs.fixed_column(), cs.fixed_column(), cs.fixed_column()
    cs.advice_column_range(l_0, l_active, l_last_active, vmin, vmax, step)
    cs.chunk_shuffles()
    return cs


def range_check_synthesize(k, seed=0x52414E4745, alloc=None, vmin=0, vmax=0xFFFF, count=0xFFFF):
    """`synthesize` of examples/range-check.rs:62-93: l_0 = 1 on row 0, l_last_active = 1 on the last usable row,
    l_active = 1 on every usable row; 65535 random 16-bit values on the rows 0 .. 65534 of the range-checked column (a
    seeded generator instead of OsRng).  The `sort` column and the planted range values are the prover's job
    (prover.complete_range_check_witness, as in the reference's create_proof).  Returns (advice[2], fixed[3], copies)."""
    n = 1 << k
    usable = n - 6
    assert count < usable, "the witness does not fit 2^%d rows" % k
    adv = alloc(2, n) if alloc else [np.zeros((n, 4), dtype=np.uint64) for _ in range(2)]
    fixed = [np.zeros((n, 4), dtype=np.uint64) for _ in range(3)]
    fixed[0][0, 0] = 1
    fixed[1][:usable, 0] = 1
    fixed[2][usable - 1, 0] = 1
    rng = np.random.Generator(np.random.PCG64(seed))
    adv[0][:count, 0] = rng.integers(vmin, vmax + 1, size=count, dtype=np.uint64)
    return adv, fixed, np.zeros((0, 4), dtype=np.int64)


def lookup_api():
    """`configure` of examples/lookup_api.rs:53-103: three advice columns, fixed s_0, s_1 and a table column; one gate
    s_0 * (input_0 - input_1); three `lookup`s into the table column (input_0, 2 * input_1, input_2) and one `lookup_any`
    of (s_0 input_0, s_1 input_0) into (s_0 input_1, s_1 input_2) -- the traced front end: lookups into one table are
    collected and packed into input sets by the chunking pass (ConstraintSystem.chunk_lookups)"""
    cs = ConstraintSystem("lookup-api")
    i0, i1, i2 = cs.advice_column(), cs.advice_column(), cs.advice_column()
    s0, s1 = cs.fixed_column(), cs.fixed_column()
    table = cs.fixed_column()
    cs.create_gate("", [cs.query_fixed(s0) * (cs.query_advice(i0) * 1 - cs.query_advice(i1))])
    cs.lookup("table1", [(cs.query_advice(i0), cs.query_fixed(table))])
    cs.lookup("table2", [(cs.query_advice(i1) * 2, cs.query_fixed(table))])
    cs.lookup("table3", [(cs.query_advice(i2), cs.query_fixed(table))])
    q0, q1, q2 = cs.query_advice(i0), cs.query_advice(i1), cs.query_advice(i2)
    f0, f1 = cs.query_fixed(s0), cs.query_fixed(s1)
    cs.lookup("any", [(f0 * q0, f0 * q1), (f1 * q0, f1 * q2)])
    cs.chunk_lookups()
    return cs


def lookup_api_synthesize(k):
    """`synthesize` of examples/lookup_api.rs:121-160; the table column holds 0 .. 8 and its default (0) below"""
    n = 1 << k
    adv = [np.zeros((n, 4), dtype=np.uint64) for _ in range(3)]
    fixed = [np.zeros((n, 4), dtype=np.uint64) for _ in range(3)]
    adv[0][0, 0], adv[1][0, 0], fixed[0][0, 0] = 1, 1, 1
    adv[0][1, 0], adv[2][1, 0], fixed[1][1, 0] = 3, 3, 1
    fixed[2][:9, 0] = np.arange(9, dtype=np.uint64)
    return adv, fixed, np.zeros((0, 4), dtype=np.int64)


def shuffle_api_group():
    """`configure` of examples/shuffle_api_group.rs:53-113: five input and five shuffle advice columns, two + two fixed
    selectors; a gate s_in0 * (in0 - in1); four traced `shuffle`s (a two-column one, a plain one, one under one selector
    pair, one under two) that the chunking pass groups by degree (ConstraintSystem.chunk_shuffles)"""
    cs = ConstraintSystem("shuffle-api-group")
    ins = [cs.advice_column() for _ in range(5)]
    shs = [cs.advice_column() for _ in range(5)]
    s_in = [cs.fixed_column() for _ in range(2)]
    s_sh = [cs.fixed_column() for _ in range(2)]
    cs.create_gate("", [cs.query_fixed(s_in[0]) * (cs.query_advice(ins[0]) - cs.query_advice(ins[1]))])
    cs.shuffle("shuffle1", [(cs.query_advice(ins[0]), cs.query_advice(shs[0])), (cs.query_advice(ins[1]), cs.query_advice(shs[1]))])
    cs.shuffle("shuffle2", [(cs.query_advice(ins[2]), cs.query_advice(shs[2]))])
    cs.shuffle("shuffle3", [(cs.query_advice(ins[3]) * cs.query_fixed(s_in[0]), cs.query_advice(shs[3]) * cs.query_fixed(s_sh[0]))])
    cs.shuffle("shuffle4", [(cs.query_advice(ins[4]) * cs.query_fixed(s_in[0]) * cs.query_fixed(s_in[1]),
                             cs.query_advice(shs[4]) * cs.query_fixed(s_sh[0]) * cs.query_fixed(s_sh[1]))])
    cs.chunk_shuffles()
    return cs


def shuffle_api_group_synthesize(k, input0=(1, 2, 4, 1), input1=(4, 1, 1, 2)):
    """`synthesize` of examples/shuffle_api_group.rs:143-176: every input column holds `input0`, every shuffle column
    `input1` (a permutation of it), the selectors are 1 on those rows"""
    n = 1 << k
    adv = [np.zeros((n, 4), dtype=np.uint64) for _ in range(10)]
    fixed = [np.zeros((n, 4), dtype=np.uint64) for _ in range(4)]
    m = len(input0)
    for c in range(5):
        adv[c][:m, 0] = np.array(input0, dtype=np.uint64)
        adv[5 + c][:m, 0] = np.array(input1, dtype=np.uint64)
    for f in fixed:
        f[:m, 0] = 1
    return adv, fixed, np.zeros((0, 4), dtype=np.int64)


def lookup_api_set():
    """`configure` of examples/lookup_api_set.rs:52-103: six advice columns, fixed s_0, s_1 and a table column; the gate
    s_0 * (input_0 * 1 - input_1); six traced `lookup`s into the one table column (input_0, 2 input_1, input_2, 10 input_3,
    input_4, input_5).  The chunking pass packs them as the example's comments say: set 0 = {input_0} (it shares its
    polynomial with the table term), then {2 input_1, input_2}, {10 input_3, input_4}, {input_5} (degree 4)."""
    cs = ConstraintSystem("lookup-api-set")
    ins = [cs.advice_column() for _ in range(6)]
    s0, s1 = cs.fixed_column(), cs.fixed_column()
    table = cs.fixed_column()
    del s1                                                           # allocated, never queried (as in the example)
    cs.create_gate("", [cs.query_fixed(s0) * (cs.query_advice(ins[0]) * 1 - cs.query_advice(ins[1]))])
    for i, scale in enumerate((None, 2, None, 10, None, None)):
        q = cs.query_advice(ins[i])
        cs.lookup("table%d" % i, [(q * scale if scale else q, cs.query_fixed(table))])
    cs.chunk_lookups()
    return cs


def lookup_api_set_synthesize(k):
    """`synthesize` of examples/lookup_api_set.rs:121-170: every input column holds 1 on row 0 and 3 on row 1, s_0 = 1 on
    row 0, s_1 = 1 on row 1, the table column 0 .. 99"""
    n = 1 << k
    adv = [np.zeros((n, 4), dtype=np.uint64) for _ in range(6)]
    fixed = [np.zeros((n, 4), dtype=np.uint64) for _ in range(3)]
    for a in adv:
        a[0, 0], a[1, 0] = 1, 3
    fixed[0][0, 0], fixed[1][1, 0] = 1, 1
    fixed[2][:100, 0] = np.arange(100, dtype=np.uint64)
    return adv, fixed, np.zeros((0, 4), dtype=np.int64)


def shuffle_api():
    """`configure` of examples/shuffle_api.rs:47-84: advice input_0, input_1, shuffle_0, shuffle_1; fixed s_input,
    s_shuffle; the gate s_input * (10 input_0 - input_1) and ONE traced `shuffle` of two pairs
    (s_input input_j, s_shuffle shuffle_j) -- degree 2 + 2 = 4"""
    cs = ConstraintSystem("shuffle-api")
    in0, in1, sh0, sh1 = (cs.advice_column() for _ in range(4))
    s_in, s_sh = cs.fixed_column(), cs.fixed_column()
    cs.create_gate("", [cs.query_fixed(s_in) * (cs.query_advice(in0) * 10 - cs.query_advice(in1))])
    q_in0, q_sh0, q_in1, q_sh1 = cs.query_advice(in0), cs.query_advice(sh0), cs.query_advice(in1), cs.query_advice(sh1)
    f_in, f_sh = cs.query_fixed(s_in), cs.query_fixed(s_sh)
    cs.shuffle("shuffle", [(f_in * q_in0, f_sh * q_sh0), (f_in * q_in1, f_sh * q_sh1)])
    cs.chunk_shuffles()
    return cs


def shuffle_api_synthesize(k, input0=(1, 2, 4, 1), shuffle0=(4, 1, 1, 2)):
    """`synthesize` of examples/shuffle_api.rs:129-166: input_1 = 10 input_0 and shuffle_1 = 10 shuffle_0 row by row, the
    selectors 1 on those rows"""
    n = 1 << k
    adv = [np.zeros((n, 4), dtype=np.uint64) for _ in range(4)]
    fixed = [np.zeros((n, 4), dtype=np.uint64) for _ in range(2)]
    m = len(input0)
    adv[0][:m, 0] = np.array(input0, dtype=np.uint64)
    adv[1][:m, 0] = 10 * np.array(input0, dtype=np.uint64)
    adv[2][:m, 0] = np.array(shuffle0, dtype=np.uint64)
    adv[3][:m, 0] = 10 * np.array(shuffle0, dtype=np.uint64)
    fixed[0][:m, 0] = 1
    fixed[1][:m, 0] = 1
    return adv, fixed, np.zeros((0, 4), dtype=np.int64)


def shuffle_gates(width=4, theta=111, beta=222):
    """`MyConfig::configure` of examples/shuffle.rs:50-107: a shuffle argument written out as GATES -- fixed q_shuffle,
    q_first, q_last; advice original[W], shuffled[W] and a running product z with z(first) = z(last) = 1 and
    z(X) (compress(original) + beta) = z(wX) (compress(shuffled) + beta), theta and beta being CONSTANTS of the circuit"""
    from .circuit import Constant

    cs = ConstraintSystem("shuffle-gates-%d-%d-%d" % (width, theta, beta))
    q_shuffle, q_first, q_last = cs.fixed_column(), cs.fixed_column(), cs.fixed_column()
    original = [cs.advice_column() for _ in range(width)]
    shuffled = [cs.advice_column() for _ in range(width)]
    z = cs.advice_column()
    th, be = Constant(theta), Constant(beta)
    cs.create_gate("z should start with 1", [cs.query_fixed(q_first) * (Constant(1) - cs.query_advice(z))])
    cs.create_gate("z should end with 1", [cs.query_fixed(q_last) * (Constant(1) - cs.query_advice(z))])
    qs = cs.query_fixed(q_shuffle)
    orig = [cs.query_advice(c) for c in original]
    shuf = [cs.query_advice(c) for c in shuffled]
    z_cur, z_next = cs.query_advice(z), cs.query_advice(z, 1)

    def compress(cells):
        acc = cells[0]
        for cell in cells[1:]:
            acc = acc * th + cell
        return acc

    cs.create_gate("z should have valid transition", [qs * (z_cur * (compress(orig) + be) - z_next * (compress(shuf) + be))])
    return cs


def shuffle_gates_witness(k, width=4, height=32, theta=111, beta=222, seed=0x5348554646):
    """the witness of examples/shuffle.rs:150-238 as canonical integers: `width` columns of `height` random field
    elements (a seeded generator instead of OsRng), the same rows in another order (its Fisher-Yates walk), and the
    running product z[0] = 1, z[i + 1] = z[i] (compress(original_i) + beta) / (compress(shuffled_i) + beta).
    Returns (advice[2 W + 1], fixed[3]) as lists of n integers."""
    import random

    from .circuit import R_MOD

    n = 1 << k
    assert height + 1 <= n - 6
    rnd = random.Random(seed)
    original = [[rnd.randrange(R_MOD) for _ in range(height)] for _ in range(width)]
    order = list(range(height))
    for row in range(height - 1, 0, -1):
        other = rnd.getrandbits(32) % row
        order[row], order[other] = order[other], order[row]
    shuffled = [[col[i] for i in order] for col in original]

    def compress(cols, i):
        acc = 0
        for col in cols:
            acc = (acc * theta + col[i]) % R_MOD
        return acc

    z = [1]
    for i in range(height):
        z.append(z[-1] * (compress(original, i) + beta) % R_MOD * pow((compress(shuffled, i) + beta) % R_MOD, -1, R_MOD) % R_MOD)
    assert z[-1] == 1
    pad = lambda col: col + [0] * (n - len(col))  # noqa: E731
    adv = [pad(c) for c in original] + [pad(c) for c in shuffled] + [pad(z)]
    fixed = [pad([1] * height), pad([1]), pad([0] * height + [1])]           # q_shuffle, q_first, q_last
    return adv, fixed


def shuffle_gates_synthesize(k, width=4, height=32, theta=111, beta=222, seed=0x5348554646):
    """... as the prover's (n, 4) u64 columns.  Returns (advice, fixed, copies)."""
    adv, fixed = shuffle_gates_witness(k, width, height, theta, beta, seed)
    to_arr = lambda col: np.array([[(v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(4)] for v in col], dtype=np.uint64)  # noqa: E731
    return [to_arr(c) for c in adv], [to_arr(c) for c in fixed], np.zeros((0, 4), dtype=np.int64)
