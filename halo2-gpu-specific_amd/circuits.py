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


def wide_synthesize(k, quads=16, alloc=None):
    """Witness of `wide`: t[i] = i for i < T = min(usable rows, 2^16); a, b, c < T from multiplicative hashes of (row,
    column), d = a b c on the usable rows; b of quad 0 is a of quad 0 one row up, tied by copy constraints on the first
    min(usable - 1, 2^16) rows.  Returns (advice[4 * quads], fixed[2], copies) in the layout of mini_plonk_synthesize."""
    n = 1 << k
    usable = n - 6
    T = min(usable, 1 << 16)
    ncols = 4 * quads
    adv = alloc(ncols, n) if alloc else [np.zeros((n, 4), dtype=np.uint64) for _ in range(ncols)]
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
            adv[4 * qd + j][:usable, 0] = vals[j]
        adv[4 * qd + 3][:usable, 0] = vals[0] * vals[1] * vals[2]          # < 2^48
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
