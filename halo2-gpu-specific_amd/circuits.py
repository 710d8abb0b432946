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
