"""Random circuits through the whole prover: keygen + create_proof on the device (generated gate kernels) against the CPU
prover over the C oracle (tests/oracle_prover.py: the oracle's evaluator interpreter and loops), same SRS, witness and
randomness -- the proof bytes must be equal.

A circuit is drawn from a seed: 2-6 advice / 1-3 fixed / 0-1 instance columns, 1-4 gates of random expressions (queries
at rotations -2 .. 2, constants, sums, products, scalings; degree <= 5), random equality columns and copy constraints,
optionally a logup lookup (1-2 input sets into a fixed table) and a shuffle.  The witness is random field elements:
a proof is a deterministic function of (key, witness, randomness) whether or not the gates hold -- the quotient phase
divides pointwise on the extended domain either way -- so unsatisfied gates and copies exercise the same arithmetic;
only what the prover itself checks is made true (lookup inputs come from the table, the shuffle is a permutation).

usage: [H2_FUZZ_KMAX=14] [H2_FUZZ_HOST_API=1] python tools/prover_fuzz.py [seconds] [first seed] [satisfiable]
(H2_FUZZ_HOST_API=1: every circuit is also proved through the host-slice entry points, halo2-gpu-specific_amd/host_api.py)
"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

from halo2_gpu_specific_amd import circuit as hc  # noqa: E402
from halo2_gpu_specific_amd.transcript import R_MOD  # noqa: E402

S_TRAPDOOR = 0x1D0C5F0A3B7E91C2A4D6F8091B2C3D4E5F60718293A4B5C6D7E8F9010203


def _arr(vals):
    a = np.zeros((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        for limb in range(4):
            a[i, limb] = (v >> (64 * limb)) & 0xFFFFFFFFFFFFFFFF
    return a


def _evaluate(e, row, n, cols):
    """an Expression at one row over canonical integer columns {kind: [column][row]}"""
    if isinstance(e, hc.Constant):
        return e.v
    if isinstance(e, hc.Query):
        return cols[e.name][e.column][(row + e.rotation) % n]
    if isinstance(e, hc.Negated):
        return -_evaluate(e.e, row, n, cols) % R_MOD
    if isinstance(e, hc.Sum):
        return (_evaluate(e.a, row, n, cols) + _evaluate(e.b, row, n, cols)) % R_MOD
    if isinstance(e, hc.Product):
        return _evaluate(e.a, row, n, cols) * _evaluate(e.b, row, n, cols) % R_MOD
    return _evaluate(e.e, row, n, cols) * e.c % R_MOD          # Scaled


def random_case(seed, satisfiable=False, k=None, witness_seed=None):
    """-> (cs, k, advice, fixed, copies, instances): see the module docstring.  `satisfiable`: every gate reads
    sel (E - d) with a selector that is on where all of E's rotations stay inside the usable rows and d a witness column
    of its own set to E, and the copy constraints hold (cells of a cycle share one value): the quotient is then a
    polynomial, and the coset-by-coset routes of the extended-domain phase must give the bytes of the extended route."""
    rnd = random.Random(seed)
    cs = hc.ConstraintSystem("fuzz-%d" % seed)
    n_adv, n_fix, n_inst = rnd.randint(2, 6), rnd.randint(1, 3), rnd.randint(0, 1)
    adv = [cs.advice_column() for _ in range(n_adv)]
    fix = [cs.fixed_column() for _ in range(n_fix)]
    inst = [cs.instance_column() for _ in range(n_inst)]
    with_lookup, with_shuffle = rnd.random() < 0.5, rnd.random() < 0.5
    table_col = cs.fixed_column() if with_lookup else None
    # (satisfiable circuits: two inputs in the first set, so that the lookup constraint really has the degree 5 it declares)
    look_cols = [cs.advice_column() for _ in range(rnd.randint(2 if satisfiable else 1, 3))] if with_lookup else []
    shuf_cols = [cs.advice_column(), cs.advice_column()] if with_shuffle else []

    def leaf():
        kind = rnd.random()
        if kind < 0.15:
            return hc.Constant(rnd.randrange(R_MOD)) if rnd.random() < 0.5 else hc.Constant(rnd.randrange(5))
        if kind < 0.70 or not (fix or inst):
            return cs.query_advice(rnd.choice(adv), rnd.randint(-2, 2))
        if kind < 0.92 or not inst:
            return cs.query_fixed(rnd.choice(fix), rnd.choice((0, 0, 1, -1)))
        return cs.query_instance(rnd.choice(inst), rnd.choice((0, 1)))

    def expr(depth, max_degree):
        if depth == 0 or max_degree <= 1 or rnd.random() < 0.25:
            return leaf()
        op = rnd.random()
        if op < 0.35:
            return expr(depth - 1, max_degree) + expr(depth - 1, max_degree)
        if op < 0.55:
            return expr(depth - 1, max_degree) - expr(depth - 1, max_degree)
        if op < 0.90:
            left = max(1, max_degree // 2)
            return expr(depth - 1, left) * expr(depth - 1, max_degree - left)
        return expr(depth - 1, max_degree) * rnd.randrange(1, R_MOD)

    defined = []                                   # satisfiable: (witness column of its own, the expression it equals)
    sel_col = cs.fixed_column() if satisfiable else None
    for g in range(rnd.randint(1, 4)):
        polys = []
        for _ in range(rnd.randint(1, 2)):
            if satisfiable:
                d_col, e = cs.advice_column(), expr(3, 4)
                defined.append((d_col, e))
                polys.append(cs.query_fixed(sel_col) * (e - cs.query_advice(d_col)))
            else:
                polys.append(cs.query_fixed(rnd.choice(fix)) * expr(3, 4))
        cs.create_gate("g%d" % g, polys)
    for col in (rnd.sample(adv, rnd.randint(1, len(adv))) + ([rnd.choice(fix)] if rnd.random() < 0.5 else []) +
                ([] if satisfiable else inst)):
        cs.enable_equality(col)
    if with_lookup:
        tq = cs.query_fixed(table_col)
        first = [[cs.query_advice(c)] for c in look_cols[:2]]
        sets = [first] + ([[[cs.query_advice(look_cols[2])]]] if len(look_cols) > 2 else [])
        cs.lookup_any("lk", [tq], sets)
    if with_shuffle:
        cs.shuffle_group([("sh", [cs.query_advice(shuf_cols[0])], [cs.query_advice(shuf_cols[1])])])
    # (a lookup set of two inputs needs degree 5: its table + inputs + 2)
    if satisfiable:        # no higher than what the constraints reach, or the top piece of h(X) is zero (see run_case)
        cs.set_minimum_degree(5 if with_lookup else 3)
    else:
        cs.set_minimum_degree(rnd.choice((5, 6)) if with_lookup else rnd.choice((3, 4, 5, 6)))
    k_draw = rnd.randint(5, 9)                          # (always drawn: the stream must not depend on the arguments)
    if os.environ.get("H2_FUZZ_KMAX"):                  # larger circuits: multi-workgroup scans, two-pass transforms
        k_draw = 5 + (k_draw - 5 + seed) % (int(os.environ["H2_FUZZ_KMAX"]) - 4)
    k = k_draw if k is None else k                      # `k` given: the caller's size, raised if the circuit needs more
    while (1 << k) < cs.minimum_rows() + 8:
        k += 1
    n = 1 << k
    usable = n - (cs.blinding_factors() + 1)
    # circuit-level data first (fixed columns, the lookup table, the copies), from the circuit's stream; then the witness --
    # with `witness_seed` another witness (advice, instances) of the SAME circuit: the circuits of one proof over several
    # circuit instances
    rf = lambda r=rnd: r.randrange(R_MOD) if r.random() < 0.8 else r.randrange(4)  # noqa: E731
    fixed = [[rnd.randrange(2) if rnd.random() < 0.5 else rf() for _ in range(n)] for _ in range(cs.num_fixed)]
    if with_lookup:
        t_idx = table_col[1]
        distinct = [rnd.randrange(R_MOD) for _ in range(rnd.randint(1, 40))]
        fixed[t_idx] = [rnd.choice(distinct) for _ in range(n)]
    ncols = len(cs.perm_columns)
    copies = [(rnd.randrange(ncols), rnd.randrange(usable), rnd.randrange(ncols), rnd.randrange(usable))
              for _ in range(rnd.randint(0, 12))]
    wr = rnd if witness_seed is None else random.Random((seed << 20) ^ witness_seed)
    advice = [[rf(wr) for _ in range(n)] for _ in range(cs.num_advice)]
    if with_lookup:
        for c in look_cols:
            advice[c[1]] = [fixed[t_idx][wr.randrange(usable)] for _ in range(n)]
    if with_shuffle:
        src = advice[shuf_cols[0][1]]
        perm = list(range(usable))
        wr.shuffle(perm)
        advice[shuf_cols[1][1]] = [src[perm[i]] if i < usable else 0 for i in range(n)]
    # (at least one non-zero public input: the commitment of an all-zero column is the identity, which the transcript
    # refuses as the reference's does)
    instances = [[wr.randrange(1, R_MOD)] + [rf(wr) for _ in range(wr.randint(0, min(usable, 5) - 1))] for _ in range(n_inst)]
    if satisfiable:
        cells = {"advice": advice, "fixed": fixed}
        parent = {}

        def find(c):
            while parent.setdefault(c, c) != c:
                parent[c] = parent[parent[c]]
                c = parent[c]
            return c

        for c1, r1, c2, r2 in copies:                       # the cells of a cycle take the value of its representative
            parent[find((c2, r2))] = find((c1, r1))
        for c, r in list(parent):
            (k0, i0), (rc, rr) = cs.perm_columns[c], find((c, r))
            k1, i1 = cs.perm_columns[rc]
            cells[k0][i0][r] = cells[k1][i1][rr]
        fixed[sel_col[1]] = [1 if 2 <= r < usable - 2 else 0 for r in range(n)]
        cols = dict(cells, instance=[v + [0] * (n - len(v)) for v in instances])
        for d_col, e in defined:
            advice[d_col[1]] = [_evaluate(e, r, n, cols) if 2 <= r < usable - 2 else rf(wr) for r in range(n)]
    return cs, k, [_arr(c) for c in advice], [_arr(c) for c in fixed], copies, instances


def run_case(device, seed, cache={}, satisfiable=False):
    """one random circuit on the device and on the CPU; returns a description, raises on a mismatch.  `satisfiable`: also
    by the coset routes of the device (every coset at once, and coset by coset with no table set retained)"""
    return compare(device, random_case(seed, satisfiable), seed, cache, satisfiable)


def reference_examples():
    """The fixed part of the corpus: every live example of the reference that runs create_proof, at small sizes, with the
    example's own witness -- (name, (cs, k, advice, fixed, copies, instances)).  All satisfied: `compare` also takes them
    through the coset routes of the device."""
    from halo2_gpu_specific_amd import circuits as c

    yield "simple-example-2 (mini-PLONK)", (c.mini_plonk(), 7) + tuple(c.mini_plonk_synthesize(7)) + ((),)
    yield "range-check", (c.range_check(0, 61, 4), 8) + tuple(c.range_check_synthesize(8, vmin=0, vmax=61, count=150)) + ((),)
    yield "lookup_api", (c.lookup_api(), 7) + tuple(c.lookup_api_synthesize(7)) + ((),)
    yield "lookup_api_set", (c.lookup_api_set(), 8) + tuple(c.lookup_api_set_synthesize(8)) + ((),)
    yield "shuffle_api", (c.shuffle_api(), 6) + tuple(c.shuffle_api_synthesize(6)) + ((),)
    yield "shuffle_api_group", (c.shuffle_api_group(), 7) + tuple(c.shuffle_api_group_synthesize(7)) + ((),)
    yield "shuffle (gates)", (c.shuffle_gates(), 7) + tuple(c.shuffle_gates_synthesize(7)) + ((),)


def run_examples(device, cache={}):
    out = []
    for i, (name, case) in enumerate(reference_examples()):
        out.append("%s -- %s" % (name, compare(device, case, 9000 + i, cache, satisfiable=True, several=False)))
    return out


def compare(device, case, seed, cache={}, satisfiable=False, several=True):
    import oracle_prover as op
    from halo2_gpu_specific_amd import prover
    from halo2_gpu_specific_amd.rng import ProverRng

    cs, k, advice, fixed, copies, instances = case
    if k not in cache:
        params = prover.Params.unsafe_setup(device, k, S_TRAPDOOR)
        cpu = op.OracleDevice(threads=4)
        cache[k] = (params, cpu, op.params_like(cpu, params))
    params, cpu, cparams = cache[k]
    pk = prover.keygen(device, params, cs, fixed, copies)
    cpk = op.keygen(cpu, cparams, cs, fixed, copies)
    assert pk.transcript_repr == cpk.transcript_repr, "seed %d: keys differ" % seed
    for use_gwc in (False, True):
        try:
            want = prover.create_proof_ext(cpu, cparams, cpk, advice, ProverRng(seed), use_gwc, instances=instances)
        except ValueError as e:
            # a satisfied circuit whose constraints stay below the declared degree has a zero top piece of h(X): its
            # commitment is the identity, which the transcript refuses (as the reference's does): not a case
            if "infinity" not in str(e):
                raise
            with_device = None
            try:
                prover.create_proof_ext(device, params, pk, advice, ProverRng(seed), use_gwc, instances=instances)
            except ValueError as e2:
                with_device = str(e2)
            assert with_device == str(e), "seed %d: the CPU prover refused (%s), the device did not" % (seed, e)
            return "seed %d: k=%d degree=%d -- refused by both (%s)" % (seed, k, cs.degree(), e)
        got = prover.create_proof_ext(device, params, pk, advice, ProverRng(seed), use_gwc, instances=instances)
        if got != want:
            first = next(i for i in range(min(len(got), len(want))) if got[i] != want[i])
            raise AssertionError("seed %d (%s): proof differs at byte %d of %d / %d" % (
                seed, "gwc" if use_gwc else "shplonk", first, len(got), len(want)))
    if os.environ.get("H2_FUZZ_HOST_API") == "1":
        # the literal drop-in's data flow on the same circuit (host vectors, the fused host-slice entry points:
        # h2_permutation_product, h2_msm_intt, h2_quotient_poly_coeff, h2_eval_polynomial_batch, h2_quotient_sum): the same bytes
        from halo2_gpu_specific_amd import host_api

        if ("host", k) not in cache:
            H = host_api.HostApiDevice(pinned=(seed % 2 == 0))
            cache[("host", k)] = (H, host_api.params_like(H, params))
        H, hparams = cache[("host", k)]
        hpk = prover.keygen(H, hparams, cs, fixed, copies)
        for use_gwc in (False, True):
            want_h = prover.create_proof_ext(device, params, pk, advice, ProverRng(seed + 7), use_gwc, instances=instances)
            got_h = prover.create_proof_ext(H, hparams, hpk, advice, ProverRng(seed + 7), use_gwc, instances=instances)
            assert got_h == want_h, "seed %d (%s): the host-slice data flow changed the proof" % (seed, "gwc" if use_gwc else "shplonk")
    if several and not satisfiable and seed % 4 == 0:
        # several circuit instances in one proof (plonk/prover.rs:206-232): two more witnesses of the same circuit
        advs, insts = [advice], [instances]
        for ws in (1, 2):
            other = random_case(seed, False, k, witness_seed=ws)
            advs.append(other[2])
            insts.append(other[5])
        got = prover.create_proof_ext(device, params, pk, advs, ProverRng(seed), False, instances=insts)
        want = prover.create_proof_ext(cpu, cparams, cpk, advs, ProverRng(seed), False, instances=insts)
        assert got == want, "seed %d: the proof over three circuit instances differs" % seed
    if satisfiable:
        for kw in (dict(force_cosets=True), dict(eval_cache=0)):
            D2 = prover.Device(**kw)
            params2 = prover.Params(D2, k, params.g, params.g_lagrange, tables=False)
            pk2 = prover.keygen(D2, params2, cs, fixed, copies)
            other = prover.create_proof_ext(D2, params2, pk2, advice, ProverRng(seed), True, instances=instances)
            assert other == got, "seed %d: the coset route (%s) changed the proof" % (seed, kw)
    return "seed %d: k=%d degree=%d advice=%d fixed=%d gates=%d lookups=%d shuffles=%d jit=%s  %d bytes" % (
        seed, k, cs.degree(), cs.num_advice, cs.num_fixed, len(cs.gates), len(cs.lookups), len(cs.shuffles),
        bool(pk.evalh_stats), len(got))


def main():
    from halo2_gpu_specific_amd import prover

    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    satisfiable = len(sys.argv) > 3 and sys.argv[3] == "satisfiable"
    device = prover.Device()
    t_end, done = time.time() + seconds, 0
    for line in run_examples(device):
        print(line, flush=True)
    while time.time() < t_end:
        print(run_case(device, seed, satisfiable=satisfiable), flush=True)
        seed += 1
        done += 1
    print("%d random circuits: device proof bytes == CPU proof bytes (GWC and SHPLONK each)" % done)


if __name__ == "__main__":
    main()
