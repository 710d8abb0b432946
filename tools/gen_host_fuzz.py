"""The evaluate_h generator fuzzed WITHOUT a GPU: for random circuits (tools/prover_fuzz.random_case), the product circuits
and random Evaluator programs (tests/evalh_cases.random_case), the source libhalo2_hip.so generates is compiled for the host
with g++ and run on random columns against the CPU oracle's evaluate_h, under random generator options (grouping, fold order,
stage size, fp_mul2 pairing, live budget, gaps).  The harness is tests/test_evalh_host_exec.py's.
usage: python tools/gen_host_fuzz.py [seconds] [seed]"""
import os
import random
import sys
import tempfile
import time
from pathlib import Path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np  # noqa: E402

import evalh_cases  # noqa: E402
import prover_fuzz  # noqa: E402
import test_evalh_host_exec as hx  # noqa: E402
from h2util import Oracle, fr_mont  # noqa: E402
from halo2_gpu_specific_amd import circuits, evaluation as ev, prover  # noqa: E402

ROOT_OF_UNITY = evalh_cases.ROOT


def circuit_case(cs, k, oracle, seed):
    """Builder kwargs of a circuit's program over random columns (what evaluate_h computes does not care what they mean)"""
    ek = k + max(1, (cs.degree() - 1 - 1).bit_length())
    size = 1 << ek
    graph, value_parts, lookup_calcs, shuffle_calcs = prover.compile_evaluator(cs)
    ncols, chunk = len(cs.perm_columns), cs.degree() - 2
    nsets = (ncols + chunk - 1) // chunk if ncols else 0
    nz = [len(sets) for _, _, sets in cs.lookups]
    col = [0]

    def rnd():
        col[0] += 1
        return oracle.random_fr(seed * 1000 + col[0], size)

    ch = random.Random(seed)
    omega = pow(ROOT_OF_UNITY, 1 << (28 - ek), evalh_cases.R_MOD)
    return dict(
        k=k, extended_k=ek, blinding_factors=cs.blinding_factors(), chunk_len=chunk,
        constants=np.array([prover.fr_to_mont_limbs(c) for c in graph.constants], dtype=np.uint64), rotations=graph.rotations,
        calculations=graph.calculations, value_parts=value_parts, lookups=lookup_calcs, shuffles=shuffle_calcs,
        fixed=[rnd() for _ in range(cs.num_fixed)], advice=[rnd() for _ in range(cs.num_advice)],
        instance=[rnd() for _ in range(cs.num_instance)], l0=rnd(), l_last=rnd(), l_active_row=rnd(),
        perm_z=[rnd() for _ in range(nsets)], perm_columns=[(prover._ANY[kd], i) for kd, i in cs.perm_columns],
        perm_sigma=[rnd() for _ in range(ncols)], lookup_z=[rnd() for _ in range(sum(nz))], lookup_m=[rnd() for _ in range(len(nz))],
        shuffle_z=[rnd() for _ in range(len(shuffle_calcs))],
        y=fr_mont(ch.randrange(evalh_cases.R_MOD)), beta=fr_mont(ch.randrange(evalh_cases.R_MOD)),
        gamma=fr_mont(ch.randrange(evalh_cases.R_MOD)), theta=fr_mont(ch.randrange(evalh_cases.R_MOD)),
        delta=fr_mont(evalh_cases.DELTA), zeta=fr_mont(evalh_cases.ZETA), extended_omega=fr_mont(omega))


def random_options(rnd):
    env = {}
    if rnd.random() < 0.3:
        env["H2_JIT_FACTOR"] = "0"
    if rnd.random() < 0.4:
        env["H2_JIT_STAGE_PRODUCTS"] = str(rnd.choice([4, 8, 13, 20, 40]))
    if rnd.random() < 0.3:
        env["H2_JIT_MUL2"] = str(rnd.choice([0, 1, 3, 96]))
    if rnd.random() < 0.3:
        env["H2_JIT_LIVE"] = str(rnd.choice([4, 6, 10, 20, 40]))
    if rnd.random() < 0.3:
        env["H2_JIT_GAP"] = str(rnd.choice([1, 2, 4, 30, 200]))
    if rnd.random() < 0.3:
        env["H2_JIT_GROUP"] = str(rnd.choice([1, 2, 3, 6, 12]))
    if rnd.random() < 0.3:
        env["H2_JIT_MAX_AHEAD"] = str(rnd.choice([1, 2, 6, 12]))
    if rnd.random() < 0.3:
        env["H2_JIT_MIN_GROUP"] = str(rnd.choice([1, 2, 3, 5]))
    if rnd.random() < 0.2:
        env["H2_JIT_INLINE_MULS"] = str(rnd.choice([0, 1000]))
    if rnd.random() < 0.3:
        env["H2_JIT_LDS_ARGS"] = str(rnd.choice([1, 40, 1000000]))
    return env


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rnd = random.Random(seed)
    oracle = Oracle.get()
    fixed_corpus = [("mini-PLONK", circuits.mini_plonk(), 4), ("wide-2", circuits.wide(2), 4), ("wide-6", circuits.wide(6), 3),
                    ("range-check", circuits.range_check(), 5)]
    t_end, cases, kinds = time.time() + budget, 0, {}
    with tempfile.TemporaryDirectory() as tmp:
        os.environ["H2_JIT_CACHE"] = os.path.join(tmp, "cache")
        while time.time() < t_end:
            pick = rnd.random()
            if cases < len(fixed_corpus):
                name, cs, k = fixed_corpus[cases]
                kw = circuit_case(cs, k, oracle, seed * 7919 + cases)
            elif pick < 0.6:
                s = rnd.randrange(1 << 30)
                cs, k, *_ = prover_fuzz.random_case(s, satisfiable=bool(s & 1), k=rnd.choice([3, 4, 5]))
                name, kw = "circuit-%d" % s, circuit_case(cs, k, oracle, s % 100003)
            else:
                s = rnd.randrange(1 << 20)
                k = rnd.choice([2, 3, 4, 5])
                ek = k + rnd.choice([0, 1, 2, 3])
                name = "program-%d" % s
                kw = evalh_cases.random_case(s, k, ek, oracle, n_calcs=rnd.choice([3, 10, 24, 40, 80]),
                                             with_perm=rnd.random() < 0.7, lookup_sets=rnd.choice([(), (1,), (1, 3), (2, 2, 1)]),
                                             n_shuffles=rnd.choice([0, 1, 2]))
            env = random_options(rnd)
            for key in [key for key in os.environ if key.startswith("H2_JIT_") and key != "H2_JIT_CACHE"]:
                del os.environ[key]
            os.environ.update(env)
            b = ev.Builder().build(**kw)
            want = evalh_cases.oracle_evaluate_h(oracle, b)
            got, stages = hx._run_generated(Path(tmp), b, kw, "f%d" % cases)
            if not np.array_equal(got, want):
                print("MISMATCH %s options %s: first wrong row %d (seed %d, case %d)" % (
                    name, env, int(np.argmax((got != want).any(axis=1))), seed, cases))
                sys.exit(1)
            for f in Path(tmp).glob("f%d_*" % cases):
                f.unlink()
            cases += 1
            kinds[name.split("-")[0]] = kinds.get(name.split("-")[0], 0) + 1
    print("gen_host_fuzz: %d programs (%s) generated, compiled for the host and run: all equal to the oracle's evaluate_h (seed %d)" % (
        cases, ", ".join("%d %s" % (v, k_) for k_, v in sorted(kinds.items())), seed))


if __name__ == "__main__":
    main()
