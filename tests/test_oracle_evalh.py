"""Pin the C oracle's evaluate_h (oracle_evaluate_h, restating plonk/evaluation.rs:778-1226) against an
independent Python big-integer restatement on small random Evaluator programs.  CPU only."""
import pytest

from evalh_cases import oracle_evaluate_h, python_evaluate_h, random_case
from halo2_gpu_specific_amd import evaluation as ev
from h2util import from_mont


@pytest.mark.parametrize("seed,k,ek", [(1, 2, 3), (2, 3, 5), (3, 4, 5), (4, 3, 3)])
def test_oracle_evaluate_h_matches_bigint(oracle, seed, k, ek):
    kw = random_case(seed, k, ek, oracle)
    b = ev.Builder().build(**kw)
    assert from_mont(oracle_evaluate_h(oracle, b)) == python_evaluate_h(kw)


def test_oracle_evaluate_h_degenerate(oracle):
    # no permutation, no lookups, no shuffles: gates only
    kw = random_case(9, 3, 4, oracle, with_perm=False, lookup_sets=(), n_shuffles=0)
    b = ev.Builder().build(**kw)
    assert from_mont(oracle_evaluate_h(oracle, b)) == python_evaluate_h(kw)
