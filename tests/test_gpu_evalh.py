"""GPU parity of evaluate_h (h2_evaluate_h through the C ABI) against the CPU oracle, bit-exact."""
import numpy as np
import pytest

from evalh_cases import oracle_evaluate_h, random_case
from halo2_gpu_specific_amd import evaluation as ev

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,k,ek", [(1, 2, 3), (2, 5, 7), (3, 8, 10), (4, 10, 12), (5, 12, 14), (6, 13, 13)])
def test_evaluate_h_vs_oracle(oracle, seed, k, ek):
    kw = random_case(seed, k, ek, oracle, n_calcs=40)
    b = ev.Builder().build(**kw)
    assert np.array_equal(ev.evaluate_h(b), oracle_evaluate_h(oracle, b))


def test_evaluate_h_shapes(oracle):
    for kwargs in (dict(with_perm=False), dict(lookup_sets=()), dict(n_shuffles=0), dict(lookup_sets=(2,), n_shuffles=1),
                   dict(with_perm=False, lookup_sets=(), n_shuffles=0, n_calcs=3)):
        kw = random_case(11, 6, 8, oracle, **kwargs)
        b = ev.Builder().build(**kw)
        assert np.array_equal(ev.evaluate_h(b), oracle_evaluate_h(oracle, b)), kwargs


def test_evaluate_h_k16(oracle):
    kw = random_case(21, 16, 18, oracle, n_calcs=60)
    b = ev.Builder().build(**kw)
    assert np.array_equal(ev.evaluate_h(b), oracle_evaluate_h(oracle, b))


@pytest.mark.timeout(900)
def test_evaluate_h_k20(oracle):
    """the size class of the proofs (2^20 rows, 2^22 extended points, permutation + lookups + shuffles)"""
    kw = random_case(23, 20, 22, oracle, n_calcs=40)
    b = ev.Builder().build(**kw)
    assert np.array_equal(ev.evaluate_h(b), oracle_evaluate_h(oracle, b))


@pytest.mark.parametrize("seed,k,ek,kwargs", [
    (1, 2, 3, {}), (2, 5, 7, {}), (4, 10, 12, {}), (6, 13, 13, {}), (7, 6, 8, dict(with_perm=False, lookup_sets=(), n_shuffles=0, n_calcs=3)),
    (8, 9, 11, dict(lookup_sets=(2,), n_shuffles=1, n_calcs=80)), (9, 16, 18, dict(n_calcs=60))])
def test_generated_kernel_matches_interpreter_and_oracle(oracle, seed, k, ek, kwargs):
    """the gate program as generated straight-line HIP (jit.py -> hipcc --genco -> h2_jit_load -> desc.jit_function):
    every opcode, challenge powers, rotations, the lookup / shuffle result calculations -- same bits as the interpreter
    and as the CPU oracle"""
    from halo2_gpu_specific_amd import jit

    kw = random_case(seed, k, ek, oracle, **({"n_calcs": 40} | kwargs))
    b = ev.Builder().build(**kw)
    want = oracle_evaluate_h(oracle, b)
    assert np.array_equal(ev.evaluate_h(b), want)
    path = jit.compile_program(kw["rotations"], kw["calculations"], kw["value_parts"], kw["lookups"], kw["shuffles"])
    assert path is not None, "hipcc could not build the generated kernel"
    b.desc.jit_function = jit.load(path)
    assert np.array_equal(ev.evaluate_h(b), want)
