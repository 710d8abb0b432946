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
def test_generated_kernel_matches_interpreter_and_oracle(oracle, monkeypatch, seed, k, ek, kwargs):
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
    # ... and the whole of evaluate_h as one generated kernel: gate program + permutation / lookup / shuffle terms folded in
    # registers (h2_evalh_desc::jit_covers), loads value-numbered and issued a group ahead
    nsets = (len(kw["perm_columns"]) + kw["chunk_len"] - 1) // kw["chunk_len"] if kw["perm_columns"] else 0
    monkeypatch.setenv("H2_EVALH_FUSED", "1")        # (by default only programs of up to 64 products per row are fused)
    fused, covers = jit.compile_program(kw["rotations"], kw["calculations"], kw["value_parts"], kw["lookups"], kw["shuffles"],
                                        perm=dict(n_sets=nsets, chunk_len=kw["chunk_len"], columns=kw["perm_columns"],
                                                  last_rotation=-(kw["blinding_factors"] + 1)))
    assert fused is not None and fused != path
    assert covers == ((ev.JIT_PERMUTATION if nsets else 0) | (ev.JIT_LOOKUPS if kw["lookups"] else 0) |
                      (ev.JIT_SHUFFLES if kw["shuffles"] else 0))
    b.desc.jit_function, b.desc.jit_covers = jit.load(fused), covers
    assert np.array_equal(ev.evaluate_h(b), want)
    # ... and the gate program alone under that generator's load scheduling (what a program too wide to fuse gets)
    gsrc, _ = jit.generate_fused_source(kw["rotations"], kw["calculations"], kw["value_parts"], kw["lookups"], kw["shuffles"],
                                        dict(n_sets=nsets, chunk_len=kw["chunk_len"], columns=kw["perm_columns"],
                                             last_rotation=-(kw["blinding_factors"] + 1)), fold_args=False)
    b.desc.jit_function, b.desc.jit_covers = jit.load(jit.compile_source(gsrc, "_fused")), 0
    assert np.array_equal(ev.evaluate_h(b), want)


@pytest.mark.parametrize("seed,j,k,kwargs", [(31, 3, 5, {}), (32, 5, 8, {}), (33, 9, 11, dict(n_calcs=60)), (34, 2, 6, {}),
                                             (35, 5, 7, dict(with_perm=False, lookup_sets=(), n_shuffles=0, n_calcs=5)),
                                             (36, 4, 14, dict(lookup_sets=(2,), n_shuffles=1))])
def test_evaluate_h_from_coefficient_forms(oracle, seed, j, k, kwargs):
    """h2_evaluate_h_coeff -- coefficient forms in, the numerator on the extended domain out, computed coset by coset
    (the cuda evaluate_h's shape, plonk/evaluation.rs:1229-1241) -- against the oracle's evaluate_h on the oracle's own
    extended cosets (coeff_to_extended, poly/domain.rs:270-287): 2, 4, 4, 1 (extended_k = k), 4 and 4 cosets"""
    import copy

    d, _ = oracle.domain(j, k)
    ek = d.extended_k
    kw = random_case(seed, k, ek, oracle, **({"n_calcs": 40} | kwargs))
    kw["zeta"], kw["extended_omega"] = d.fr("g_coset"), d.fr("extended_omega")
    n = 1 << k
    names = ("fixed", "advice", "instance", "perm_z", "perm_sigma", "lookup_z", "lookup_m", "shuffle_z")
    coeff = copy.copy(kw)
    for name in names:
        coeff[name] = [np.ascontiguousarray(col[:n]) for col in kw[name]]           # random coefficient vectors
    coeff["l0"], coeff["l_last"] = np.ascontiguousarray(kw["l0"][:n]), np.ascontiguousarray(kw["l_last"][:n])
    ext = copy.copy(kw)
    for name in names:
        ext[name] = [oracle.coeff_to_extended(col, d, threads=8) for col in coeff[name]]
    ext["l0"], ext["l_last"] = oracle.coeff_to_extended(coeff["l0"], d, threads=8), oracle.coeff_to_extended(coeff["l_last"], d, threads=8)
    want = oracle_evaluate_h(oracle, ev.Builder().build(**ext))
    got = ev.evaluate_h_coeff(ev.Builder().build(**coeff))
    assert np.array_equal(got, want)
    assert np.array_equal(ev.evaluate_h(ev.Builder().build(**ext)), want)            # the extended-coset entry point agrees


def test_host_lincomb(oracle):
    """h2_lincomb (host buffers; the GWC cuda branch's shape, gwc/prover.rs:57-151): res = sum_j coeffs[j] * polys[j]"""
    import ctypes

    import halo2_gpu_specific_amd as h2
    from halo2_gpu_specific_amd import arithmetic as ar

    n, count = 5000, 5
    polys = [oracle.random_fr(700 + i, n) for i in range(count)]
    coeffs = oracle.random_fr(777, count)
    want = np.zeros((n, 4), dtype=np.uint64)
    for p, c in zip(polys, coeffs):
        want = oracle.eval_op(ar.OP_SUM, want, oracle.eval_op(ar.OP_MUL_C, p, None, 0, 0, c), 0, 0, None)
    res = np.zeros((n, 4), dtype=np.uint64)
    ptrs = (ctypes.c_void_p * count)(*[p.ctypes.data for p in polys])
    assert h2.lib().h2_lincomb(res.ctypes.data, ptrs, coeffs.ctypes.data, count, n) == 0
    assert np.array_equal(res, want)


@pytest.mark.parametrize("seed,k,ek", [(41, 6, 8), (42, 10, 12), (43, 13, 14)])
def test_row_ranges_of_the_device_evaluator(oracle, monkeypatch, seed, k, ek):
    """h2_evalh_desc::row_begin / row_count (one evaluation split over several devices by row range): the interpreter, the
    generated gate kernel with the library's argument kernels, and the fused generated kernel write exactly the rows asked
    for -- the oracle's values there, the buffer untouched elsewhere -- for ranges at the start, across the wrap of the
    rotations, a single row and the whole domain"""
    import ctypes

    import torch

    import halo2_gpu_specific_amd as h2
    from halo2_gpu_specific_amd import jit
    from halo2_gpu_specific_amd._lib import check

    kw = random_case(seed, k, ek, oracle, n_calcs=30, lookup_sets=(1, 2), n_shuffles=1)
    want = oracle_evaluate_h(oracle, ev.Builder().build(**kw))
    size = 1 << ek
    dev = torch.device("cuda", 0)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(dev)  # noqa: E731
    names = ("fixed", "advice", "instance", "perm_z", "perm_sigma", "lookup_z", "lookup_m", "shuffle_z")
    tensors = {name: [up(c) for c in kw[name]] for name in names}
    singles = {name: up(kw[name]) for name in ("l0", "l_last", "l_active_row")}
    dkw = dict(kw)
    for name in names:
        dkw[name] = [t.data_ptr() for t in tensors[name]]
    for name in singles:
        dkw[name] = singles[name].data_ptr()
    nsets = (len(kw["perm_columns"]) + kw["chunk_len"] - 1) // kw["chunk_len"]
    path = jit.compile_program(kw["rotations"], kw["calculations"], kw["value_parts"], kw["lookups"], kw["shuffles"])
    monkeypatch.setenv("H2_EVALH_FUSED", "1")
    fused, covers = jit.compile_program(kw["rotations"], kw["calculations"], kw["value_parts"], kw["lookups"], kw["shuffles"],
                                        perm=dict(n_sets=nsets, chunk_len=kw["chunk_len"], columns=kw["perm_columns"],
                                                  last_rotation=-(kw["blinding_factors"] + 1)))
    L = h2.lib()
    sentinel = 0x5A5A5A5A5A5A5A5A
    for fn, cov in ((None, 0), (jit.load(path), 0), (jit.load(fused), covers)):
        for lo, cnt in ((0, size // 4), (size - 5, 5), (size // 2 - 3, 1), (7, size - 7), (0, 0)):
            b = ev.Builder().build(**dkw, jit_function=fn, jit_covers=cov, row_begin=lo, row_count=cnt)
            out = torch.full((size, 4), sentinel, dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            check(L.h2_dev_evaluate_h(ctypes.byref(b.desc), out.data_ptr(), None), "h2_dev_evaluate_h")
            torch.cuda.synchronize()
            got = out.cpu().numpy().view(np.uint64)
            hi = lo + cnt if cnt else size
            lo_ = lo if cnt else 0
            assert np.array_equal(got[lo_:hi], want[lo_:hi]), (fn is not None, cov, lo, cnt)
            rest = np.concatenate([got[:lo_], got[hi:]])
            assert (rest == np.uint64(sentinel)).all(), "rows outside the range were written"
    # the host-buffer entry points take no row range
    bad = ev.Builder().build(**kw, row_begin=0, row_count=4)
    assert L.h2_evaluate_h(ctypes.byref(bad.desc), np.zeros((size, 4), dtype=np.uint64).ctypes.data) != 0
