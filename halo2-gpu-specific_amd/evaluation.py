"""ctypes mirror of the flattened `Evaluator` descriptor (include/halo2_hip.h, h2_evalh_desc) and the
evaluate_h entry point -- Evaluator::evaluate_h, plonk/evaluation.rs:778-1226 / :1229-1985."""
import ctypes

import numpy as np

from ._lib import check, lib

VS_CONSTANT, VS_INTERMEDIATE, VS_FIXED, VS_ADVICE, VS_INSTANCE = range(5)
CALC_ADD, CALC_SUB, CALC_MUL, CALC_NEGATE, CALC_LC_CHALLENGE, CALC_LC_THETA, CALC_ADD_CHALLENGE, CALC_STORE = range(8)
CHALLENGE_BETA, CHALLENGE_GAMMA = 0, 1
EVALH_INTERPRET = 1                                        # h2_evalh_desc::flags
ANY_ADVICE, ANY_FIXED, ANY_INSTANCE = 0, 1, 2

_u32, _vp = ctypes.c_uint32, ctypes.c_void_p
_fr = ctypes.c_uint64 * 4


class ValueSource(ctypes.Structure):
    _fields_ = [("kind", _u32), ("index", _u32), ("rot", _u32)]


class Calculation(ctypes.Structure):
    _fields_ = [("op", _u32), ("a", ValueSource), ("b", ValueSource), ("challenge", _u32), ("power", _u32)]


class EvalHDesc(ctypes.Structure):
    _fields_ = [
        ("k", _u32), ("extended_k", _u32), ("blinding_factors", _u32), ("chunk_len", _u32),
        ("constants", _vp), ("n_constants", _u32),
        ("rotations", _vp), ("n_rotations", _u32),
        ("calculations", _vp), ("n_calculations", _u32),
        ("value_parts", _vp), ("n_value_parts", _u32),
        ("n_lookups", _u32), ("lookup_sets", _vp), ("lookup_calcs", _vp),
        ("n_shuffles", _u32), ("shuffle_calcs", _vp),
        ("fixed", _vp), ("n_fixed", _u32),
        ("advice", _vp), ("n_advice", _u32),
        ("instance", _vp), ("n_instance", _u32),
        ("l0", _vp), ("l_last", _vp), ("l_active_row", _vp),
        ("n_perm_sets", _u32), ("perm_z", _vp),
        ("n_perm_columns", _u32), ("perm_col_type", _vp), ("perm_col_index", _vp),
        ("perm_sigma", _vp),
        ("lookup_z", _vp), ("lookup_m", _vp),
        ("shuffle_z", _vp),
        ("y", _fr), ("beta", _fr), ("gamma", _fr), ("theta", _fr),
        ("delta", _fr), ("zeta", _fr), ("extended_omega", _fr),
        ("reserved", _vp), ("flags", _u32), ("row_begin", _u32), ("row_count", _u32),
    ]


def vs(kind, index=0, rot=0):
    return ValueSource(kind, index, rot)


def calc(op, a, b=None, challenge=0, power=0):
    return Calculation(op, a, b if b is not None else ValueSource(0, 0, 0), challenge, power)


class Builder:
    """Keeps every array the descriptor points to alive.  Columns are numpy (size, 4) uint64 arrays (host
    variant) or integer device addresses (device variant)."""

    def __init__(self):
        self.keep = []
        self.desc = EvalHDesc()

    def _arr(self, ctype, items):
        a = (ctype * max(len(items), 1))(*items)
        self.keep.append(a)
        return ctypes.cast(a, _vp)

    def _ptrs(self, cols):
        addrs = []
        for c in cols:
            if isinstance(c, np.ndarray):
                assert c.dtype == np.uint64 and c.flags["C_CONTIGUOUS"]
                self.keep.append(c)
                addrs.append(c.ctypes.data)
            else:
                addrs.append(int(c))
        return self._arr(ctypes.c_void_p, addrs)

    def _one(self, c):
        if c is None:
            return None
        if isinstance(c, np.ndarray):
            self.keep.append(c)
            return c.ctypes.data
        return int(c)

    def build(self, *, k, extended_k, blinding_factors, chunk_len, constants, rotations, calculations, value_parts,
              lookups=(), shuffles=(), fixed=(), advice=(), instance=(), l0=None, l_last=None, l_active_row=None,
              perm_z=(), perm_columns=(), perm_sigma=(), lookup_z=(), lookup_m=(), shuffle_z=(), y, beta, gamma, theta,
              delta, zeta, extended_omega, flags=0, row_begin=0, row_count=0):
        """lookups: list of (table_calc, [product_calcs], [sum_calcs]); shuffles: list of (input_calc, shuffle_calc);
        perm_columns: list of (ANY_*, index)."""
        d = self.desc
        d.k, d.extended_k, d.blinding_factors, d.chunk_len = k, extended_k, blinding_factors, chunk_len
        consts = np.ascontiguousarray(constants, dtype=np.uint64).reshape(-1, 4)
        self.keep.append(consts)
        d.constants, d.n_constants = consts.ctypes.data, len(consts)
        d.rotations, d.n_rotations = self._arr(ctypes.c_int32, list(rotations)), len(rotations)
        d.calculations, d.n_calculations = self._arr(Calculation, list(calculations)), len(calculations)
        d.value_parts, d.n_value_parts = self._arr(ValueSource, list(value_parts)), len(value_parts)
        flat, sets = [], []
        for table, prods, sums in lookups:
            assert len(prods) == len(sums) >= 1
            sets.append(len(prods))
            flat.append(table)
            for pc, sc in zip(prods, sums):
                flat += [pc, sc]
        d.n_lookups, d.lookup_sets, d.lookup_calcs = len(sets), self._arr(_u32, sets), self._arr(Calculation, flat)
        sflat = [c for pair in shuffles for c in pair]
        d.n_shuffles, d.shuffle_calcs = len(shuffles), self._arr(Calculation, sflat)
        d.fixed, d.n_fixed = self._ptrs(fixed), len(fixed)
        d.advice, d.n_advice = self._ptrs(advice), len(advice)
        d.instance, d.n_instance = self._ptrs(instance), len(instance)
        d.l0, d.l_last, d.l_active_row = self._one(l0), self._one(l_last), self._one(l_active_row)
        d.n_perm_sets, d.perm_z = len(perm_z), self._ptrs(perm_z)
        d.n_perm_columns = len(perm_columns)
        d.perm_col_type = self._arr(_u32, [t for t, _ in perm_columns])
        d.perm_col_index = self._arr(_u32, [i for _, i in perm_columns])
        d.perm_sigma = self._ptrs(perm_sigma)
        assert len(lookup_z) == sum(sets) and len(lookup_m) == len(sets) and len(shuffle_z) == len(shuffles)
        d.lookup_z, d.lookup_m, d.shuffle_z = self._ptrs(lookup_z), self._ptrs(lookup_m), self._ptrs(shuffle_z)
        d.reserved = None
        d.flags = flags
        d.row_begin, d.row_count = row_begin, row_count
        for name, val in (("y", y), ("beta", beta), ("gamma", gamma), ("theta", theta), ("delta", delta), ("zeta", zeta),
                          ("extended_omega", extended_omega)):
            setattr(d, name, _fr(*[int(x) for x in val]))
        return self


def _rebind(self, *, fixed, advice, instance, y, theta):
    """the same program over other columns / another challenge: only the pointer tables and the two scalars change
    (what a cached theta-compression descriptor needs from one proof to the next)"""
    d = self.desc
    assert (d.n_fixed, d.n_advice, d.n_instance) == (len(fixed), len(advice), len(instance))
    self.bound = [self._ptrs(fixed), self._ptrs(advice), self._ptrs(instance)]
    del self.keep[-3:]                       # _ptrs parked them in `keep`; `bound` owns them (replaced on the next rebind)
    d.fixed, d.advice, d.instance = self.bound
    d.y = _fr(*[int(x) for x in y])
    d.theta = _fr(*[int(x) for x in theta])
    return self


Builder.rebind = _rebind


class EvalHInfo(ctypes.Structure):
    """h2_evalh_info: what the library generated for a program"""
    _fields_ = [("stages", _u32), ("terms", _u32), ("products_per_row", _u32), ("reference_products_per_row", _u32),
                ("vectors_read", _u32), ("max_registers", _u32), ("scratch_bytes", _u32), ("from_cache", _u32),
                ("fused_pairs_per_row", _u32)]

    def as_dict(self):
        return {name: int(getattr(self, name)) for name, _ in self._fields_}


def prepare(builder):
    """h2_evalh_prepare: generate + compile + load the program's kernels on the current device (keygen time) -> dict"""
    info = EvalHInfo()
    check(lib().h2_evalh_prepare(ctypes.byref(builder.desc), ctypes.byref(info)), "h2_evalh_prepare")
    return info.as_dict()


def compile_only(builder):
    """h2_evalh_compile: generate + hipRTC into the caches, no device needed -> dict"""
    info = EvalHInfo()
    check(lib().h2_evalh_compile(ctypes.byref(builder.desc), ctypes.byref(info)), "h2_evalh_compile")
    return info.as_dict()


def generated_source(builder, stage=0):
    """h2_evalh_source: the HIP text of one stage"""
    n = ctypes.c_size_t(0)
    check(lib().h2_evalh_source(ctypes.byref(builder.desc), stage, None, 0, ctypes.byref(n)), "h2_evalh_source")
    buf = ctypes.create_string_buffer(n.value + 1)
    check(lib().h2_evalh_source(ctypes.byref(builder.desc), stage, buf, n.value + 1, ctypes.byref(n)), "h2_evalh_source")
    return buf.value.decode()


def stage_args(builder, stage, values_ptr, tw_lo_ptr, tw_hi_ptr, row_begin, row_end):
    """h2_evalh_stage_args: the bytes stage `stage` of the generated program receives by value for this descriptor"""
    n = ctypes.c_size_t(0)
    buf = ctypes.create_string_buffer(4096)
    check(lib().h2_evalh_stage_args(ctypes.byref(builder.desc), stage, values_ptr, tw_lo_ptr, tw_hi_ptr, row_begin, row_end, buf, 4096,
                                    ctypes.byref(n)), "h2_evalh_stage_args")
    return buf.raw[:n.value]


def generated_launches():
    return int(lib().h2_evalh_generated_launches())


def evaluate_h(builder):
    """host buffers in, numpy (2^extended_k, 4) out"""
    out = np.zeros((1 << builder.desc.extended_k, 4), dtype=np.uint64)
    check(lib().h2_evaluate_h(ctypes.byref(builder.desc), out.ctypes.data_as(_vp)), "h2_evaluate_h")
    return out


def evaluate_h_coeff(builder):
    """h2_evaluate_h_coeff: the descriptor's column pointers are host COEFFICIENT vectors of 2^k elements (l_active_row
    extended values); numpy (2^extended_k, 4) out -- the shape of the reference's cuda evaluate_h (evaluation.rs:1229-1241)"""
    out = np.zeros((1 << builder.desc.extended_k, 4), dtype=np.uint64)
    check(lib().h2_evaluate_h_coeff(ctypes.byref(builder.desc), out.ctypes.data_as(_vp)), "h2_evaluate_h_coeff")
    return out
