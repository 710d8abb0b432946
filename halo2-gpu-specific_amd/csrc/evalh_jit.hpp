// evalh_jit.hpp -- the launch interface between libhalo2_hip.so and a generated gate-program kernel.
//
// The interpreter k_evalh_expr (evalh.hip) walks the flattened `Calculation` program of Evaluator::evaluate_h
// (plonk/evaluation.rs:875-997) per row and keeps every intermediate in memory: for a circuit with hundreds of gates
// that traffic is the bound (3e10 multiplications/s, a quarter of the multiplier ceiling).  halo2-gpu-specific_amd/jit.py
// turns one program into straight-line HIP -- intermediates become registers -- which hipcc compiles once per circuit
// at keygen into a code object; `h2_jit_load` loads it and the descriptor's `jit_function` makes evaluate_h launch it
// in place of the interpreter.  Outputs and layout are identical: values[idx], then the lookup / shuffle compressed
// expressions [slot][idx].
#pragma once
#include "field.hpp"

namespace h2 {

struct JitArgs {
    const Fr* constants;
    const Fr* const* fixed;
    const Fr* const* advice;
    const Fr* const* instance;
    Fr* values;
    Fr* lk_out;
    Fr* sh_out;
    uint32_t extended_k, rot_scale;
    Fr y, beta, gamma, theta;
    // for kernels that also fold the permutation / lookup / shuffle terms (h2_evalh_desc::jit_covers): the argument
    // columns of evaluate_h (plonk/evaluation.rs:1004-1219) and the constants of k_evalh_perm
    const Fr* const* perm_z;
    const Fr* const* perm_sigma;
    const Fr *l0, *l_last, *l_active_row;
    const Fr *tw_lo, *tw_hi;          // extended_omega^i = tw_lo[i & 4095] * tw_hi[i >> 12] (tw_lo[i] alone up to 2^12 points)
    const Fr* const* lookup_z;
    const Fr* const* lookup_m;
    const Fr* const* shuffle_z;
    Fr delta, delta_start;            // DELTA, beta * ZETA
    size_t row_begin, row_end;        // the rows to evaluate (h2_evalh_desc::row_begin / row_count; the whole domain by default)
};

}  // namespace h2
