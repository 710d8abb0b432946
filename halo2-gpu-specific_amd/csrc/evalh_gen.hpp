// evalh_gen.hpp -- evaluate_h as generated straight-line HIP, built inside the library (evalh_gen.cpp; host C++ only).
//
// The reference's cuda `Evaluator::evaluate_h` is self-contained in the host language (plonk/evaluation.rs:1229-1985,
// plonk/evaluation_gpu.rs:594-803: it walks its expression trees and launches prebuilt elementwise kernels).  Here the
// flattened program of one circuit (h2_evalh_desc: constants, rotations, `Calculation`s, value parts, lookup / shuffle
// result calculations, the permutation argument's shape) is compiled ONCE into one or a few straight-line kernels --
// intermediates in registers, every argument term folded in, loads issued ahead of their use -- by hipRTC, cached by the
// hash of the program (memory, then a private directory on disk), and launched whenever a descriptor with that program
// arrives through h2_evaluate_h / h2_evaluate_h_coeff / h2_dev_evaluate_h (evalh.hip).  No Python, no hipcc subprocess:
// any host that can fill the descriptor (the Rust drop-in of INTEGRATION.md) gets the generated kernels.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/halo2_hip.h"

namespace h2 {
namespace evgen {

// ---- what a generated kernel reads: a compact list of column vectors and of uniform scalars ----
enum Table : uint8_t {
    T_FIXED = 0, T_ADVICE, T_INSTANCE, T_PERM_Z, T_PERM_SIGMA, T_LOOKUP_Z, T_LOOKUP_M, T_SHUFFLE_Z, T_L0, T_L_LAST, T_L_ACTIVE,
    T_COUNT
};
struct ColRef {
    uint8_t table;
    uint32_t index;
};
enum ScalarKind : uint8_t {
    SC_ONE = 0,     // R mod r
    SC_CONST,       // constants[arg]
    SC_Y_POW,       // y^arg
    SC_BETA_POW,    // beta^arg
    SC_GAMMA_POW,   // gamma^arg
    SC_THETA,
    SC_DELTA_TERM   // beta * ZETA * DELTA^arg (the permutation argument's `current_delta` of column arg, evaluation.rs:1012,1074)
};
struct ScalarRef {
    uint8_t kind;
    uint32_t arg;
};

// Layout of a generated kernel's single argument (passed by value; the launcher in evalh.hip fills it):
//   offset 0   Fr* values; const Fr* tw_lo; const Fr* tw_hi; u64 row_begin; u64 row_end; u32 extended_k; u32 rot_scale;
//   offset 48  Fr sc[n_scalars]  (32 B each, at least one)
//   then       const Fr* cols[n_cols]  (at least one)
constexpr size_t ARGS_FIXED_BYTES = 48;
constexpr const char* KERNEL_NAME = "h2_evalh_gen";

struct Stage {
    std::string source;        // one translation unit: `extern "C" __global__ void h2_evalh_gen(Args)`
    std::vector<ColRef> cols;  // Args::cols[i]
    std::vector<ScalarRef> scalars;  // Args::sc[i]
    bool accumulate = false;   // values[idx] += (this stage's terms) instead of values[idx] = ...
    bool uses_omega = false;   // reads the power tables of extended_omega (tw_lo / tw_hi)
    uint32_t products = 0;     // field products per row
    uint32_t fused_pairs = 0;  // ... of which this many PAIRS run as one fp_mul2 (a b + c d under one reduction)
    uint32_t statements = 0;
    uint32_t max_live = 0;     // the generator's estimate of simultaneously live field values
    // after compile():
    std::vector<char> code;    // the gfx950 code object
    uint32_t vgprs = 0, agprs = 0, scratch = 0;  // from the code object's metadata
};

struct Generated {
    std::vector<Stage> stages;
    uint32_t terms = 0;             // y-folded terms of the quotient numerator (value parts + argument terms)
    uint32_t products_per_row = 0;  // over all stages
    uint32_t fused_pairs_per_row = 0;
    uint32_t reference_products_per_row = 0;  // the products the formulas of evaluation.rs:875-1219 spend as written
    uint32_t vectors_read = 0;      // distinct column vectors
    bool from_disk = false;
};

struct Options {
    uint32_t group = 6;        // statements per scheduling group
    uint32_t max_ahead = 6;    // loads issued ahead per group (8 VGPRs each)
    uint32_t gap = 30;         // a loaded value unused for this many statements is dropped and loaded again
    uint32_t live_budget = 20; // field values (8 registers each) kept alive at once: statement results + loaded values
    uint32_t lds_args = 320;   // scalars + pointers of more than this many dwords are read through an LDS copy of the
                               // argument block instead of field by field from the kernel arguments (0 never, 1 always)
    uint32_t inline_muls = 40; // programs with at most this many products inline the multiplier (mini-PLONK's 29: 7.35 -> 7.11 ms;
                               // the wide circuit's 120 fully inlined: 0.5 MB of code for no gain)
    uint32_t stage_products = 0;   // cut the program into stages of about this many products (0 = as few stages as fit)
    uint32_t max_cols = 440;   // column pointers per stage (kernel arguments are limited to 4 KiB)
    uint32_t waves = 0;        // ask the compiler for at least this many waves per SIMD (amdgpu_waves_per_eu; 0 = its own choice)
    uint32_t max_regs = 256;   // a stage compiled to more registers than this (or to scratch) is cut in two and rebuilt
    uint32_t mul2 = 96;        // a b + c d under ONE reduction (fp_mul2: 3/4 of the multiply-adds of two products) where both
                               // products -- or a Horner step and a product -- allow it; at most this many per stage (each is
                               // inlined: an out-of-line fp_mul2 needs 67 registers and would save / restore 16 callee-saved
                               // VGPRs through scratch per call).  0 = off
    uint32_t min_group = 3;    // a factor group with fewer members than this is folded term by term instead (with fp_mul2 pairing)
    bool factor = true;        // terms that share a factor (a selector, l_0, l_last, l_active_row) are summed before it multiplies them
    static Options from_env();
};

// Throws std::runtime_error on a malformed program (an index outside its table).
Generated generate(const h2_evalh_desc* d, const Options& opt);
// 32-byte identity of (program, options, generator version, field-layer sources)
void program_hash(const h2_evalh_desc* d, const Options& opt, uint8_t out[32]);
// generate + hipRTC for gfx950 (no device needed), through the on-disk cache; throws std::runtime_error with the
// compiler's log when hipRTC is missing or rejects the source
Generated compile(const h2_evalh_desc* d, const Options& opt);
// the directory of the on-disk cache (H2_JIT_CACHE, default <tmp>/halo2_hip_jit_<uid>): owned by this user, mode 0700 --
// a planted code object would be loaded into the prover; when the default is not private a per-process directory is made
std::string cache_dir();

}  // namespace evgen
}  // namespace h2
