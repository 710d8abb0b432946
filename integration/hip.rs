//! halo2_proofs/src/hip.rs -- the Rust side of the MI355X drop-in: `extern "C"` declarations of
//! `include/halo2_hip.h` (libhalo2_hip.so), the `#[repr(C)]` mirror of `h2_evalh_desc`, and the thin wrappers the
//! `#[cfg(feature = "hip")]` bodies call (integration/halo2_proofs_hip.patch adds those bodies, this file and
//! plonk/evaluation_hip.rs).
//!
//! UNCOMPILED in the build image (no Rust toolchain there): kept as source so that a maintainer with `cargo` can
//! apply the patch, point HALO2_HIP_LIB_DIR at the directory holding libhalo2_hip.so and build with
//! `--features hip`.  The wrappers keep the reference's conventions: the same `transmute`s from the generic
//! `C::Scalar` / `C` / `C::Curve` to bn256 memory images (arithmetic.rs:351-352,364-365,391-394,507-508), and a
//! panic on any GPU failure (the reference `unwrap()`s its kernel results, arithmetic.rs:358,360,509).
//!
//! Call sites re-targeted by the patch:
//!   arithmetic.rs   gpu_multiexp*, gpu_multiexp_bound_and_fft, gpu_fft, gpu_ifft          -> msm, msm_intt, ntt, intt
//!   poly/domain.rs  coeff_to_extended :270-287, extended_to_coeff :328-350,
//!                   divide_by_vanishing_poly :354-373                                     -> the functions of the same name
//!   poly/commitment.rs  Params::unsafe_setup :56-124, Params::read :256-294               -> register_params (+ Drop)
//!   plonk/keygen.rs     keygen_pk :330-455, keygen_pk_from_info :458-553                   -> register_proving_key (+ Drop)
//!   poly/multiopen/gwc/prover.rs :57-151  (poly_batch = sum v^i p_i)                      -> lincomb
//!   plonk/evaluation.rs  Evaluator::evaluate_h (cuda) :1229-1241,
//!                        evaluate / evaluate_with_theta (cuda) :2315-2326, :2393-2396     -> plonk/evaluation_hip.rs over
//!                                                                                            H2EvalhDesc + evaluate_h*
#![cfg(feature = "hip")]
#![allow(missing_docs)]

// `hip` supplies the same items `cuda` does (the gpu_* functions, `Evaluator::evaluate_h` with the coefficient-form
// signature): one accelerator back end per build.
#[cfg(feature = "cuda")]
compile_error!("the `hip` and `cuda` features are mutually exclusive");

use crate::arithmetic::{CurveAffine, Group};
use std::os::raw::{c_char, c_int, c_void};

// ---------------------------------------------------------------------------------------------------------------
// the plain-C flattening of `Evaluator` (include/halo2_hip.h, "evaluate_h"): field for field, same order, same types.
// tests/test_integration_patch.py compiles a C program that prints offsetof() of every h2_evalh_desc field and
// compares with H2_EVALH_DESC_OFFSETS below.
// ---------------------------------------------------------------------------------------------------------------
pub const H2_VS_CONSTANT: u32 = 0;
pub const H2_VS_INTERMEDIATE: u32 = 1;
pub const H2_VS_FIXED: u32 = 2;
pub const H2_VS_ADVICE: u32 = 3;
pub const H2_VS_INSTANCE: u32 = 4;

pub const H2_CALC_ADD: u32 = 0;
pub const H2_CALC_SUB: u32 = 1;
pub const H2_CALC_MUL: u32 = 2;
pub const H2_CALC_NEGATE: u32 = 3;
pub const H2_CALC_LC_CHALLENGE: u32 = 4;
pub const H2_CALC_LC_THETA: u32 = 5;
pub const H2_CALC_ADD_CHALLENGE: u32 = 6;
pub const H2_CALC_STORE: u32 = 7;

pub const H2_CHALLENGE_BETA: u32 = 0;
pub const H2_CHALLENGE_GAMMA: u32 = 1;

pub const H2_ANY_ADVICE: u32 = 0;
pub const H2_ANY_FIXED: u32 = 1;
pub const H2_ANY_INSTANCE: u32 = 2;

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct H2ValueSource {
    pub kind: u32,
    pub index: u32,
    pub rot: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct H2Calculation {
    pub op: u32,
    pub a: H2ValueSource,
    pub b: H2ValueSource,
    pub challenge: u32,
    pub power: u32,
}

#[repr(C)]
pub struct H2EvalhDesc {
    pub k: u32,
    pub extended_k: u32,
    pub blinding_factors: u32,
    pub chunk_len: u32,
    pub constants: *const u64,
    pub n_constants: u32,
    pub rotations: *const i32,
    pub n_rotations: u32,
    pub calculations: *const H2Calculation,
    pub n_calculations: u32,
    pub value_parts: *const H2ValueSource,
    pub n_value_parts: u32,
    pub n_lookups: u32,
    pub lookup_sets: *const u32,
    pub lookup_calcs: *const H2Calculation,
    pub n_shuffles: u32,
    pub shuffle_calcs: *const H2Calculation,
    pub fixed: *const *const u64,
    pub n_fixed: u32,
    pub advice: *const *const u64,
    pub n_advice: u32,
    pub instance: *const *const u64,
    pub n_instance: u32,
    pub l0: *const u64,
    pub l_last: *const u64,
    pub l_active_row: *const u64,
    pub n_perm_sets: u32,
    pub perm_z: *const *const u64,
    pub n_perm_columns: u32,
    pub perm_col_type: *const u32,
    pub perm_col_index: *const u32,
    pub perm_sigma: *const *const u64,
    pub lookup_z: *const *const u64,
    pub lookup_m: *const *const u64,
    pub shuffle_z: *const *const u64,
    pub y: [u64; 4],
    pub beta: [u64; 4],
    pub gamma: [u64; 4],
    pub theta: [u64; 4],
    pub delta: [u64; 4],
    pub zeta: [u64; 4],
    pub extended_omega: [u64; 4],
    /// must be null: the library generates, compiles (hipRTC) and caches the program's kernels itself
    pub reserved: *const c_void,
    /// H2_EVALH_INTERPRET = 1 keeps the interpreter kernels for this call
    pub flags: u32,
    pub row_begin: u32,
    pub row_count: u32,
}

/// `h2_evalh_info` (include/halo2_hip.h): what the library generated for a program
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct H2EvalhInfo {
    pub stages: u32,
    pub terms: u32,
    pub products_per_row: u32,
    pub reference_products_per_row: u32,
    pub vectors_read: u32,
    pub max_registers: u32,
    pub scratch_bytes: u32,
    pub from_cache: u32,
    pub fused_pairs_per_row: u32,
}

/// (field, byte offset) of `h2_evalh_desc` on the LP64 ABI both sides are built for; checked against the C header by
/// tests/test_integration_patch.py (and, once compiled, by `desc_layout_matches_header` below).
pub const H2_EVALH_DESC_OFFSETS: &[(&str, usize)] = &[
    ("k", 0),
    ("extended_k", 4),
    ("blinding_factors", 8),
    ("chunk_len", 12),
    ("constants", 16),
    ("n_constants", 24),
    ("rotations", 32),
    ("n_rotations", 40),
    ("calculations", 48),
    ("n_calculations", 56),
    ("value_parts", 64),
    ("n_value_parts", 72),
    ("n_lookups", 76),
    ("lookup_sets", 80),
    ("lookup_calcs", 88),
    ("n_shuffles", 96),
    ("shuffle_calcs", 104),
    ("fixed", 112),
    ("n_fixed", 120),
    ("advice", 128),
    ("n_advice", 136),
    ("instance", 144),
    ("n_instance", 152),
    ("l0", 160),
    ("l_last", 168),
    ("l_active_row", 176),
    ("n_perm_sets", 184),
    ("perm_z", 192),
    ("n_perm_columns", 200),
    ("perm_col_type", 208),
    ("perm_col_index", 216),
    ("perm_sigma", 224),
    ("lookup_z", 232),
    ("lookup_m", 240),
    ("shuffle_z", 248),
    ("y", 256),
    ("beta", 288),
    ("gamma", 320),
    ("theta", 352),
    ("delta", 384),
    ("zeta", 416),
    ("extended_omega", 448),
    ("reserved", 480),
    ("flags", 488),
    ("row_begin", 492),
    ("row_count", 496),
];
pub const H2_EVALH_DESC_SIZE: usize = 504;
pub const H2_VALUE_SOURCE_SIZE: usize = 12;
pub const H2_CALCULATION_SIZE: usize = 36;

#[cfg(test)]
mod layout {
    use super::*;
    #[test]
    fn desc_layout_matches_header() {
        assert_eq!(std::mem::size_of::<H2ValueSource>(), H2_VALUE_SOURCE_SIZE);
        assert_eq!(std::mem::size_of::<H2Calculation>(), H2_CALCULATION_SIZE);
        assert_eq!(std::mem::size_of::<H2EvalhDesc>(), H2_EVALH_DESC_SIZE);
    }
}

extern "C" {
    pub fn h2_version() -> c_int;
    pub fn h2_device_count() -> c_int;
    pub fn h2_last_error() -> *const c_char;
    // best_fft / gpu_fft (arithmetic.rs:495-512,546-554) and gpu_ifft (:515-534): in place, natural order
    pub fn h2_ntt(a: *mut u64, omega: *const u64, log_n: u32) -> c_int;
    pub fn h2_intt(a: *mut u64, omega_inv: *const u64, divisor: *const u64, log_n: u32) -> c_int;
    pub fn h2_intt_to(a: *const u64, out: *mut u64, omega_inv: *const u64, divisor: *const u64, log_n: u32) -> c_int;
    // gpu_multiexp_single_gpu_with_bound (:334-367), gpu_multiexp_bound (:413-440), gpu_multiexp_bound_and_fft (:375-410)
    pub fn h2_msm(scalars: *const u64, bases: *const u64, n: usize, max_bits: u32, out_xyz: *mut u64) -> c_int;
    pub fn h2_msm_multi(scalars: *const u64, bases: *const u64, n: usize, max_bits: u32, out_xyz: *mut u64) -> c_int;
    pub fn h2_msm_intt(
        scalars: *mut u64,
        bases: *const u64,
        n: usize,
        max_bits: u32,
        omega_inv: *const u64,
        divisor: *const u64,
        log_n: u32,
        out_xyz: *mut u64,
    ) -> c_int;
    // poly/domain.rs:270-287,328-350,354-373
    pub fn h2_coeff_to_extended(
        coeffs: *const u64,
        out: *mut u64,
        k: u32,
        extended_k: u32,
        g_coset: *const u64,
        g_coset_inv: *const u64,
        extended_omega: *const u64,
    ) -> c_int;
    pub fn h2_extended_to_coeff(
        a: *const u64,
        out: *mut u64,
        out_len: usize,
        extended_k: u32,
        g_coset: *const u64,
        g_coset_inv: *const u64,
        extended_omega_inv: *const u64,
        extended_ifft_divisor: *const u64,
    ) -> c_int;
    pub fn h2_divide_by_vanishing_poly(a: *mut u64, size: usize, t_evals: *const u64, t_len: usize) -> c_int;
    pub fn h2_eval_op(
        op: c_int,
        res: *mut u64,
        l: *const u64,
        r: *const u64,
        l_rot: i32,
        r_rot: i32,
        size: usize,
        c: *const u64,
    ) -> c_int;
    // gwc/prover.rs:57-151: res = sum_j coeffs[j] * polys[j]
    pub fn h2_lincomb(res: *mut u64, polys: *const *const u64, coeffs: *const u64, count: usize, size: usize) -> c_int;
    pub fn h2_eval_polynomial(poly: *const u64, n: usize, point: *const u64, out: *mut u64) -> c_int;
    pub fn h2_eval_polynomial_batch(polys: *const *const u64, count: usize, n: usize, points: *const u64, out: *mut u64) -> c_int;
    pub fn h2_batch_invert(a: *mut u64, n: usize) -> c_int;
    pub fn h2_random_fr(key: *const u8, n: usize, out: *mut u64) -> c_int;
    pub fn h2_logup_grand_sum(
        z: *mut u64,
        inputs: *const *const u64,
        count: usize,
        table: *const u64,
        m: *const u64,
        n: usize,
        beta: *const u64,
        init: *const u64,
    ) -> c_int;
    pub fn h2_prefix_sum(f: *const u64, n: usize, init: *const u64, z: *mut u64) -> c_int;
    pub fn h2_distribute_powers(a: *mut u64, n: usize, g: *const u64) -> c_int;
    pub fn h2_permutation_sigma(
        out: *mut u64,
        map_col: *const u32,
        map_row: *const u32,
        n: usize,
        delta: *const u64,
        omega: *const u64,
    ) -> c_int;
    pub fn h2_logup_multiplicity(
        table: *const u64,
        inputs: *const *const u64,
        n_inputs: usize,
        usable_rows: usize,
        n: usize,
        m: *mut u64,
        max_bits_out: *mut u32,
    ) -> c_int;
    pub fn h2_quotient_sum(
        out: *mut u64,
        n: usize,
        n_sets: usize,
        counts: *const usize,
        polys: *const *const u64,
        coeffs: *const u64,
        low_counts: *const usize,
        low: *const u64,
        point_counts: *const usize,
        points: *const u64,
        remainders: *mut u64,
    ) -> c_int;
    pub fn h2_permutation_product(
        z: *mut u64,
        values: *const *const u64,
        sigmas: *const *const u64,
        count: usize,
        n: usize,
        beta: *const u64,
        gamma: *const u64,
        delta_pow: *const u64,
        delta: *const u64,
        omega: *const u64,
        init: *const u64,
    ) -> c_int;
    pub fn h2_batch_mont(a: *mut u64, n: usize) -> c_int;
    pub fn h2_batch_unmont(a: *mut u64, n: usize) -> c_int;
    // Evaluator::evaluate_h: extended cosets in (the CPU twin's shape, evaluation.rs:778-1226) / coefficient forms in
    // (the cuda shape, :1229-1241); evaluate_with_theta is h2_evaluate_h with y := theta, extended_k := k
    pub fn h2_evaluate_h(desc: *const H2EvalhDesc, values: *mut u64) -> c_int;
    pub fn h2_evaluate_h_coeff(desc: *const H2EvalhDesc, values: *mut u64) -> c_int;
    pub fn h2_quotient_poly_coeff(
        desc: *const H2EvalhDesc,
        t_evaluations: *const u64,
        t_len: usize,
        g_coset: *const u64,
        g_coset_inv: *const u64,
        extended_omega_inv: *const u64,
        extended_ifft_divisor: *const u64,
        out: *mut u64,
        out_len: usize,
    ) -> c_int;
    // The library turns a descriptor's program into generated kernels (hipRTC, cached by program hash in memory and on
    // disk) the first time it sees it; h2_evalh_prepare does that ahead of the first proof, e.g. from keygen_pk
    pub fn h2_evalh_prepare(desc: *const H2EvalhDesc, info: *mut H2EvalhInfo) -> c_int;
    // resident SRS: Params::g / g_lagrange are uploaded once per device instead of once per MSM
    pub fn h2_bases_register(bases: *const u64, n: usize) -> c_int;
    pub fn h2_bases_unregister(bases: *const u64) -> c_int;
    // resident polynomials: Fr vectors the prover will not modify any more (the proving key's coefficient forms, a proof's final
    // polynomials) are uploaded once per device; every entry point that only READS a vector looks them up
    pub fn h2_poly_register(values: *const u64, n: usize) -> c_int;
    pub fn h2_poly_unregister(values: *const u64) -> c_int;
    pub fn h2_g1_sum(points: *const c_void, count: usize, out_xyz: *mut c_void) -> c_int;
    // device memory and streams for a host without a HIP binding (the device-resident h2_dev_* family of the header takes
    // these pointers: what `Polynomial` would hold instead of a Vec once the data stays on the device)
    pub fn h2_set_device(device: c_int) -> c_int;
    pub fn h2_dev_alloc(bytes: usize, d_out: *mut *mut c_void) -> c_int;
    pub fn h2_dev_free(d_ptr: *mut c_void) -> c_int;
    pub fn h2_host_alloc_pinned(bytes: usize, out: *mut *mut c_void) -> c_int;
    pub fn h2_host_free_pinned(ptr: *mut c_void) -> c_int;
    pub fn h2_stream_create(stream_out: *mut *mut c_void) -> c_int;
    pub fn h2_stream_destroy(stream: *mut c_void) -> c_int;
    pub fn h2_stream_synchronize(stream: *mut c_void) -> c_int;
    pub fn h2_dev_upload(d_dst: *mut c_void, src: *const c_void, bytes: usize, stream: *mut c_void) -> c_int;
    pub fn h2_dev_download(dst: *mut c_void, d_src: *const c_void, bytes: usize, stream: *mut c_void) -> c_int;
    // device memory the library keeps between calls (plans, last-pass twiddle tables): budget / release / report
    pub fn h2_release_plans() -> c_int;
    pub fn h2_set_table_budget(bytes: usize) -> c_int;
    pub fn h2_library_memory_bytes() -> usize;
}

/// The reference unwrap()s its GPU results: keep that behaviour.
pub fn check(rc: c_int, what: &str) {
    if rc != 0 {
        let msg = unsafe { std::ffi::CStr::from_ptr(h2_last_error()) }.to_string_lossy().into_owned();
        panic!("{} failed on the GPU: {}", what, msg);
    }
}

/// Memory-image contract of the shim, the same one the reference's own transmutes rely on: a scalar is 4 x u64
/// (Montgomery), an affine point 8 x u64 {x, y} with identity (0, 0), a projective point 12 x u64 {x, y, z}.
/// A `pairing_bn256` build whose `G1Affine` carried an infinity flag (72-byte stride) would trip this at once.
#[inline]
fn assert_layout<C: CurveAffine>() {
    assert_eq!(std::mem::size_of::<C::Scalar>(), 32, "Fr is not a 32-byte memory image");
    assert_eq!(std::mem::size_of::<C>(), 64, "G1Affine is not a 64-byte {{x, y}} memory image");
    assert_eq!(std::mem::size_of::<C::Curve>(), 96, "G1 is not a 96-byte {{x, y, z}} memory image");
}

/// The 4 x u64 Montgomery image of a scalar (what the descriptor's challenge fields hold).
#[inline]
pub fn limbs<F>(v: &F) -> [u64; 4] {
    assert_eq!(std::mem::size_of::<F>(), 32);
    unsafe { std::mem::transmute_copy::<F, [u64; 4]>(v) }
}

/// `gpu_multiexp_single_gpu_with_bound` (multi = false) / `gpu_multiexp_bound` (multi = true: the library cuts the
/// MSM into ceil(n / N_GPU) chunks over its device pool and folds the partial points, arithmetic.rs:413-440).
pub fn msm<C: CurveAffine>(coeffs: &[C::Scalar], bases: &[C], max_bits: usize, multi: bool) -> C::Curve {
    assert_layout::<C>();
    assert_eq!(coeffs.len(), bases.len());
    let mut out = [0u64; 12];
    let f = if multi { h2_msm_multi } else { h2_msm };
    let rc = unsafe {
        f(
            coeffs.as_ptr() as *const u64,
            bases.as_ptr() as *const u64,
            coeffs.len(),
            max_bits as u32,
            out.as_mut_ptr(),
        )
    };
    check(rc, "multiexp");
    // the reference reads its kernel's result the same way (arithmetic.rs:364-365)
    unsafe { std::mem::transmute_copy::<[u64; 12], C::Curve>(&out) }
}

/// `gpu_multiexp_bound_and_fft` (arithmetic.rs:375-410): commitment of the Lagrange values + in-place iFFT.
pub fn msm_intt<C: CurveAffine>(
    coeffs: &mut [C::Scalar],
    bases: &[C],
    max_bits: usize,
    omega_inv: &C::Scalar,
    divisor: &C::Scalar,
    log_n: u32,
) -> C::Curve {
    assert_layout::<C>();
    assert_eq!(coeffs.len(), 1usize << log_n);
    let mut out = [0u64; 12];
    let rc = unsafe {
        h2_msm_intt(
            coeffs.as_mut_ptr() as *mut u64,
            bases.as_ptr() as *const u64,
            coeffs.len(),
            max_bits as u32,
            omega_inv as *const C::Scalar as *const u64,
            divisor as *const C::Scalar as *const u64,
            log_n,
            out.as_mut_ptr(),
        )
    };
    check(rc, "multiexp_bound_and_ifft");
    unsafe { std::mem::transmute_copy::<[u64; 12], C::Curve>(&out) }
}

/// `gpu_fft` (arithmetic.rs:495-512).  `G` is only ever a scalar field in the prover.
pub fn ntt<G: Group>(a: &mut [G], omega: &G::Scalar, log_n: u32) {
    assert_eq!(std::mem::size_of::<G>(), 32, "best_fft over a non-scalar group is not accelerated");
    assert_eq!(a.len(), 1usize << log_n);
    let rc = unsafe { h2_ntt(a.as_mut_ptr() as *mut u64, omega as *const G::Scalar as *const u64, log_n) };
    check(rc, "fft");
}

/// `gpu_ifft` (arithmetic.rs:515-534): inverse transform with the 1/n scale fused.
pub fn intt<G: Group>(a: &mut [G], omega_inv: &G::Scalar, divisor: &G::Scalar, log_n: u32) {
    assert_eq!(std::mem::size_of::<G>(), 32, "ifft over a non-scalar group is not accelerated");
    assert_eq!(a.len(), 1usize << log_n);
    let rc = unsafe {
        h2_intt(
            a.as_mut_ptr() as *mut u64,
            omega_inv as *const G::Scalar as *const u64,
            divisor as *const G::Scalar as *const u64,
            log_n,
        )
    };
    check(rc, "ifft");
}

/// `EvaluationDomain::coeff_to_extended` (poly/domain.rs:270-287): the zeta-power pre-scale (distribute_powers_zeta,
/// :382-398), the zero padding and the extended NTT are one fused device pass over the 2^k inputs.
pub fn coeff_to_extended<G: Group>(
    coeffs: &[G],
    k: u32,
    extended_k: u32,
    g_coset: &G::Scalar,
    g_coset_inv: &G::Scalar,
    extended_omega: &G::Scalar,
) -> Vec<G> {
    assert_eq!(std::mem::size_of::<G>(), 32, "coeff_to_extended over a non-scalar group is not accelerated");
    assert_eq!(coeffs.len(), 1usize << k);
    let mut out = vec![G::group_zero(); 1usize << extended_k];
    let rc = unsafe {
        h2_coeff_to_extended(
            coeffs.as_ptr() as *const u64,
            out.as_mut_ptr() as *mut u64,
            k,
            extended_k,
            g_coset as *const G::Scalar as *const u64,
            g_coset_inv as *const G::Scalar as *const u64,
            extended_omega as *const G::Scalar as *const u64,
        )
    };
    check(rc, "coeff_to_extended");
    out
}

/// `EvaluationDomain::extended_to_coeff` (poly/domain.rs:328-350): inverse extended NTT, 1 / n_ext and the zeta^-1
/// powers fused into its last pass, truncated to `out_len = n * quotient_poly_degree` coefficients.
pub fn extended_to_coeff<G: Group>(
    a: &[G],
    out_len: usize,
    extended_k: u32,
    g_coset: &G::Scalar,
    g_coset_inv: &G::Scalar,
    extended_omega_inv: &G::Scalar,
    extended_ifft_divisor: &G::Scalar,
) -> Vec<G> {
    assert_eq!(std::mem::size_of::<G>(), 32, "extended_to_coeff over a non-scalar group is not accelerated");
    assert_eq!(a.len(), 1usize << extended_k);
    let mut out = vec![G::group_zero(); out_len];
    let rc = unsafe {
        h2_extended_to_coeff(
            a.as_ptr() as *const u64,
            out.as_mut_ptr() as *mut u64,
            out_len,
            extended_k,
            g_coset as *const G::Scalar as *const u64,
            g_coset_inv as *const G::Scalar as *const u64,
            extended_omega_inv as *const G::Scalar as *const u64,
            extended_ifft_divisor as *const G::Scalar as *const u64,
        )
    };
    check(rc, "extended_to_coeff");
    out
}

/// `EvaluationDomain::divide_by_vanishing_poly` (poly/domain.rs:354-373): a[i] *= t_evaluations[i % t_len], in place.
pub fn divide_by_vanishing_poly<G: Group>(a: &mut [G], t_evaluations: &[G::Scalar]) {
    assert_eq!(std::mem::size_of::<G>(), 32, "divide_by_vanishing_poly over a non-scalar group is not accelerated");
    let rc = unsafe {
        h2_divide_by_vanishing_poly(
            a.as_mut_ptr() as *mut u64,
            a.len(),
            t_evaluations.as_ptr() as *const u64,
            t_evaluations.len(),
        )
    };
    check(rc, "divide_by_vanishing_poly");
}

/// The GWC batching loop `poly_batch = poly_batch * v + poly` (poly/multiopen/gwc/prover.rs:45-151) in closed form:
/// `sum_i v^(m-1-i) polys[i]`, one upload per operand and one fused device pass.
pub fn lincomb<F>(polys: &[&[F]], coeffs: &[F]) -> Vec<F>
where
    F: Copy + Default,
{
    assert_eq!(std::mem::size_of::<F>(), 32);
    assert_eq!(polys.len(), coeffs.len());
    let size = polys.first().map(|p| p.len()).unwrap_or(0);
    assert!(polys.iter().all(|p| p.len() == size));
    let mut out = vec![F::default(); size];
    let ptrs: Vec<*const u64> = polys.iter().map(|p| p.as_ptr() as *const u64).collect();
    let rc = unsafe {
        h2_lincomb(
            out.as_mut_ptr() as *mut u64,
            ptrs.as_ptr(),
            coeffs.as_ptr() as *const u64,
            polys.len(),
            size,
        )
    };
    check(rc, "lincomb");
    out
}

/// One grand-product column of the permutation argument (plonk/permutation/prover.rs:72-165, one set of columns):
/// `z[0] = init`, `z[i + 1] = z[i] * prod_j (values[j][i] + beta delta_pow delta^j omega^i + gamma) / prod_j (values[j][i] +
/// beta sigmas[j][i] + gamma)`.  The per-column products, the batch inversion and the running product stay on the device; the
/// sigma columns (`permutation::ProvingKey::permutations`) are registered with the proving key and cross PCIe once per key.
/// The caller writes its blinding rows and takes `z[n - (blinding_factors + 1)]` as the next set's `init`.
pub fn permutation_product<F>(
    values: &[&[F]],
    sigmas: &[&[F]],
    beta: &F,
    gamma: &F,
    delta_pow: &F,
    delta: &F,
    omega: &F,
    init: &F,
) -> Vec<F>
where
    F: Copy + Default,
{
    assert_eq!(std::mem::size_of::<F>(), 32);
    assert!(!values.is_empty() && values.len() == sigmas.len());
    let n = values[0].len();
    assert!(values.iter().chain(sigmas.iter()).all(|v| v.len() == n));
    let mut z = vec![F::default(); n];
    let vp: Vec<*const u64> = values.iter().map(|v| v.as_ptr() as *const u64).collect();
    let sp: Vec<*const u64> = sigmas.iter().map(|v| v.as_ptr() as *const u64).collect();
    let (b, g, dp, d, w, i0) = (limbs(beta), limbs(gamma), limbs(delta_pow), limbs(delta), limbs(omega), limbs(init));
    let rc = unsafe {
        h2_permutation_product(
            z.as_mut_ptr() as *mut u64,
            vp.as_ptr(),
            sp.as_ptr(),
            values.len(),
            n,
            b.as_ptr(),
            g.as_ptr(),
            dp.as_ptr(),
            d.as_ptr(),
            w.as_ptr(),
            i0.as_ptr(),
        )
    };
    check(rc, "permutation_product");
    z
}

/// `h2_quotient_poly_coeff`: h(X) in coefficient form (`out_len = n * quotient_poly_degree` scalars) from the descriptor
/// `evaluate_h` takes -- `Evaluator::evaluate_h` (plonk/evaluation.rs:1229-1985), `divide_by_vanishing_poly`
/// (poly/domain.rs:354-373) and `extended_to_coeff` (:328-350) without the 2^extended_k values leaving the device.
/// Optional: the patch keeps the reference's three steps (`vanishing::Argument::construct` takes the extended values).
pub fn quotient_poly_coeff<F: Copy + Default>(
    desc: &H2EvalhDesc,
    t_evaluations: &[F],
    g_coset: &F,
    g_coset_inv: &F,
    extended_omega_inv: &F,
    extended_ifft_divisor: &F,
    out_len: usize,
) -> Vec<F> {
    assert_eq!(std::mem::size_of::<F>(), 32);
    let mut out = vec![F::default(); out_len];
    let (g, gi, wi, dv) = (limbs(g_coset), limbs(g_coset_inv), limbs(extended_omega_inv), limbs(extended_ifft_divisor));
    let rc = unsafe {
        h2_quotient_poly_coeff(
            desc as *const H2EvalhDesc,
            t_evaluations.as_ptr() as *const u64,
            t_evaluations.len(),
            g.as_ptr(),
            gi.as_ptr(),
            wi.as_ptr(),
            dv.as_ptr(),
            out.as_mut_ptr() as *mut u64,
            out_len,
        )
    };
    check(rc, "quotient_poly_coeff");
    out
}

/// `polys[j](points[j])` for every j in one device call (plonk/prover.rs:700-790: the evaluations of every committed polynomial
/// at x, omega x, ... -- a rayon `par_iter` over `eval_polynomial_st` in the reference).
pub fn eval_polynomial_batch<F>(polys: &[&[F]], points: &[F]) -> Vec<F>
where
    F: Copy + Default,
{
    assert_eq!(std::mem::size_of::<F>(), 32);
    assert_eq!(polys.len(), points.len());
    let n = polys.first().map(|p| p.len()).unwrap_or(0);
    assert!(polys.iter().all(|p| p.len() == n));
    let ptrs: Vec<*const u64> = polys.iter().map(|p| p.as_ptr() as *const u64).collect();
    let mut out = vec![F::default(); polys.len()];
    let rc = unsafe {
        h2_eval_polynomial_batch(
            ptrs.as_ptr(),
            polys.len(),
            n,
            points.as_ptr() as *const u64,
            out.as_mut_ptr() as *mut u64,
        )
    };
    check(rc, "eval_polynomial_batch");
    out
}

/// One rotation set of a multi-point opening as `quotient_sum` takes it: the polynomials that share the set's points, the
/// coefficient of each in the set's linear combination (the caller folds `v^(R-1-s)` in), the low coefficients subtracted from
/// the combination (the combined low-degree equivalents `r_i(X)`, same factor folded in) and the points to divide by.
pub struct QuotientSet<'a, F> {
    pub polys: Vec<&'a [F]>,
    pub coeffs: Vec<F>,
    pub low: Vec<F>,
    pub points: Vec<F>,
}

/// `sum_s (sum_i coeffs[s][i] polys[s][i](X) - low_s(X)) / prod_j (X - points[s][j])` in one device call
/// (poly/multiopen/shplonk/prover.rs:95-153 -- `quotient_contribution` of every rotation set folded by powers of `v` -- and
/// :205-219 with one set: `div_by_vanishing(l_x, &[u])`).  Returns the n coefficients and, when asked, what every division left
/// (the reference's `must_be_zero`).  Polynomials registered with `register_polys` are read on the device.
pub fn quotient_sum<F>(n: usize, sets: &[QuotientSet<F>], want_remainders: bool) -> (Vec<F>, Vec<F>)
where
    F: Copy + Default,
{
    assert_eq!(std::mem::size_of::<F>(), 32);
    let counts: Vec<usize> = sets.iter().map(|s| s.polys.len()).collect();
    let low_counts: Vec<usize> = sets.iter().map(|s| s.low.len()).collect();
    let point_counts: Vec<usize> = sets.iter().map(|s| s.points.len()).collect();
    let mut ptrs: Vec<*const u64> = Vec::new();
    let (mut coeffs, mut low, mut points): (Vec<F>, Vec<F>, Vec<F>) = (Vec::new(), Vec::new(), Vec::new());
    for s in sets.iter() {
        assert_eq!(s.polys.len(), s.coeffs.len());
        for p in s.polys.iter() {
            assert_eq!(p.len(), n);
            ptrs.push(p.as_ptr() as *const u64);
        }
        coeffs.extend_from_slice(&s.coeffs);
        low.extend_from_slice(&s.low);
        points.extend_from_slice(&s.points);
    }
    let mut out = vec![F::default(); n];
    let mut remainders = vec![F::default(); if want_remainders { points.len() } else { 0 }];
    let rc = unsafe {
        h2_quotient_sum(
            out.as_mut_ptr() as *mut u64,
            n,
            sets.len(),
            counts.as_ptr(),
            ptrs.as_ptr(),
            coeffs.as_ptr() as *const u64,
            low_counts.as_ptr(),
            low.as_ptr() as *const u64,
            point_counts.as_ptr(),
            points.as_ptr() as *const u64,
            if want_remainders { remainders.as_mut_ptr() as *mut u64 } else { std::ptr::null_mut() },
        )
    };
    check(rc, "quotient_sum");
    (out, remainders)
}

/// `h2_evaluate_h_coeff` / `h2_evaluate_h`: `values` receives 2^extended_k scalars.  The descriptor's pointers must
/// outlive the call (plonk/evaluation_hip.rs builds it from borrowed slices on its own stack frame).
pub fn evaluate_h<F: Copy + Default>(desc: &H2EvalhDesc, from_coefficient_forms: bool) -> Vec<F> {
    assert_eq!(std::mem::size_of::<F>(), 32);
    let mut values = vec![F::default(); 1usize << desc.extended_k];
    let rc = unsafe {
        if from_coefficient_forms {
            h2_evaluate_h_coeff(desc as *const H2EvalhDesc, values.as_mut_ptr() as *mut u64)
        } else {
            h2_evaluate_h(desc as *const H2EvalhDesc, values.as_mut_ptr() as *mut u64)
        }
    };
    check(rc, "evaluate_h");
    values
}

/// Builds (or finds in the cache) the generated kernels of a descriptor's program on the current device, so that the first
/// `evaluate_h` of a proof does not pay the hipRTC compile; column pointers inside `desc` are not read.  Optional: without
/// it the first call does the same work.  Returns None when the library keeps the interpreter kernels (H2_EVALH_JIT=0,
/// no libhiprtc.so).
pub fn prepare_evaluate_h(desc: &H2EvalhDesc) -> Option<H2EvalhInfo> {
    let mut info = H2EvalhInfo::default();
    let rc = unsafe { h2_evalh_prepare(desc as *const H2EvalhDesc, &mut info as *mut H2EvalhInfo) };
    if rc == 0 {
        Some(info)
    } else {
        None
    }
}

/// Called by `Params::unsafe_setup` / `Params::read` (poly/commitment.rs:56-124,256-294): later `commit*` calls whose
/// bases lie inside a registered range skip the 64 B/point upload and run over the device copy's shifted-base table
/// (include/halo2_hip.h, h2_dev_bases_precompute; built by the library on the first MSM).  `Params`'s `Drop`
/// (added by the patch) unregisters before the vectors go.
pub fn register_params<C: CurveAffine>(g: &[C], g_lagrange: &[C]) {
    assert_layout::<C>();
    unsafe {
        check(h2_bases_register(g.as_ptr() as *const u64, g.len()), "bases_register");
        check(h2_bases_register(g_lagrange.as_ptr() as *const u64, g_lagrange.len()), "bases_register");
    }
}

pub fn unregister_params<C: CurveAffine>(g: &[C], g_lagrange: &[C]) {
    unsafe {
        check(h2_bases_unregister(g.as_ptr() as *const u64), "bases_unregister");
        check(h2_bases_unregister(g_lagrange.as_ptr() as *const u64), "bases_unregister");
    }
}

/// Registers coefficient / value vectors that will not change while registered (include/halo2_hip.h, h2_poly_register): the
/// host-slice calls that only READ a vector -- `evaluate_h`'s columns, `eval_polynomial`, the operands of `lincomb`, the
/// dividend of `kate_division` -- then use a device copy uploaded once per device instead of crossing PCIe per call.
pub fn register_polys<F>(polys: &[&[F]]) {
    assert_eq!(std::mem::size_of::<F>(), 32);
    for p in polys {
        if !p.is_empty() {
            unsafe { check(h2_poly_register(p.as_ptr() as *const u64, p.len()), "poly_register") };
        }
    }
}

pub fn unregister_polys<F>(polys: &[&[F]]) {
    for p in polys {
        if !p.is_empty() {
            unsafe { check(h2_poly_unregister(p.as_ptr() as *const u64), "poly_unregister") };
        }
    }
}

/// RAII form of `register_polys` for vectors that are final for the rest of a scope -- a proof's advice / product / quotient
/// coefficient forms from the point where they are made to the end of `create_proof` (plonk/prover.rs:639-850): every way out
/// of the scope, `?` included, unregisters.  The vectors must not be moved out of their `Vec`s or modified while the guard lives.
pub struct RegisteredPolys {
    ptrs: Vec<*const u64>,
}

impl RegisteredPolys {
    pub fn new<'a, F: 'a, I: IntoIterator<Item = &'a [F]>>(polys: I) -> Self {
        assert_eq!(std::mem::size_of::<F>(), 32);
        let mut ptrs = Vec::new();
        for p in polys {
            if !p.is_empty() {
                unsafe { check(h2_poly_register(p.as_ptr() as *const u64, p.len()), "poly_register") };
                ptrs.push(p.as_ptr() as *const u64);
            }
        }
        RegisteredPolys { ptrs }
    }
}

impl Drop for RegisteredPolys {
    fn drop(&mut self) {
        for p in self.ptrs.iter() {
            unsafe { h2_poly_unregister(*p) };
        }
    }
}

/// The vectors of a proving key that every proof reads and none writes (plonk.rs:226-240 under `hip`: coefficient forms):
/// `fixed_polys`, `permutation.polys`, `permutation.permutations`, `l0`, `l_last`, `l_active_row`.  Called at the end of `keygen_pk` / `keygen_pk_from_info`
/// (plonk/keygen.rs:442-455, :540-553); `ProvingKey`'s `Drop` (added by the patch) unregisters before the vectors go.
/// A proving key is not moved out of its `Vec`s after keygen: the registered addresses are those of the heap buffers.
pub fn proving_key_polys<'a, C: CurveAffine>(pk: &'a crate::plonk::ProvingKey<C>) -> Vec<&'a [C::Scalar]> {
    let mut out: Vec<&[C::Scalar]> = Vec::new();
    for p in pk.fixed_polys.iter() {
        out.push(&p.values[..]);
    }
    for p in pk.permutation.polys.iter() {
        out.push(&p.values[..]);
    }
    for p in pk.permutation.permutations.iter() {
        out.push(&p.values[..]); // the sigma columns' values: `permutation_product` reads them in every proof
    }
    out.push(&pk.l0.values[..]);
    out.push(&pk.l_last.values[..]);
    out.push(&pk.l_active_row.values[..]); // extended values: what `evaluate_h` takes as they are
    out
}

pub fn register_proving_key<C: CurveAffine>(pk: &crate::plonk::ProvingKey<C>) {
    register_polys(&proving_key_polys(pk));
}

pub fn unregister_proving_key<C: CurveAffine>(pk: &crate::plonk::ProvingKey<C>) {
    unregister_polys(&proving_key_polys(pk));
}

/// `N_GPU`'s default (plonk/prover.rs:56-74: `Device::all().len()` under cuda): the devices the library's pool sees.
pub fn device_count() -> usize {
    let n = unsafe { h2_device_count() };
    assert!(n > 0, "no HIP device visible");
    n as usize
}
