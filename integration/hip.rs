//! halo2_proofs/src/hip.rs -- the Rust side of the MI355X drop-in: `extern "C"` declarations of
//! `include/halo2_hip.h` (libhalo2_hip.so) and the thin wrappers the `#[cfg(feature = "hip")]` bodies in
//! `arithmetic.rs` call (integration/halo2_proofs_hip.patch adds those bodies and this file).
//!
//! UNCOMPILED in the build image (no Rust toolchain there): kept as source so that a maintainer with `cargo` can
//! apply the patch, point HALO2_HIP_LIB_DIR at the directory holding libhalo2_hip.so and build with
//! `--features hip`.  The wrappers keep the reference's conventions: the same `transmute`s from the generic
//! `C::Scalar` / `C` / `C::Curve` to bn256 memory images (arithmetic.rs:351-352,364-365,391-394,507-508), and a
//! panic on any GPU failure (the reference `unwrap()`s its kernel results, arithmetic.rs:358,360,509).
#![cfg(feature = "hip")]
#![allow(missing_docs)]

use crate::arithmetic::{CurveAffine, Group};
use std::os::raw::{c_char, c_int, c_void};

extern "C" {
    pub fn h2_version() -> c_int;
    pub fn h2_device_count() -> c_int;
    pub fn h2_last_error() -> *const c_char;
    // best_fft / gpu_fft (arithmetic.rs:495-512,546-554) and gpu_ifft (:515-534): in place, natural order
    pub fn h2_ntt(a: *mut u64, omega: *const u64, log_n: u32) -> c_int;
    pub fn h2_intt(a: *mut u64, omega_inv: *const u64, divisor: *const u64, log_n: u32) -> c_int;
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
    pub fn h2_eval_polynomial(poly: *const u64, n: usize, point: *const u64, out: *mut u64) -> c_int;
    pub fn h2_batch_invert(a: *mut u64, n: usize) -> c_int;
    pub fn h2_batch_mont(a: *mut u64, n: usize) -> c_int;
    pub fn h2_batch_unmont(a: *mut u64, n: usize) -> c_int;
    // resident SRS: Params::g / g_lagrange are uploaded once per device instead of once per MSM
    pub fn h2_bases_register(bases: *const u64, n: usize) -> c_int;
    pub fn h2_bases_unregister(bases: *const u64) -> c_int;
    pub fn h2_g1_sum(points: *const c_void, count: usize, out_xyz: *mut c_void) -> c_int;
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

/// Call once after `Params::new` / `Params::read` (poly/commitment.rs:56-124,256-294): later `commit*` calls whose
/// bases lie inside a registered range skip the 64 B/point upload and run over the device copy's shifted-base table
/// (include/halo2_hip.h, h2_dev_bases_precompute; built by the library on the first MSM).  Unregister before the
/// vectors are dropped.
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
