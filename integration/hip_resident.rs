//! halo2_proofs/src/hip_resident.rs -- the device-RESIDENT data path for `--features hip`.
//!
//! `hip.rs` is the literal drop-in: host slices in, host slices out, one PCIe round trip per call.  Measured on an
//! MI355X (bench.py `create_proof.host_slice_api`, the same call sequence driven from Python): a k = 22 mini-PLONK proof
//! takes 1.31 s that way (0.71 s with every vector in page-locked memory) against 0.054 s when the vectors never leave
//! the device -- 13-24x.  This file is what closes that gap on the Rust side: the three host containers the prover moves
//! between its phases become handles to device memory,
//!
//!   `Polynomial<F, B> { values: Vec<F> }`            poly.rs:33-64            ->  `DevicePolynomial<F, B>`
//!   `Params<C> { g: Vec<C>, g_lagrange: Vec<C> }`    poly/commitment.rs:23-29 ->  `DeviceParams<C>`
//!   `ProvingKey` columns                             plonk.rs:226-240         ->  `DeviceColumns` (fixed / sigma / l0 ...)
//!
//! and every operation `create_proof` performs on them (plonk/prover.rs:206-850) is one `h2_dev_*` call on the handle's
//! pointer: nothing but commitments (96 B), evaluations (32 B) and the few low coefficients SHPLONK adjusts crosses PCIe.
//! The orchestration that uses these types phase by phase is the one `halo2-gpu-specific_amd/prover.py` runs (and the one
//! whose proof bytes the test-suite pins against the CPU path); `ResidentProver` below names that sequence call by call.
//!
//! UNCOMPILED in the build image (no Rust toolchain there), like hip.rs: kept as source, applied by
//! integration/halo2_proofs_hip.patch, checked by tests/test_integration_patch.py (the patch applies; every `h2_*` name
//! declared here is an exported symbol of libhalo2_hip.so with the header's parameter count).
#![cfg(feature = "hip")]
#![allow(missing_docs)]

use crate::arithmetic::{CurveAffine, FieldExt};
use crate::hip::{check, limbs};
use crate::poly::{Basis, Coeff, ExtendedLagrangeCoeff, LagrangeCoeff, Polynomial};
use std::marker::PhantomData;
use std::os::raw::{c_int, c_void};
use std::ptr;

extern "C" {
    pub fn h2_dev_alloc(bytes: usize, d_out: *mut *mut c_void) -> c_int;
    pub fn h2_dev_free(d_ptr: *mut c_void) -> c_int;
    pub fn h2_stream_create(stream_out: *mut *mut c_void) -> c_int;
    pub fn h2_stream_destroy(stream: *mut c_void) -> c_int;
    pub fn h2_stream_synchronize(stream: *mut c_void) -> c_int;
    pub fn h2_dev_upload(d_dst: *mut c_void, src: *const c_void, bytes: usize, stream: *mut c_void) -> c_int;
    pub fn h2_dev_download(dst: *mut c_void, d_src: *const c_void, bytes: usize, stream: *mut c_void) -> c_int;

    pub fn h2_dev_ntt(d_a: *mut c_void, d_tmp: *mut c_void, omega: *const u64, log_n: u32, stream: *mut c_void) -> c_int;
    pub fn h2_dev_intt(
        d_a: *mut c_void,
        d_tmp: *mut c_void,
        omega_inv: *const u64,
        divisor: *const u64,
        log_n: u32,
        stream: *mut c_void,
    ) -> c_int;
    pub fn h2_dev_intt_batch(
        d_a: *const *mut c_void,
        count: usize,
        d_tmp: *mut c_void,
        omega_inv: *const u64,
        divisor: *const u64,
        log_n: u32,
        stream: *mut c_void,
    ) -> c_int;
    pub fn h2_dev_coeff_to_extended(
        d_coeffs: *const c_void,
        d_out: *mut c_void,
        d_tmp: *mut c_void,
        k: u32,
        extended_k: u32,
        g_coset: *const u64,
        g_coset_inv: *const u64,
        extended_omega: *const u64,
        stream: *mut c_void,
    ) -> c_int;
    pub fn h2_dev_extended_to_coeff(
        d_a: *mut c_void,
        d_tmp: *mut c_void,
        extended_k: u32,
        g_coset: *const u64,
        g_coset_inv: *const u64,
        extended_omega_inv: *const u64,
        extended_ifft_divisor: *const u64,
        stream: *mut c_void,
    ) -> c_int;
    pub fn h2_dev_divide_by_vanishing_poly(
        d_a: *mut c_void,
        size: usize,
        d_t_evaluations: *const c_void,
        t_len: usize,
        stream: *mut c_void,
    ) -> c_int;

    pub fn h2_msm_scratch_bytes(n: usize, max_bits: u32) -> usize;
    pub fn h2_msm_batch_scratch_bytes(n: usize, max_bits: u32, count: usize) -> usize;
    pub fn h2_dev_msm(
        d_scalars: *const c_void,
        d_bases: *const c_void,
        n: usize,
        max_bits: u32,
        d_scratch: *mut c_void,
        scratch_bytes: usize,
        out_xyz: *mut u64,
        stream: *mut c_void,
    ) -> c_int;
    pub fn h2_dev_msm_batch(
        d_scalars: *const *const c_void,
        count: usize,
        d_bases: *const c_void,
        n: usize,
        max_bits: u32,
        d_scratch: *mut c_void,
        scratch_bytes: usize,
        out_xyz: *mut u64,
        stream: *mut c_void,
    ) -> c_int;
    pub fn h2_dev_bases_precompute(d_bases: *const c_void, n: usize, digits: u32, stream: *mut c_void) -> c_int;
    pub fn h2_dev_bases_forget(d_bases: *const c_void) -> c_int;

    pub fn h2_dev_eval_op(
        op: c_int,
        d_res: *mut c_void,
        d_l: *const c_void,
        d_r: *const c_void,
        l_rot: i32,
        r_rot: i32,
        size: usize,
        c: *const u64,
        stream: *mut c_void,
    ) -> c_int;
    pub fn h2_dev_lincomb(
        d_res: *mut c_void,
        d_polys: *const *const c_void,
        coeffs: *const u64,
        count: usize,
        size: usize,
        stream: *mut c_void,
    ) -> c_int;
    pub fn h2_dev_eval_polynomial_batch(
        d_polys: *const *const c_void,
        count: usize,
        n: usize,
        points: *const u64,
        out: *mut u64,
        stream: *mut c_void,
    ) -> c_int;
    pub fn h2_dev_kate_division(d_a: *const c_void, n: usize, b: *const u64, d_q: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn h2_dev_batch_invert(d_a: *mut c_void, d_tmp: *mut c_void, n: usize, stream: *mut c_void) -> c_int;
    pub fn h2_dev_prefix_product(d_f: *const c_void, n: usize, init: *const u64, d_z: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn h2_dev_prefix_sum(d_f: *const c_void, n: usize, init: *const u64, d_z: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn h2_dev_batch_mont(d_a: *mut c_void, n: usize, stream: *mut c_void) -> c_int;
    pub fn h2_dev_evaluate_h(desc: *const crate::hip::H2EvalhDesc, d_values: *mut c_void, stream: *mut c_void) -> c_int;
}

/// The stream every resident operation of one proof is ordered on (the reference serialises its GPU work per device with
/// a blocking pool, arithmetic.rs:314-331; here the order is the stream's).  Create it ONCE per process and keep it: HIP hands
/// hardware queues out in stream-creation order, and a prover that makes a fresh stream per proof ends up sharing a queue --
/// and serialising -- with streams made earlier.
pub struct DeviceStream(pub *mut c_void);

impl DeviceStream {
    pub fn new() -> Self {
        let mut s = ptr::null_mut();
        check(unsafe { h2_stream_create(&mut s) }, "h2_stream_create");
        DeviceStream(s)
    }
    pub fn synchronize(&self) {
        check(unsafe { h2_stream_synchronize(self.0) }, "h2_stream_synchronize");
    }
}
impl Drop for DeviceStream {
    fn drop(&mut self) {
        unsafe { h2_stream_destroy(self.0) };
    }
}

/// `len` elements of `T` in device memory (RAII over h2_dev_alloc / h2_dev_free).
pub struct DeviceBuffer<T> {
    pub ptr: *mut c_void,
    pub len: usize,
    _marker: PhantomData<T>,
}
unsafe impl<T: Send> Send for DeviceBuffer<T> {}
unsafe impl<T: Sync> Sync for DeviceBuffer<T> {}

impl<T: Copy> DeviceBuffer<T> {
    pub fn uninit(len: usize) -> Self {
        let mut p = ptr::null_mut();
        check(unsafe { h2_dev_alloc(len * std::mem::size_of::<T>(), &mut p) }, "h2_dev_alloc");
        DeviceBuffer { ptr: p, len, _marker: PhantomData }
    }
    /// one upload; `src` may be dropped when the stream has been synchronised (page-locked sources are DMA-copied)
    pub fn from_host(src: &[T], stream: &DeviceStream) -> Self {
        let buf = Self::uninit(src.len());
        let bytes = src.len() * std::mem::size_of::<T>();
        check(unsafe { h2_dev_upload(buf.ptr, src.as_ptr() as *const c_void, bytes, stream.0) }, "h2_dev_upload");
        buf
    }
    /// the only way data comes back: synchronous
    pub fn to_host(&self, stream: &DeviceStream) -> Vec<T> {
        let mut out = Vec::<T>::with_capacity(self.len);
        let bytes = self.len * std::mem::size_of::<T>();
        check(unsafe { h2_dev_download(out.as_mut_ptr() as *mut c_void, self.ptr, bytes, stream.0) }, "h2_dev_download");
        unsafe { out.set_len(self.len) };
        out
    }
}
impl<T> Drop for DeviceBuffer<T> {
    fn drop(&mut self) {
        unsafe { h2_dev_free(self.ptr) };
    }
}

/// `Polynomial<F, B>` (poly.rs:33-64) with its values on the device.  The basis marker keeps the reference's typing:
/// a `DevicePolynomial<F, LagrangeCoeff>` can be committed against `g_lagrange`, only a `Coeff` one evaluated or opened.
pub struct DevicePolynomial<F, B> {
    pub values: DeviceBuffer<F>,
    _marker: PhantomData<B>,
}

impl<F: FieldExt, B: Basis> DevicePolynomial<F, B> {
    pub fn from_host(p: &Polynomial<F, B>, stream: &DeviceStream) -> Self {
        DevicePolynomial { values: DeviceBuffer::from_host(&p[..], stream), _marker: PhantomData }
    }
    pub fn uninit(len: usize) -> Self {
        DevicePolynomial { values: DeviceBuffer::uninit(len), _marker: PhantomData }
    }
    pub fn len(&self) -> usize {
        self.values.len
    }
    /// `self = self * c`, `self + rhs`, ... (poly.rs:191-257): the elementwise kernels, H2_OP_* of include/halo2_hip.h
    pub fn scale(&mut self, c: &F, stream: &DeviceStream) {
        let c = limbs(c);
        check(
            unsafe { h2_dev_eval_op(0, self.values.ptr, self.values.ptr, ptr::null(), 0, 0, self.len(), c.as_ptr(), stream.0) },
            "h2_dev_eval_op(MUL_C)",
        );
    }
    pub fn add_assign(&mut self, rhs: &DevicePolynomial<F, B>, stream: &DeviceStream) {
        assert_eq!(self.len(), rhs.len());
        check(
            unsafe { h2_dev_eval_op(2, self.values.ptr, self.values.ptr, rhs.values.ptr, 0, 0, self.len(), ptr::null(), stream.0) },
            "h2_dev_eval_op(SUM)",
        );
    }
}

/// the scalars of `EvaluationDomain` (poly/domain.rs:24-39) the transforms take, as Montgomery limbs
pub struct DomainScalars {
    pub k: u32,
    pub extended_k: u32,
    pub omega_inv: [u64; 4],
    pub ifft_divisor: [u64; 4],
    pub g_coset: [u64; 4],
    pub g_coset_inv: [u64; 4],
    pub extended_omega: [u64; 4],
    pub extended_omega_inv: [u64; 4],
    pub extended_ifft_divisor: [u64; 4],
}

impl<F: FieldExt> DevicePolynomial<F, LagrangeCoeff> {
    /// `EvaluationDomain::lagrange_to_coeff` (poly/domain.rs:233-266), in place on the device
    pub fn into_coeff(self, d: &DomainScalars, tmp: &DeviceBuffer<F>, stream: &DeviceStream) -> DevicePolynomial<F, Coeff> {
        check(
            unsafe { h2_dev_intt(self.values.ptr, tmp.ptr, d.omega_inv.as_ptr(), d.ifft_divisor.as_ptr(), d.k, stream.0) },
            "h2_dev_intt",
        );
        DevicePolynomial { values: self.values, _marker: PhantomData }
    }
}

impl<F: FieldExt> DevicePolynomial<F, Coeff> {
    /// `EvaluationDomain::coeff_to_extended` (poly/domain.rs:270-287): zeta powers, zero padding and the transform fused
    pub fn to_extended(&self, d: &DomainScalars, tmp: &DeviceBuffer<F>, stream: &DeviceStream) -> DevicePolynomial<F, ExtendedLagrangeCoeff> {
        let out = DevicePolynomial::<F, ExtendedLagrangeCoeff>::uninit(1usize << d.extended_k);
        check(
            unsafe {
                h2_dev_coeff_to_extended(
                    self.values.ptr,
                    out.values.ptr,
                    tmp.ptr,
                    d.k,
                    d.extended_k,
                    d.g_coset.as_ptr(),
                    d.g_coset_inv.as_ptr(),
                    d.extended_omega.as_ptr(),
                    stream.0,
                )
            },
            "h2_dev_coeff_to_extended",
        );
        out
    }
}

impl<F: FieldExt> DevicePolynomial<F, ExtendedLagrangeCoeff> {
    /// `divide_by_vanishing_poly` + `extended_to_coeff` (poly/domain.rs:328-373), in place; the caller truncates to
    /// `n * quotient_poly_degree` coefficients by taking a prefix of the buffer
    pub fn into_quotient_coeffs(
        self,
        d: &DomainScalars,
        t_evaluations: &DeviceBuffer<F>,
        tmp: &DeviceBuffer<F>,
        stream: &DeviceStream,
    ) -> DevicePolynomial<F, Coeff> {
        check(
            unsafe { h2_dev_divide_by_vanishing_poly(self.values.ptr, self.len(), t_evaluations.ptr, t_evaluations.len, stream.0) },
            "h2_dev_divide_by_vanishing_poly",
        );
        check(
            unsafe {
                h2_dev_extended_to_coeff(
                    self.values.ptr,
                    tmp.ptr,
                    d.extended_k,
                    d.g_coset.as_ptr(),
                    d.g_coset_inv.as_ptr(),
                    d.extended_omega_inv.as_ptr(),
                    d.extended_ifft_divisor.as_ptr(),
                    stream.0,
                )
            },
            "h2_dev_extended_to_coeff",
        );
        DevicePolynomial { values: self.values, _marker: PhantomData }
    }
}

/// `Params<C>` (poly/commitment.rs:23-29) with both base tables uploaded ONCE and expanded into shifted-base tables
/// (h2_dev_bases_precompute: every later commitment adds all digits of a scalar into one bucket set); the reference
/// re-uploads `g` / `g_lagrange` on every MSM (arithmetic.rs:354-360).
pub struct DeviceParams<C: CurveAffine> {
    pub k: u32,
    pub n: usize,
    pub g: DeviceBuffer<C>,
    pub g_lagrange: DeviceBuffer<C>,
    scratch: DeviceBuffer<u8>,
}

impl<C: CurveAffine> DeviceParams<C> {
    pub fn from_params(params: &crate::poly::commitment::Params<C>, stream: &DeviceStream) -> Self {
        let n = params.g.len();
        let g = DeviceBuffer::from_host(&params.g[..], stream);
        let g_lagrange = DeviceBuffer::from_host(&params.g_lagrange[..], stream);
        for t in [&g_lagrange, &g] {
            // digits = 0: the library's choice for n; skipped by the library below 2^15 points
            check(unsafe { h2_dev_bases_precompute(t.ptr, n, 0, stream.0) }, "h2_dev_bases_precompute");
        }
        let scratch = DeviceBuffer::uninit(unsafe { h2_msm_batch_scratch_bytes(n, 254, 8) });
        stream.synchronize();
        DeviceParams { k: params.k, n, g, g_lagrange, scratch }
    }

    fn msm<B>(&self, bases: &DeviceBuffer<C>, polys: &[&DevicePolynomial<C::Scalar, B>], max_bits: u32, stream: &DeviceStream) -> Vec<C::Curve> {
        let ptrs: Vec<*const c_void> = polys.iter().map(|p| p.values.ptr as *const c_void).collect();
        let mut out = vec![[0u64; 12]; polys.len()];
        check(
            unsafe {
                h2_dev_msm_batch(
                    ptrs.as_ptr(),
                    ptrs.len(),
                    bases.ptr,
                    polys[0].len(),
                    max_bits,
                    self.scratch.ptr,
                    self.scratch.len,
                    out.as_mut_ptr() as *mut u64,
                    stream.0,
                )
            },
            "h2_dev_msm_batch",
        );
        // the same transmute the reference's cuda path uses for its result (arithmetic.rs:364-365)
        out.iter().map(|xyz| unsafe { std::mem::transmute_copy::<[u64; 12], C::Curve>(xyz) }).collect()
    }

    /// `Params::commit_lagrange` (poly/commitment.rs:136-142) for a batch of columns: one pipelined / fused call
    pub fn commit_lagrange(&self, polys: &[&DevicePolynomial<C::Scalar, LagrangeCoeff>], max_bits: u32, stream: &DeviceStream) -> Vec<C::Curve> {
        self.msm(&self.g_lagrange, polys, max_bits, stream)
    }
    /// `Params::commit` (poly/commitment.rs:129-134)
    pub fn commit(&self, polys: &[&DevicePolynomial<C::Scalar, Coeff>], stream: &DeviceStream) -> Vec<C::Curve> {
        self.msm(&self.g, polys, 254, stream)
    }
}
impl<C: CurveAffine> Drop for DeviceParams<C> {
    fn drop(&mut self) {
        unsafe {
            h2_dev_bases_forget(self.g.ptr);
            h2_dev_bases_forget(self.g_lagrange.ptr);
        }
    }
}

/// What `ProvingKey` (plonk.rs:226-240) and `permutation::ProvingKey` (permutation/keygen.rs:248-259) hold, resident:
/// made once at keygen, read by every proof.
pub struct DeviceColumns<F> {
    pub fixed_values: Vec<DevicePolynomial<F, LagrangeCoeff>>,
    pub fixed_polys: Vec<DevicePolynomial<F, Coeff>>,
    pub fixed_cosets: Vec<DevicePolynomial<F, ExtendedLagrangeCoeff>>,
    pub sigma_values: Vec<DevicePolynomial<F, LagrangeCoeff>>,
    pub sigma_polys: Vec<DevicePolynomial<F, Coeff>>,
    pub sigma_cosets: Vec<DevicePolynomial<F, ExtendedLagrangeCoeff>>,
    pub l0: DevicePolynomial<F, ExtendedLagrangeCoeff>,
    pub l_last: DevicePolynomial<F, ExtendedLagrangeCoeff>,
    pub l_active_row: DevicePolynomial<F, ExtendedLagrangeCoeff>,
    pub t_evaluations: DeviceBuffer<F>,
}

/// The order of a resident `create_proof` (plonk/prover.rs:206-850), one line per phase and the entry points it uses --
/// the sequence halo2-gpu-specific_amd/prover.py::create_proof_ext executes (k = 22: 54 ms, k = 24: 0.19 s on one MI355X):
///
///  1. advice columns: `DeviceBuffer::from_host` per column (page-locked sources: DMA), blinding rows written on the
///     device, `h2_dev_max_scalar_bits` + `h2_dev_batch_mont`, `DeviceParams::commit_lagrange` in groups      (:255-312)
///  2. theta; lookups: `h2_dev_evaluate_h` with y := theta (evaluate_with_theta), `h2_dev_logup_multiplicity`,
///     commit m                                                                                             (logup/prover.rs:63-240)
///  3. beta, gamma; `h2_dev_permutation_terms`, ONE `h2_dev_batch_invert` for every denominator of the proof,
///     `h2_dev_prefix_product` / `h2_dev_prefix_sum`, commit z, `h2_dev_intt_batch`                         (permutation/prover.rs:47-165)
///  4. y; `DevicePolynomial::to_extended` for advice / z / m, `h2_dev_evaluate_h` (H2EvalhDesc with DEVICE column
///     pointers), `into_quotient_coeffs`, commit the h pieces                                               (vanishing/prover.rs:69-112)
///  5. x; `h2_dev_eval_polynomial_batch` for every (polynomial, rotation) of the proof in one launch         (:700-790)
///  6. multiopen: `h2_dev_lincomb` per rotation set, `h2_dev_kate_division`, commit                          (shplonk/prover.rs:89-225)
///
/// Only steps 1 (the witness) and the 96-byte / 32-byte results of the commit / evaluate calls touch PCIe.
pub struct ResidentProver<C: CurveAffine> {
    pub stream: DeviceStream,
    pub params: DeviceParams<C>,
    pub columns: DeviceColumns<C::Scalar>,
    pub domain: DomainScalars,
}

impl<C: CurveAffine> ResidentProver<C> {
    /// step 3's tail for a batch of product columns: commit, then coefficient forms in place (commit_lagrange_and_ifft,
    /// poly/commitment.rs:144-197, for every z of a proof at once)
    pub fn commit_lagrange_and_ifft(
        &self,
        zs: Vec<DevicePolynomial<C::Scalar, LagrangeCoeff>>,
    ) -> (Vec<DevicePolynomial<C::Scalar, Coeff>>, Vec<C::Curve>) {
        let refs: Vec<&DevicePolynomial<C::Scalar, LagrangeCoeff>> = zs.iter().collect();
        let points = self.params.commit_lagrange(&refs, 254, &self.stream);
        let n = 1usize << self.domain.k;
        let tmp = DeviceBuffer::<C::Scalar>::uninit(n * zs.len().min(16));
        let ptrs: Vec<*mut c_void> = zs.iter().map(|z| z.values.ptr).collect();
        check(
            unsafe {
                h2_dev_intt_batch(
                    ptrs.as_ptr(),
                    ptrs.len(),
                    tmp.ptr,
                    self.domain.omega_inv.as_ptr(),
                    self.domain.ifft_divisor.as_ptr(),
                    self.domain.k,
                    self.stream.0,
                )
            },
            "h2_dev_intt_batch",
        );
        let polys = zs.into_iter().map(|z| DevicePolynomial { values: z.values, _marker: PhantomData }).collect();
        (polys, points)
    }

    /// step 5: every evaluation of the proof in one launch, one read-back (the par_iter of plonk/prover.rs:731-737)
    pub fn evaluate(&self, polys: &[&DevicePolynomial<C::Scalar, Coeff>], points: &[C::Scalar]) -> Vec<C::Scalar> {
        assert_eq!(polys.len(), points.len());
        let ptrs: Vec<*const c_void> = polys.iter().map(|p| p.values.ptr as *const c_void).collect();
        let pts: Vec<[u64; 4]> = points.iter().map(limbs).collect();
        let mut out = vec![[0u64; 4]; polys.len()];
        check(
            unsafe {
                h2_dev_eval_polynomial_batch(
                    ptrs.as_ptr(),
                    ptrs.len(),
                    1usize << self.domain.k,
                    pts.as_ptr() as *const u64,
                    out.as_mut_ptr() as *mut u64,
                    self.stream.0,
                )
            },
            "h2_dev_eval_polynomial_batch",
        );
        out.iter().map(|l| unsafe { std::mem::transmute_copy::<[u64; 4], C::Scalar>(l) }).collect()
    }
}
