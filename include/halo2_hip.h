/*
 * halo2_hip.h -- C ABI of libhalo2_hip.so: the MI355X (gfx950) drop-in for the `cuda`
 * feature of halo2_proofs (DelphinusLab/halo2-gpu-specific).
 *
 * Every entry point replaces one `#[cfg(feature = "cuda")]` call site of the reference
 * (file:line under /root/reference/halo2_proofs/src/, cited per function).  The Rust side
 * keeps its generic signatures and forwards the transmuted slices (exactly what it hands to
 * ec-gpu-gen today, arithmetic.rs:156-161,351-352,391-394,507-508); see INTEGRATION.md.
 *
 * Conventions
 *   Fr            32 B = 4 x u64 little-endian limbs, Montgomery form (R = 2^256)
 *   G1Affine      64 B = {x: Fq, y: Fq} Montgomery, identity = (0, 0)
 *   G1 (result)   96 B = Jacobian {x, y, z: Fq} Montgomery, identity has z = 0
 *   value range   every field element handed to the library is a valid residue image: below the modulus (the
 *                 reference's own Fr / Fq invariant).  The multiplier relies on inputs below 2^254 (it skips the
 *                 carry handling that such inputs cannot trigger); out-of-range limbs give unspecified residues, not
 *                 an error.
 *   return value  0 = H2_OK, non-zero = error; h2_last_error() gives the text.  The
 *                 reference unwrap()s its GPU Results (arithmetic.rs:358,360,509), so the
 *                 Rust shim panics on non-zero.
 *   threading     every function may be called concurrently from any thread (rayon workers);
 *                 host-buffer calls take a device from the blocking pool sized by
 *                 HALO2_PROOFS_N_GPU (plonk/prover.rs:56-74; arithmetic.rs:314-331).  A device serves
 *                 H2_HOST_SLOTS (default 2, 1..4) such calls at a time, each on its own stream and staging, so that
 *                 one caller's transfers overlap another's kernels; 1 = the reference's one operation per device.
 *   h2_dev_*      operate on HIP device pointers on the caller's stream (void* = hipStream_t,
 *                 NULL = the library's stream for the current device) and do not synchronise.
 *                 NULL does NOT mean HIP's legacy default stream: the library's stream is non-blocking, so work
 *                 the caller queued on stream 0 is not ordered against it -- pass a stream of your own (every
 *                 call then runs in its order), or bracket the calls with h2_synchronize / hipDeviceSynchronize.
 */
#ifndef HALO2_HIP_H
#define HALO2_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    H2_OK = 0,
    H2_ERR_INVALID = 1,
    H2_ERR_NO_DEVICE = 2,
    H2_ERR_HIP = 3,
    H2_ERR_OOM = 4
};

/* ---- library / device pool ------------------------------------------------------------ */
int h2_version(void);
/* Device::all().len() clipped by HALO2_PROOFS_N_GPU -- plonk/prover.rs:57-70 */
int h2_device_count(void);
const char *h2_last_error(void);
/* blocks until all work queued by this library on every pooled device has finished */
int h2_synchronize(void);

/* ---- device memory the library keeps between calls ---------------------------------------------
 * The reference bounds its device memory by keeping coefficient forms and re-deriving extended cosets through a small
 * cache (plonk/evaluation_gpu.rs:335-468, HALO2_PROOF_GPU_EVAL_CACHE).  Here the caller owns every polynomial; what the
 * LIBRARY holds per device is: NTT plans (twiddle tables, a few MiB per (log_n, omega)), the optional complete last-pass
 * twiddle tables (32 B x n per (plan, divisor): 512 MiB at 2^24), shifted-base tables (h2_dev_bases_precompute), device
 * copies of registered SRS ranges and the host-API staging arenas.
 *  - The last-pass tables live inside a per-device budget: H2_NTT_TABLE_BUDGET in the environment (bytes, K / M / G
 *    suffixes) or h2_set_table_budget; default 1/32 of the device's memory.  When a new table would exceed it, idle
 *    tables leave in least-recently-used order; a transform without a table composes its twiddles (same values).
 *  - h2_release_plans frees every plan (and its tables) that no call is using, on every device: for callers that
 *    cycle through domains or coset generators.  Synchronises the devices.  It also gives back every slot's block of
 *    column vectors of h2_evaluate_h_coeff / h2_quotient_poly_coeff ((2 x distinct columns + a few) x 2^k x 32 B plus two
 *    extended vectors: 6.4 GiB for 16 columns at k = 22; kept between calls because allocating it cost up to 160 ms).
 *  - h2_library_memory_bytes: what the library holds on the current device right now. */
int h2_release_plans(void);
int h2_set_table_budget(size_t bytes);
size_t h2_library_memory_bytes(void);

/* ---- NTT (host buffers) ---------------------------------------------------------------- */
/* best_fft -> gpu_fft: arithmetic.rs:546-554, :495-512.  a: 2^log_n Fr, in place. */
int h2_ntt(uint64_t *a, const uint64_t omega[4], uint32_t log_n);
/* gpu_ifft: arithmetic.rs:515-534 (EvaluationDomain::ifft, poly/domain.rs:400-414).
 * a <- NTT(a, omega_inv) * divisor. */
int h2_intt(uint64_t *a, const uint64_t omega_inv[4], const uint64_t divisor[4], uint32_t log_n);
/* the same out of place: `a` (2^log_n values the caller keeps: the reference clones the column first, plonk/prover.rs:643-646)
 * is only read, the coefficients go to `out`. */
int h2_intt_to(const uint64_t *a, uint64_t *out, const uint64_t omega_inv[4], const uint64_t divisor[4], uint32_t log_n);
/* EvaluationDomain::coeff_to_extended: poly/domain.rs:270-287 (+ distribute_powers_zeta :382-398).
 * coeffs: 2^k Fr (read only); out: 2^extended_k Fr. */
int h2_coeff_to_extended(const uint64_t *coeffs, uint64_t *out, uint32_t k, uint32_t extended_k,
                         const uint64_t g_coset[4], const uint64_t g_coset_inv[4],
                         const uint64_t extended_omega[4]);
/* EvaluationDomain::extended_to_coeff: poly/domain.rs:328-350.  a: 2^extended_k Fr (read only);
 * out: out_len = n * quotient_poly_degree Fr (the truncation of :346-347). */
int h2_extended_to_coeff(const uint64_t *a, uint64_t *out, size_t out_len, uint32_t extended_k,
                         const uint64_t g_coset[4], const uint64_t g_coset_inv[4],
                         const uint64_t extended_omega_inv[4], const uint64_t extended_ifft_divisor[4]);

/* ---- MSM (host buffers) ---------------------------------------------------------------- */
/* gpu_multiexp_single_gpu_with_bound: arithmetic.rs:334-367 (kern.multiexp_bound :360).
 * Only the low max_bits bits of each canonical scalar are used (callers guarantee the rest
 * are zero: plonk/prover.rs:237-254).  max_bits == 0 or n == 0 -> identity (:346, :421). */
int h2_msm(const uint64_t *scalars, const uint64_t *bases, size_t n, uint32_t max_bits, uint64_t out_xyz[12]);
/* gpu_multiexp_bound: arithmetic.rs:413-440 -- contiguous ceil(n/N_GPU) chunks, one pooled
 * device each, partial points summed on the host. */
int h2_msm_multi(const uint64_t *scalars, const uint64_t *bases, size_t n, uint32_t max_bits, uint64_t out_xyz[12]);
/* Resident SRS (an improvement the reference lacks: it re-uploads the bases on every call,
 * arithmetic.rs:354-360).  After h2_bases_register(g, n) every host-buffer MSM whose `bases` range lies
 * inside [g, g + n) uses a device copy uploaded once per device (Params::g / g_lagrange never change,
 * poly/commitment.rs:23-29).  The caller must not modify a registered range before unregistering it.
 * From 2^15 points on the device copy also gets a shifted-base table (see h2_dev_bases_precompute: digits x n x 64 B
 * more device memory, built on first use, freed with the copy) when that fits in half of the free device memory;
 * H2_MSM_TABLES=0 in the environment keeps the copy only. */
int h2_bases_register(const uint64_t *bases, size_t n);
int h2_bases_unregister(const uint64_t *bases);
/* Resident polynomials, the same idea for Fr vectors: the proving key's coefficient forms -- fixed_polys, permutation.polys, l0,
 * l_last (plonk.rs:226-240, made once by keygen_pk, plonk/keygen.rs:330-440) -- are read by EVERY proof: by the evaluator
 * (plonk/evaluation.rs:1229-1241), by the evaluations at x (plonk/prover.rs:731-737) and by the multiopen folds
 * (poly/multiopen/shplonk/prover.rs:110-209, gwc/prover.rs:57-151), and the reference's cuda path uploads them each time.
 * After h2_poly_register(values, n) every host-slice entry point that only READS a vector lying inside [values, values + n)
 * -- h2_evaluate_h_coeff's columns, h2_eval_polynomial, h2_lincomb's operands, h2_kate_division's dividend, h2_eval_op's
 * operands, h2_coeff_to_extended's input -- uses a device copy uploaded once per device (on first use).  The caller must not
 * modify a registered range before h2_poly_unregister, which frees the copies on every device; a stale registration is never
 * served: register / unregister bump a generation and a copy made for another generation (or length) is dropped.  A host may
 * also register a proof's own final polynomials (advice / product coefficient forms) for the rest of that proof.
 * Entry points that WRITE through the pointer (h2_ntt, h2_intt, h2_batch_mont, h2_eval_op's result ...) never consult the
 * registry for it. */
int h2_poly_register(const uint64_t *values, size_t n);
int h2_poly_unregister(const uint64_t *values);
/* Host-side fold of `count` partial results (12 x u64 Jacobian each): the
 * `.reduce(|acc, x| acc + x)` of arithmetic.rs:433-435; also used after an all-gather of per-rank
 * partial points when one MSM is split across processes (one process per GPU). */
int h2_g1_sum(const uint64_t *points_xyz, size_t count, uint64_t out_xyz[12]);
/* The same fold on the device, for the partial points of `count` range-split MSMs gathered from `world` ranks
 * (d_points_xyz[(r * count + j) * 12 ..] = rank r's Jacobian partial of MSM j): d_out_xyz[j] = sum over r in rank order.
 * Keeps the multi-GPU exchange on the device: all-gather (RCCL) -> this kernel -> one read-back of count x 96 B. */
int h2_dev_g1_fold(const void *d_points_xyz, uint32_t world, uint32_t count, void *d_out_xyz, void *stream);
/* gpu_multiexp_bound_and_fft: arithmetic.rs:375-410 (Params::commit_lagrange_and_ifft,
 * poly/commitment.rs:148-170): MSM over `bases` and, sharing the one upload, scalars <-
 * NTT(scalars, omega_inv) * divisor. */
int h2_msm_intt(uint64_t *scalars, const uint64_t *bases, size_t n, uint32_t max_bits,
                const uint64_t omega_inv[4], const uint64_t divisor[4], uint32_t log_n, uint64_t out_xyz[12]);

/* ---- Montgomery conversion (host buffers) ------------------------------------------------ */
/* gpu_mont / gpu_unmont: arithmetic.rs:263-306, :218-261 (kernels batch_mont / batch_unmont).
 * Input ranges: the conversions (and the scalars of every MSM, which are taken out of Montgomery form first) accept ANY
 * 256-bit value and return / use its residue; every other entry point takes field elements as the reference holds them
 * (canonical residues below the modulus, < 2^254: the multiplier's carry analysis relies on it). */
int h2_batch_mont(uint64_t *a, size_t n);
int h2_batch_unmont(uint64_t *a, size_t n);

/* ---- elementwise polynomial kernels (host buffers) ---------------- */
enum {
    H2_OP_MUL_C = 0,    /* res[i] = l[i+l_rot] * c          eval_mul_c   evaluation_gpu.rs:669,1034 */
    H2_OP_SUM_C = 1,    /* res[i] = l[i+l_rot] + c          eval_sum_c   evaluation_gpu.rs:560,668  */
    H2_OP_SUM = 2,      /* res[i] = l[i+l_rot] + r[i+r_rot] eval_sum     evaluation_gpu.rs:279-305  */
    H2_OP_MUL = 3,      /* res[i] = l[i+l_rot] * r[i+r_rot] eval_mul     evaluation_gpu.rs:622-648  */
    H2_OP_SUB = 4,      /* res[i] = l[i+l_rot] - r[i+r_rot] Polynomial - poly.rs:205-217            */
    H2_OP_LCTHETA = 5,  /* res = l*c + r                    eval_lctheta evaluation_gpu.rs:148-163  */
    H2_OP_LCBETA = 6,   /* res = (l + c) * r                eval_lcbeta  evaluation_gpu.rs:202-217  */
    H2_OP_ADDGAMMA = 7, /* res = l + c                      eval_addgamma evaluation_gpu.rs:246-259 */
    H2_OP_CONSTANT = 8  /* res = c                          eval_constant evaluation_gpu.rs:579-585 */
};
/* Rotations are signed element offsets taken modulo size (evaluation.rs:40-42).  res may alias
 * l or r when that operand's rotation is 0 (evaluation_gpu.rs:631-639).  Unused operands NULL. */
int h2_eval_op(int op, uint64_t *res, const uint64_t *l, const uint64_t *r, int32_t l_rot, int32_t r_rot,
               size_t size, const uint64_t c[4]);
/* EvaluationDomain::divide_by_vanishing_poly: poly/domain.rs:354-373: a[i] *= t_evals[i % t_len] */
int h2_divide_by_vanishing_poly(uint64_t *a, size_t size, const uint64_t *t_evaluations, size_t t_len);


/* ---- adjacent numerics the prover runs between the transforms ------------ */
/* eval_polynomial: arithmetic.rs:714-735 (Horner; prover.rs:731-737 evaluates every committed polynomial
 * at x).  out = sum_i poly[i] * point^i.  Synchronous (the result is host memory). */
int h2_eval_polynomial(const uint64_t *poly, size_t n, const uint64_t point[4], uint64_t out[4]);
/* `count` evaluations in one call, polynomial j (n coefficients) at points[4j..4j+4] -> out[4j..4j+4]: what
 * plonk/prover.rs:700-790 computes in a rayon par_iter over eval_polynomial_st.  One launch per fold level over all of them,
 * one copy back; registered polynomials are read on the device, the others go up once each. */
int h2_eval_polynomial_batch(const uint64_t *const *polys, size_t count, size_t n, const uint64_t *points, uint64_t *out);
int h2_dev_eval_polynomial(const void *d_poly, size_t n, const uint64_t point[4], uint64_t out[4], void *stream);
/* `count` evaluations, polynomial j (n coefficients, device) at points[4 j .. 4 j + 3], enqueued back to back with one
 * read-back: the rayon `par_iter` over eval_polynomial_st of plonk/prover.rs:731-737.  d_polys: HOST array of device
 * pointers; out: count x 4 u64 on the host.  Synchronous. */
int h2_dev_eval_polynomial_batch(const void *const *d_polys, size_t count, size_t n, const uint64_t *points,
                                 uint64_t *out, void *stream);
/* batch_invert: arithmetic.rs:840-844 -- every non-zero element replaced by its inverse, zeros kept
 * (ff::BatchInvert).  d_tmp: n Fr of scratch. */
int h2_batch_invert(uint64_t *a, size_t n);
int h2_dev_batch_invert(void *d_a, void *d_tmp, size_t n, void *stream);
/* kate_division: arithmetic.rs:754-773 -- q(X) = a(X) / (X - b) without remainder; a: n coefficients,
 * q: n - 1 (multiopen: gwc/prover.rs:154-160, shplonk/prover.rs).  A blocked affine prefix scan. */
int h2_kate_division(const uint64_t *a, size_t n, const uint64_t b[4], uint64_t *q);
int h2_dev_kate_division(const void *d_a, size_t n, const uint64_t b[4], void *d_q, void *stream);
/* Grand-product column: z[0] = init, z[i] = z[i-1] * f[i-1] for i < n (f has n - 1 used entries) -- the
 * serial loops of permutation/prover.rs:151-160 (`z.push(last_z); for row in 1..n { tmp *= modified_values[row-1] }`),
 * shuffle/prover.rs and the logup grand sums' multiplicative twin.  f and z must not alias. */
int h2_prefix_product(const uint64_t *f, size_t n, const uint64_t init[4], uint64_t *z);
int h2_dev_prefix_product(const void *d_f, size_t n, const uint64_t init[4], void *d_z, void *stream);
/* res[i] = sum_j coeffs[j] * polys[j][i] -- the GWC / SHPLONK batching loops (poly/multiopen/gwc/prover.rs:39-151:
 * `poly_batch = poly_batch * v + poly`, cuda branch eval_mul_c + eval_sum per polynomial; shplonk/prover.rs:110-209)
 * with the challenge powers supplied by the caller.  d_polys: HOST array of `count` device pointers; coeffs:
 * count x 4 u64 on the host.  d_res may alias d_polys[0] only. */
int h2_dev_lincomb(void *d_res, const void *const *d_polys, const uint64_t *coeffs, size_t count, size_t size,
                   void *stream);
/* the same on host buffers (the GWC cuda branch's shape, gwc/prover.rs:57-151: every p_i is uploaded for its
 * eval_mul_c / eval_sum pair): polys = host array of `count` host pointers; one upload per operand, one fused pass. */
int h2_lincomb(uint64_t *res, const uint64_t *const *polys, const uint64_t *coeffs, size_t count, size_t size);

/* The quotient contributions of a multi-point opening in one call (poly/multiopen/shplonk/prover.rs:95-153: every rotation
 * set's linear combination minus its low-degree remainder polynomial, divided by the set's vanishing polynomial, the sets
 * folded by powers of v -- and :205-219 with n_sets = 1: the final quotient l(X) / (X - u)):
 *   out = sum_s (sum_i coeffs[s][i] * polys[s][i](X) - low_s(X)) / prod_j (X - points[s][j])
 * counts / low_counts / point_counts: per set; polys (host pointers to n coefficients each), coeffs, low, points: the sets'
 * entries one after the other.  The caller multiplies v^(R-1-s) into set s's coeffs and low (division is linear).  Operands
 * registered with h2_poly_register are read on the device; everything else stays there too: `out` (n coefficients, the top
 * ones zero as the reference resizes them, :118) crosses PCIe once.  remainders (may be NULL): 4 words per point, what each
 * division left (the value of its dividend at the point: the reference's must_be_zero, :213-214). */
int h2_quotient_sum(uint64_t *out, size_t n, size_t n_sets, const size_t *counts, const uint64_t *const *polys,
                    const uint64_t *coeffs, const size_t *low_counts, const uint64_t *low, const size_t *point_counts,
                    const uint64_t *points, uint64_t *remainders);

/* Permutation argument, the elementwise work on either side of the grand-product scan.
 * keygen (plonk/permutation/keygen.rs:197-238): out[j] = DELTA^{map_col[j]} * omega^{map_row[j]} -- one sigma column
 * in Lagrange form from the cycle mapping (u32 device arrays of n entries). */
int h2_dev_permutation_sigma(void *d_out, const void *d_map_col, const void *d_map_row, size_t n,
                             const uint64_t delta[4], const uint64_t omega[4], void *stream);
/* prover (plonk/permutation/prover.rs:89-128), one call per column of a set:
 *   den[i] (*)= beta * sigma[i] + gamma + value[i];   num[i] (*)= delta_pow * omega^i * beta + gamma + value[i]
 * first != 0 overwrites num / den, otherwise multiplies into them.  delta_pow = DELTA^{column position}.
 * The caller follows with h2_dev_batch_invert(den), an elementwise product and h2_dev_prefix_product. */
/* the same on host buffers (the reference leaves these products to a rayon loop between its GPU calls): value / sigma in --
 * sigma, a proving-key column, is read on the device when registered with h2_poly_register -- num / den out; chunk-pipelined */
int h2_permutation_terms(uint64_t *num, uint64_t *den, const uint64_t *value, const uint64_t *sigma, size_t n,
                         const uint64_t beta[4], const uint64_t gamma[4], const uint64_t delta_pow[4],
                         const uint64_t omega[4], int first);
/* One grand-product column of the permutation argument, whole, for one set of `count` columns (permutation/prover.rs:72-165:
 * the products over the set's columns :92-127, the batch inversion :106, the running product :150-154):
 *   z[0] = init;  z[i+1] = z[i] * prod_j (values[j][i] + beta * delta_pow * delta^j * omega^i + gamma)
 *                                / prod_j (values[j][i] + beta * sigmas[j][i] + gamma),  i + 1 < n
 * values / sigmas: `count` host pointers to n-element vectors (read on the device without an upload when registered with
 * h2_poly_register); delta_pow = DELTA^(index of the set's first column).  Every intermediate stays on the device: the
 * columns cross PCIe once, z once.  The caller then overwrites its blinding rows and takes z[n - (blinding_factors + 1)] as
 * the next set's init (:157-162). */
int h2_permutation_product(uint64_t *z, const uint64_t *const *values, const uint64_t *const *sigmas, size_t count, size_t n,
                           const uint64_t beta[4], const uint64_t gamma[4], const uint64_t delta_pow[4],
                           const uint64_t delta[4], const uint64_t omega[4], const uint64_t init[4]);
int h2_dev_permutation_terms(void *d_num, void *d_den, const void *d_value, const void *d_sigma, size_t n,
                             const uint64_t beta[4], const uint64_t gamma[4], const uint64_t delta_pow[4],
                             const uint64_t omega[4], int first, void *stream);

/* The vanishing argument's blinding polynomial (plonk/vanishing/prover.rs:47-61, a parallel fill of `Scalar::random`
 * draws from thread_rng): element i = ChaCha20 block i under the caller's 256-bit key (counter i, zero nonce); the
 * 64 output bytes are cut into two 253-bit little-endian integers lo, hi and the element is lo + 2^253 hi mod r
 * (506 random bits), stored in Montgomery form.  The key MUST come from a cryptographic source in production
 * (halo2-gpu-specific_amd/rng.py draws it from os.urandom; its seeded mode is for tests); rng.py also holds the host
 * twin the reference prover of the tests draws from. */
int h2_dev_random_fr(const uint8_t key[32], size_t n, void *d_out, void *stream);
/* the same stream of elements into a host vector (one copy down) */
int h2_random_fr(const uint8_t key[32], size_t n, uint64_t *out);

/* a[i] *= g^i for i < n (n <= 2^28): distribute_powers_zeta (poly/domain.rs:382-398) for an arbitrary generator.  With
 * g = zeta * extended_omega^j followed by an n-point h2_dev_ntt it evaluates a coefficient vector on coset j of the
 * extended domain (extended index c i + j, c = 2^(extended_k - k)); with g^-1 after an n-point h2_dev_intt it inverts
 * that.  The multi-GPU proof shards the extended domain by such cosets (one proof over several ranks). */
int h2_dev_distribute_powers(void *d_a, size_t n, const uint64_t g[4], void *stream);

/* Grand-sum column: z[0] = init, z[i] = z[i-1] + f[i-1] for i < n -- the `scan` of plonk/logup/prover.rs:353-367. */
int h2_dev_prefix_sum(const void *d_f, size_t n, const uint64_t init[4], void *d_z, void *stream);
/* The multiplicity column of a logup lookup (plonk/logup/prover.rs:104-180): for every row r < usable_rows of each
 * compressed input column, the table row holding that value is credited once; m[row] = credit as a field element
 * (Montgomery), m[row >= usable_rows] = 0 (the caller writes the blinding values there, :217-221).  A duplicated table
 * value collects its credits on its lowest row (the reference's binary search lands on an implementation-defined
 * duplicate; the argument is indifferent).  d_inputs: HOST array of n_inputs device pointers.  Returns H2_ERR_INVALID
 * when an input value is absent from the table (the reference panics).  Synchronous. */
size_t h2_logup_scratch_bytes(size_t n);
/* Host-vector twins (the reference runs these steps as host loops between its GPU calls): the grand-sum scan, a[i] *= g^i,
 * one sigma column from the cycle mapping (u32 host arrays of n entries), the multiplicity column of a lookup (table and
 * inputs: host vectors of n elements, read on the device when registered with h2_poly_register; max_bits_out may be NULL). */
/* One grand-sum column of a logup lookup, whole, for one set of `count` compressed input columns (plonk/logup/prover.rs:243-347:
 * beta + f per input :259-272 / :310-324, batch_invert :266 / :318, the table's term m / (beta + t) :275-290, the scan :330-346):
 *   z[0] = init;  z[i+1] = z[i] + sum_j 1 / (beta + inputs[j][i]) - m[i] / (beta + table[i]),  i + 1 < n
 * table and m: both NULL for a set without the table (the extra input sets), both given for the first set.  Every intermediate
 * stays on the device; registered vectors are read there.  The caller writes its blinding rows into z and carries
 * z[n - (blinding_factors + 1)] into the next set's init (:333-345). */
int h2_logup_grand_sum(uint64_t *z, const uint64_t *const *inputs, size_t count, const uint64_t *table, const uint64_t *m,
                       size_t n, const uint64_t beta[4], const uint64_t init[4]);
int h2_prefix_sum(const uint64_t *f, size_t n, const uint64_t init[4], uint64_t *z);
int h2_distribute_powers(uint64_t *a, size_t n, const uint64_t g[4]);
int h2_permutation_sigma(uint64_t *out, const uint32_t *map_col, const uint32_t *map_row, size_t n,
                         const uint64_t delta[4], const uint64_t omega[4]);
int h2_logup_multiplicity(const uint64_t *table, const uint64_t *const *inputs, size_t n_inputs, size_t usable_rows,
                          size_t n, uint64_t *m, uint32_t *max_bits_out);
int h2_dev_logup_multiplicity(const void *d_table, const void *const *d_inputs, size_t n_inputs, size_t usable_rows,
                              size_t n, void *d_m, void *d_scratch, size_t scratch_bytes, void *stream);
/* h2_dev_logup_multiplicity that also returns the bit length of the LARGEST multiplicity (0 when every count is zero): the
 * bound for m's commitment (find_max_scalar_bits of the usable rows, arithmetic.rs's `max_bits`) without another pass --
 * a range lookup's multiplicities are a few bits wide, not the log2(rows x inputs) their sum allows */
int h2_dev_logup_multiplicity_bits(const void *d_table, const void *const *d_inputs, size_t n_inputs, size_t usable_rows,
                                   size_t n, void *d_m, void *d_scratch, size_t scratch_bytes, uint32_t *max_bits_out,
                                   void *stream);
/* The same counting restricted to the input rows [row_begin, row_end) -- one device's share when the rows of a proof are dealt
 * over several devices -- with the RAW counters out: d_counts = n + 1 u32 (credits per table row, then the number of input
 * values missing from the table).  Integer counts are an RCCL reduction (sum), field elements are not: the ranks all-reduce
 * d_counts, check the last word and turn the first n into the field elements of m(X) with h2_dev_logup_emit (rows >=
 * usable_rows zero).  Asynchronous on `stream`; same scratch as above. */
int h2_dev_logup_counts(const void *d_table, const void *const *d_inputs, size_t n_inputs, size_t usable_rows, size_t n,
                        size_t row_begin, size_t row_end, void *d_counts, void *d_scratch, size_t scratch_bytes, void *stream);
int h2_dev_logup_emit(const void *d_counts, size_t usable_rows, size_t n, void *d_m, void *stream);

/* Fixed-base multiplication, the work of Params::unsafe_setup (poly/commitment.rs:56-124: g[i] = [s^i] G,
 * g_lagrange[i] = [l_i(s)] G, one variable-base multiplication per point under `parallelize` there):
 * points[i] = [scalars[i]] B, with B given as d_table[j] = [2^j] B for j < 254 (affine Montgomery, 64 B each);
 * scalars Montgomery Fr, output affine Montgomery ((0,0) for a zero scalar).  Asynchronous on `stream`. */
int h2_dev_fixed_base_mul(const void *d_scalars, const void *d_table, size_t n, void *d_points, void *stream);

/* Compressed G1 points of the SRS file -- Params::{write, read}, poly/commitment.rs:241-294 (`to_bytes` / `from_bytes`
 * per point; the reference decompresses with a rayon `parallelize`).  32 bytes per point: x little-endian, bit 7 of
 * byte 31 = parity of the canonical y, identity = zeros (convention of this build: the encoding of pairing_bn256@30b052f
 * could not be checked).  d_points: n x 64 B affine Montgomery.  Decompress is synchronous and returns H2_ERR_INVALID
 * when an encoding is not a curve point (the reference unwraps `from_bytes`). */
int h2_dev_points_decompress(const void *d_bytes, size_t n, void *d_points, void *stream);
int h2_dev_points_compress(const void *d_points, size_t n, void *d_bytes, void *stream);

/* ---- evaluate_h: the quotient numerator h(X) on the extended coset ------------------------------
 * Evaluator::evaluate_h -- plonk/evaluation.rs:778-1226 (CPU twin) / :1229-1985 (cuda).
 * The Rust side flattens its `Evaluator` (plonk/evaluation.rs:270-296) into this plain descriptor:
 * ValueSource :44-57, Calculation :95-112, value_parts, lookup_results, shuffle_results.  All column
 * pointers are extended cosets (2^extended_k Fr).  One circuit instance (the cuda path asserts it,
 * plonk/evaluation.rs:1259). */
enum { H2_VS_CONSTANT = 0, H2_VS_INTERMEDIATE = 1, H2_VS_FIXED = 2, H2_VS_ADVICE = 3, H2_VS_INSTANCE = 4 };
enum {
    H2_CALC_ADD = 0, H2_CALC_SUB = 1, H2_CALC_MUL = 2, H2_CALC_NEGATE = 3,
    H2_CALC_LC_CHALLENGE = 4, /* (a + challenge^power) * b */
    H2_CALC_LC_THETA = 5,     /* a * theta + b */
    H2_CALC_ADD_CHALLENGE = 6,/* a + challenge */
    H2_CALC_STORE = 7
};
enum { H2_CHALLENGE_BETA = 0, H2_CHALLENGE_GAMMA = 1 };
enum { H2_ANY_ADVICE = 0, H2_ANY_FIXED = 1, H2_ANY_INSTANCE = 2 };

typedef struct {
    uint32_t kind;  /* H2_VS_* */
    uint32_t index; /* constant / intermediate / column index */
    uint32_t rot;   /* index into `rotations` (columns only) */
} h2_value_source;

typedef struct {
    uint32_t op; /* H2_CALC_* */
    h2_value_source a, b;
    uint32_t challenge; /* H2_CHALLENGE_* (LC_CHALLENGE, ADD_CHALLENGE) */
    uint32_t power;     /* LC_CHALLENGE exponent p (p <= 1 means the challenge itself) */
} h2_calculation;

typedef struct {
    uint32_t k, extended_k;
    uint32_t blinding_factors; /* cs.blinding_factors(): last_rotation = -(blinding_factors + 1) */
    uint32_t chunk_len;        /* cs.degree() - 2: permutation columns per product set */
    /* the straight-line program */
    const uint64_t *constants;          uint32_t n_constants;    /* Fr each */
    const int32_t *rotations;           uint32_t n_rotations;
    const h2_calculation *calculations; uint32_t n_calculations;
    const h2_value_source *value_parts; uint32_t n_value_parts;
    /* lookup_results[t] = (table, products[sets], sums[sets]); flattened per lookup as
     * table, product_0, sum_0, product_1, sum_1, ...  (lookup_sets[t] = number of sets >= 1) */
    uint32_t n_lookups; const uint32_t *lookup_sets; const h2_calculation *lookup_calcs;
    /* shuffle_results[i] = (input, shuffle), flattened */
    uint32_t n_shuffles; const h2_calculation *shuffle_calcs;
    /* columns */
    const uint64_t *const *fixed;    uint32_t n_fixed;
    const uint64_t *const *advice;   uint32_t n_advice;
    const uint64_t *const *instance; uint32_t n_instance;
    const uint64_t *l0, *l_last, *l_active_row;
    /* permutation argument: sets[i].permutation_product_coset, p.columns, pk.permutation.cosets */
    uint32_t n_perm_sets;    const uint64_t *const *perm_z;
    uint32_t n_perm_columns; const uint32_t *perm_col_type; const uint32_t *perm_col_index;
    const uint64_t *const *perm_sigma;
    /* logup lookups: z cosets of every set of every lookup (in order), m coset per lookup */
    const uint64_t *const *lookup_z; const uint64_t *const *lookup_m;
    /* shuffles: product coset per shuffle */
    const uint64_t *const *shuffle_z;
    /* challenges and field constants */
    uint64_t y[4], beta[4], gamma[4], theta[4];
    uint64_t delta[4], zeta[4], extended_omega[4]; /* FieldExt::DELTA, ::ZETA, domain.get_extended_omega() */
    /* must be NULL (the slot of round 4's caller-supplied kernel: the library now generates, compiles and caches the
     * program's kernels itself, see h2_evalh_prepare) */
    const void *reserved;
    /* H2_EVALH_INTERPRET: run this call on the interpreter kernels even when generated ones exist (tests compare the two) */
    uint32_t flags;
    /* rows of the domain to evaluate: `values[row_begin .. row_begin + row_count)` are written, nothing else (row_count = 0:
     * the whole domain).  Column values are still read at rotated indices modulo the domain, so the rows
     * [row_begin - (blinding_factors + 1) * rot_scale ..., row_begin + row_count + max rotation * rot_scale) of every column
     * must be valid.  For callers that split one evaluation over several devices by row range. */
    uint32_t row_begin, row_count;
} h2_evalh_desc;
enum { H2_EVALH_INTERPRET = 1 };

/* Host buffers everywhere (descriptor and every pointer in it); values: 2^extended_k Fr out. */
int h2_evaluate_h(const h2_evalh_desc *desc, uint64_t *values);
/* The cuda `Evaluator::evaluate_h`'s own shape (plonk/evaluation.rs:1229-1241: advice / instance as coefficient forms,
 * a proving key without extended cosets, plonk.rs:226-240): every column pointer of the descriptor -- fixed, advice,
 * instance, l0, l_last, perm_z, perm_sigma, lookup_z, lookup_m, shuffle_z -- is a HOST coefficient vector of 2^k Fr;
 * l_active_row is HOST extended values (2^extended_k, as the reference keeps it); values: 2^extended_k Fr out (host).
 * Each distinct vector is uploaded once; the extended domain is visited one coset of the n-th roots of unity at a
 * time, so device memory is (distinct columns) x 2 x 2^k x 32 B + two extended vectors whatever extended_k is -- the
 * memory-bounded route (the reference bounds it with a 5-entry cache of extended FFTs, evaluation_gpu.rs:335-468). */
int h2_evaluate_h_coeff(const h2_evalh_desc *desc, uint64_t *values);
/* The vanishing argument's quotient h(X) in COEFFICIENT form from coefficient forms, in one call: the three steps the
 * reference's cuda path takes on host vectors of 2^extended_k elements -- Evaluator::evaluate_h (plonk/evaluation.rs:1229-1985,
 * the descriptor as for h2_evaluate_h_coeff), EvaluationDomain::divide_by_vanishing_poly (poly/domain.rs:354-373) and
 * extended_to_coeff (:328-350), the latter two in vanishing::Argument::construct (plonk/vanishing/prover.rs:69-112) -- with the
 * numerator's values never leaving the device.  out: out_len = n * quotient_poly_degree coefficients (the truncation of
 * domain.rs:346-347); t_evaluations / t_len as h2_divide_by_vanishing_poly, the scalars as h2_extended_to_coeff. */
int h2_quotient_poly_coeff(const h2_evalh_desc *desc, const uint64_t *t_evaluations, size_t t_len,
                           const uint64_t g_coset[4], const uint64_t g_coset_inv[4],
                           const uint64_t extended_omega_inv[4], const uint64_t extended_ifft_divisor[4],
                           uint64_t *out, size_t out_len);
/* Column / table pointers inside `desc` are DEVICE pointers (the descriptor itself and its program
 * arrays -- constants, rotations, calculations, ... and the pointer tables -- stay in host memory).
 * d_values: 2^extended_k Fr on the device.  Work space is taken from the library's arena. */
int h2_dev_evaluate_h(const h2_evalh_desc *desc, void *d_values, void *stream);

/* The generated form of evaluate_h.  The cuda `Evaluator::evaluate_h` of the reference is self-contained in the host
 * language (plonk/evaluation.rs:1229-1985, plonk/evaluation_gpu.rs:594-803); so is this: the first time a descriptor's
 * program (constants, rotations, calculations, value parts, lookup / shuffle calculations, permutation shape) reaches one
 * of the three entry points above, the library turns it into straight-line HIP -- intermediates in registers, every
 * argument's terms folded in, terms that share a factor (l_0, l_last, l_active_row, a gate selector) summed before the
 * factor multiplies them (exact field arithmetic: the same bits) -- compiles it with hipRTC for gfx950, keeps the code
 * object in memory and in a private directory (H2_JIT_CACHE, default <tmp>/halo2_hip_jit_<uid>) under the hash of the
 * program, and launches it for every later call with the same program.  H2_EVALH_JIT=0 (environment) or
 * H2_EVALH_INTERPRET (per call) keeps the interpreter kernels; so does a machine without libhiprtc.so (one warning on
 * stderr).  Both are HIP paths with identical results.
 *
 * h2_evalh_prepare  does that work ahead of the first proof -- at keygen -- for the CURRENT device (h2_set_device) and
 *                   reports what was built; pointers to columns inside `desc` are not read.
 * h2_evalh_compile  generates and compiles into the caches without touching a device (works on a machine without a GPU).
 * h2_evalh_source   the generated source of stage `stage` (for inspection): copies up to `cap` bytes including the
 *                   terminating NUL into `buf` and stores the full length (without NUL) in *len.
 * All three may be called from several threads at once (rayon workers at keygen), for the same program too.  A cache file
 * carries a SHA-256 trailer: a torn or damaged one is a miss and is rebuilt; every writer renames a file of its own into place.
 * Loaded code objects are kept per (program, device), at most H2_EVALH_PLANS_MAX of them (environment, default 128): the least
 * recently used one is unloaded and comes back from the disk cache when its program is seen again. */
typedef struct {
    uint32_t stages;                     /* kernels the program was cut into (1 unless it is too wide for one) */
    uint32_t terms;                      /* y-folded terms of the numerator: value parts + argument terms */
    uint32_t products_per_row;           /* field products the generated kernels spend per row */
    uint32_t reference_products_per_row; /* ... and the formulas of evaluation.rs:875-1219 as written */
    uint32_t vectors_read;               /* distinct column vectors read per row */
    uint32_t max_registers;              /* VGPRs + AGPRs of the widest stage */
    uint32_t scratch_bytes;              /* per-lane scratch of the worst stage (0 = no spills) */
    uint32_t from_cache;                 /* 1 = memory, 2 = disk, 0 = compiled now */
    uint32_t fused_pairs_per_row;        /* pairs of those products that share one Montgomery reduction (a b + c d: 3/4 of the work) */
} h2_evalh_info;
int h2_evalh_prepare(const h2_evalh_desc *desc, h2_evalh_info *info);
int h2_evalh_compile(const h2_evalh_desc *desc, h2_evalh_info *info);
int h2_evalh_source(const h2_evalh_desc *desc, uint32_t stage, char *buf, size_t cap, size_t *len);
/* The argument block stage `stage` of the generated program receives for THIS descriptor (its column pointers, challenges and
 * constants), `values`, the power tables of extended_omega (tw_lo[i] = omega^i for i < min(2^extended_k, 4096), tw_hi[j] =
 * omega^(4096 j)) and the row range -- exactly the bytes the launch passes by value; host arithmetic only, no device needed.
 * For inspection and for tests: with h2_evalh_source it lets a test compile the generated text for the host and run it against
 * the CPU oracle (tests/test_evalh_host_exec.py).  Copies into `buf` (cap bytes; 0 = only report) and stores the size in *len. */
int h2_evalh_stage_args(const h2_evalh_desc *desc, uint32_t stage, uint64_t *values, const uint64_t *tw_lo, const uint64_t *tw_hi,
                        uint64_t row_begin, uint64_t row_end, void *buf, size_t cap, size_t *len);
/* stages of generated kernels launched so far in this process (a test's proof that they, not the interpreter, ran) */
uint64_t h2_evalh_generated_launches(void);

/* ---- device memory and streams for hosts without a HIP binding of their own ------------------
 * The h2_dev_* entry points below take HIP device pointers and a HIP stream.  A host that already links HIP (or, like
 * the Python harness, borrows torch's allocator and streams) passes its own; a host that does not -- the Rust side of
 * INTEGRATION.md holding `Polynomial` values on the device between calls -- gets what it needs from these: plain
 * hipMalloc / hipFree / hipMemcpyAsync / stream wrappers on the CURRENT device (h2_set_device; device 0 by default).
 * h2_dev_download synchronises `stream` before it returns; h2_dev_upload is asynchronous for pinned host memory
 * (h2_host_alloc_pinned) and otherwise returns when the source has been read. */
int h2_set_device(int device);
int h2_dev_alloc(size_t bytes, void **d_out);
int h2_dev_free(void *d_ptr);
int h2_host_alloc_pinned(size_t bytes, void **out);
int h2_host_free_pinned(void *ptr);
int h2_stream_create(void **stream_out);
int h2_stream_destroy(void *stream);
int h2_stream_synchronize(void *stream);
int h2_dev_upload(void *d_dst, const void *src, size_t bytes, void *stream);
int h2_dev_download(void *dst, const void *d_src, size_t bytes, void *stream);

/* One coset of a larger domain, without a separate scaling pass (the unit of the coset-by-coset quotient: multi-GPU proofs,
 * memory-budgeted proofs, and circuits whose extended domain is more than (degree - 1) n points wide):
 *   h2_dev_coset_ntt   out[i] = sum_t coeffs[t] g^t omega^(i t)  -- coeff_to_extended (poly/domain.rs:270-287) restricted to
 *                      the points g omega^i; g^t is applied on the first pass's load from a cached two-level table.
 *                      d_coeffs is left untouched unless d_out == d_coeffs.
 *   h2_dev_coset_intt  in place: the coefficients of the polynomial of degree < n with the values d_a on g H;
 *                      a[t] *= divisor * g_inv^t fused into the last pass's store (divisor = 1/n).
 * Same values as h2_dev_distribute_powers + h2_dev_ntt / h2_dev_intt + h2_dev_distribute_powers. */
int h2_dev_coset_ntt(const void *d_coeffs, void *d_out, void *d_tmp, uint32_t log_n, const uint64_t g[4],
                     const uint64_t omega[4], void *stream);
int h2_dev_coset_intt(void *d_a, void *d_tmp, uint32_t log_n, const uint64_t g_inv[4], const uint64_t omega_inv[4],
                      const uint64_t divisor[4], void *stream);

/* Several vectors through one transform plan, up to 16 per launch (the columns of a wide witness: plonk/prover.rs:643-646
 * runs them as a par_iter).  A 2^20-point pass alone is one resident round of the chip, every workgroup waiting out its own
 * load -> stages -> store chain; with the tiles of many vectors in one grid the rounds overlap.  d_a / d_coeffs / d_out:
 * HOST arrays of `count` device pointers; d_tmp: min(count, 16) x 2^log_n Fr of scratch (NULL allowed for log_n <= 8).
 * Same values as `count` calls of h2_dev_ntt / h2_dev_intt / h2_dev_coset_ntt. */
int h2_dev_ntt_batch(void *const *d_a, size_t count, void *d_tmp, const uint64_t omega[4], uint32_t log_n, void *stream);
int h2_dev_intt_batch(void *const *d_a, size_t count, void *d_tmp, const uint64_t omega_inv[4], const uint64_t divisor[4],
                      uint32_t log_n, void *stream);
int h2_dev_coset_ntt_batch(const void *const *d_coeffs, void *const *d_out, size_t count, void *d_tmp, uint32_t log_n,
                           const uint64_t g[4], const uint64_t omega[4], void *stream);
/* `count` coefficient vectors of 2^k elements to the extended domain (h2_dev_coeff_to_extended each); d_out[i] != d_coeffs[i];
 * d_tmp: min(count, 16) x 2^extended_k Fr. */
int h2_dev_coeff_to_extended_batch(const void *const *d_coeffs, void *const *d_out, size_t count, void *d_tmp, uint32_t k,
                                   uint32_t extended_k, const uint64_t g_coset[4], const uint64_t g_coset_inv[4],
                                   const uint64_t extended_omega[4], void *stream);

/* ---- device-resident entry points ---------------------------------------------------------- */
/* Same semantics on HIP device pointers.  d_tmp: scratch of 2^log_n Fr (may be NULL for
 * log_n <= 8).  Results land in d_a (in place from the caller's view).
 * Device memory the library keeps per (device, log_n, omega) for the life of the process: the twiddle tables of the
 * transform (a few MiB) and, for 2^18 .. 2^26 points, the last pass's complete twiddle set -- 32 B x 2^log_n, one per
 * divisor it is used with (512 MiB at 2^24; built on first use; the environment variable H2_NTT_LAST_TABLE=0 or a failed
 * allocation leave the smaller two-level tables in use, at one more multiplication per element). */
int h2_dev_ntt(void *d_a, void *d_tmp, const uint64_t omega[4], uint32_t log_n, void *stream);
int h2_dev_intt(void *d_a, void *d_tmp, const uint64_t omega_inv[4], const uint64_t divisor[4], uint32_t log_n,
                void *stream);
int h2_dev_coeff_to_extended(const void *d_coeffs, void *d_out, void *d_tmp, uint32_t k, uint32_t extended_k,
                             const uint64_t g_coset[4], const uint64_t g_coset_inv[4],
                             const uint64_t extended_omega[4], void *stream);
int h2_dev_extended_to_coeff(void *d_a, void *d_tmp, uint32_t extended_k, const uint64_t g_coset[4],
                             const uint64_t g_coset_inv[4], const uint64_t extended_omega_inv[4],
                             const uint64_t extended_ifft_divisor[4], void *stream);
/* MSM over device-resident scalars and bases.  d_scratch/scratch_bytes: device workspace sized by
 * h2_msm_scratch_bytes(n, max_bits).  out_xyz is HOST memory: the call synchronises `stream` to
 * read back the per-window partial sums (<= a few KB) and finishes the window combine on the host. */
/* Scalar distributions: zero scalars cost nothing; a column whose values crowd a few buckets takes the skew path of
 * the sort; a non-zero value found on >= 1/4 of 64 sampled rows (a grand-product column over padding rows) gets a window
 * of its own -- one addition per such row plus one scalar multiplication on the host.  The result is the same group
 * element in every case; h2_msm_scratch_bytes covers the larger layout. */
size_t h2_msm_scratch_bytes(size_t n, uint32_t max_bits);
/* the Pippenger shape the library will use: window bits c, number of windows, buckets per window */
int h2_msm_shape(size_t n, uint32_t max_bits, uint32_t *c, uint32_t *windows, uint32_t *buckets_per_window);
int h2_dev_msm(const void *d_scalars, const void *d_bases, size_t n, uint32_t max_bits, void *d_scratch,
               size_t scratch_bytes, uint64_t out_xyz[12], void *stream);
/* `count` MSMs over the SAME bases (one per column: plonk/prover.rs:293-299 commits every advice column
 * against g_lagrange; :477-487, :540-549 the z polynomials).  d_scalars: host array of `count` device
 * pointers.  scratch_bytes >= 2 * round_up(h2_msm_scratch_bytes(n, max_bits), 256): consecutive MSMs
 * alternate between two internal streams so one MSM's latency-bound bucket reduction overlaps the next
 * one's accumulation.  out_xyz: count x 12 u64 in HOST memory.  Synchronous. */
int h2_dev_msm_batch(const void *const *d_scalars, size_t count, const void *d_bases, size_t n, uint32_t max_bits,
                     void *d_scratch, size_t scratch_bytes, uint64_t *out_xyz, void *stream);
/* The same pipeline with a base table and a scalar bound PER COLUMN (host arrays of `count` entries): one call commits
 * 16-bit witness columns next to full-width ones (plonk/prover.rs:293-299 computes max_bits per column) and columns over
 * g_lagrange next to one over g.  scratch_bytes >= 2 * max over the columns of round_up(h2_msm_scratch_bytes(n, max_bits_each[i]), 256)
 * (the layout depends on the window size picked for the bound: it is not monotonic in max_bits). */
/* Scratch that lets the batch entry points commit `count` columns sharing one base table and one bound as a FUSED group
 * (their windows become the windows of one wide MSM: sampling, sort launches, finish / reduce latency, synchronisation
 * and read-back are paid once per group -- what a witness of many narrow columns needs).  With less scratch (but at
 * least the 2 x h2_msm_scratch_bytes of the pipeline) the calls still succeed, column by column. */
size_t h2_msm_batch_scratch_bytes(size_t n, uint32_t max_bits, size_t count);
int h2_dev_msm_batch_ex(const void *const *d_scalars, const void *const *d_bases_each, const uint32_t *max_bits_each,
                        size_t count, size_t n, void *d_scratch, size_t scratch_bytes, uint64_t *out_xyz, void *stream);
/* Shifted-base table for a device-resident base set that is committed against repeatedly -- the SRS (`params.g`,
 * `params.g_lagrange`: poly/commitment.rs:148-170 `commit` / `commit_lagrange`; the reference re-uploads them per call,
 * arithmetic.rs:354-360).  Builds T[j][i] = [2^(o_j)] bases[i] for the `digits` (11..32) digit offsets o_j (0 = chosen from n:
 * 12 at 2^24, 14 at 2^20) in library-owned device memory (h2_dev_bases_precompute_bytes: digits x n x 64 B) and
 * remembers it under `d_bases`.  From then on every h2_dev_msm / _batch / _batch_ex whose bases lie inside
 * [d_bases, d_bases + n) and whose bound needs more windows than digits adds all digits of a scalar into ONE shared
 * bucket set: ~20 % fewer point additions at 2^24, one reduction instead of one per window, no host Horner.  The sums
 * of every 256 consecutive bases are tabulated too (n / 256 points): a column whose dominant value fills whole blocks
 * of rows (a grand product over padding rows) adds one point per block instead of 256, in either form.  Same group
 * element (multiexp_serial, arithmetic.rs:20-108).  Call it before sizing scratch with h2_msm_scratch_bytes (over a
 * table the sort keeps a second 8-byte-per-entry buffer: 5.9 GB instead of 4.6 GB at 2^24).  The
 * bases must not change while the table exists (as a guard, every MSM compares 64 sampled base rows with the table's
 * own copy first and drops a table that no longer matches); h2_dev_bases_forget(d_bases) frees it (synchronises the
 * device).
 * H2_MSM_NO_TABLE=1 in the environment ignores all tables.  Synchronous (~0.3 s at 2^24). */
int h2_dev_bases_precompute(const void *d_bases, size_t n, uint32_t digits, void *stream);
int h2_dev_bases_forget(const void *d_bases);
size_t h2_dev_bases_precompute_bytes(size_t n, uint32_t digits);
int h2_dev_eval_op(int op, void *d_res, const void *d_l, const void *d_r, int32_t l_rot, int32_t r_rot, size_t size,
                   const uint64_t c[4], void *stream);
int h2_dev_divide_by_vanishing_poly(void *d_a, size_t size, const void *d_t_evaluations, size_t t_len, void *stream);
int h2_dev_batch_mont(void *d_a, size_t n, void *stream);
int h2_dev_batch_unmont(void *d_a, size_t n, void *stream);
/* Compact witness columns.  The reference hands create_proof 32-byte cells whatever they hold (plonk/prover.rs:255-312:
 * `advice: Vec<Polynomial<C::Scalar, LagrangeCoeff>>`), and its cuda path uploads them as such; most cells of a real trace are
 * booleans, bytes, 16-bit limbs or products of a few of them.  A caller that knows a column fits 64 bits sends 8 bytes per
 * cell (d_src: n x u64 on the device, e.g. copied from pinned host memory) and widens it here to canonical scalars:
 * d_dst[i] = {d_src[i], 0, 0, 0} (n x 32 B; canonical form -- follow with h2_dev_batch_mont as for any uploaded column). */
int h2_dev_widen_u64(const void *d_src, size_t n, void *d_dst, void *stream);
/* find_max_scalar_bits (plonk/prover.rs:237-254) for `count` CANONICAL columns of n scalars resident on the device, in
 * one launch and one synchronisation: out_bits[i] (host) = bit length of the largest value of column i (0 for an all-zero
 * column) -- the `max_bits` of commit_lagrange_with_bound.  d_words: count * 32 bytes of device scratch. */
int h2_dev_max_scalar_bits(const void *const *d_cols, size_t count, size_t n, void *d_words, uint32_t *out_bits,
                           void *stream);

/* ---- synthetic workload (bench.py / tests; not a reference entry point) --------------------- */
/* n deterministic valid G1Affine points (try-and-increment on y^2 = x^3 + 3) into d_out (n x 64 B). */
int h2_dev_random_points(uint64_t seed, size_t n, void *d_out, void *stream);

/* ---- measurement hooks (bench.py) ------------------------------------------------------------ */
/* Wall time in ms of the last timed region recorded with HIP events on `stream`:
 * h2_timer_start / h2_timer_stop bracket any sequence of h2_dev_* calls on that stream. */
int h2_timer_start(void *stream);
int h2_timer_stop(void *stream, float *ms_out);

#ifdef __cplusplus
}
#endif
#endif /* HALO2_HIP_H */
