// ntt.hip -- radix-2^B multi-pass NTT over BN254 Fr for gfx950.
//
// Replaces ec-gpu-gen's `SingleFftKernel::{radix_fft, radix_ifft}` (called from
// /root/reference/halo2_proofs/src/arithmetic.rs:495-534) and the device-resident driver
// `do_fft_core` (plonk/evaluation_gpu.rs:976-1052); semantics are those of the CPU twin
// `best_fft_cpu` (arithmetic.rs:556-645): natural order in, natural order out,
// X[k] = sum_j x[j] * omega^(j*k).
//
// Decomposition (not the reference's): n = R_0 * R_1 * ... * R_{P-1}, R_p = 2^{B_p} <= 2^9.
//   input index   j = sum_p j_p * S_p,   S_p = 2^(L - B_0 - ... - B_p)   (j_0 most significant)
//   output index  k = sum_p k_p * T_p,   T_p = 2^(B_0 + ... + B_{p-1})   (k_0 least significant)
// Pass p replaces digit j_p by k_p *in place* (an R_p-point DFT along stride S_p) after
// multiplying element j_p by omega^(j_p * S_p * K_{p-1}), K_{p-1} = sum_{q<p} k_q T_q.
// The last pass reads R_{P-1} contiguous elements per DFT and scatters to the natural output
// order, so it is out of place; tiles hold C DFTs with consecutive K so both the loads
// (contiguous rows) and the stores (C consecutive outputs) are coalesced.
// One workgroup = one LDS tile of R_p x C elements (32 B each); butterflies are radix-2
// DIT on bit-reversed rows, twiddles w_R^e from a per-pass LDS table.
//
// Fused into the passes (so the reference's separate kernels/loops disappear):
//   * zero padding  (eval_fft_prepare, evaluation_gpu.rs:890-900; domain.rs:280)
//   * zeta-power coset pre-scale (distribute_powers_zeta, domain.rs:382-398)
//   * 1/n and zeta^-1 post-scale (domain.rs:404-409, :341)
#include <algorithm>
#include <cstdlib>

#include "common.hpp"
#include "ntt.hpp"

namespace h2 {

static constexpr int NTT_BATCH_MAX = 16;  // vectors per launch of a batched transform (pointers travel in the kernel arguments)
static constexpr int LO_BITS = 12;  // two-level twiddle tables: w^e = lo[e & 4095] * hi[e >> 12]

// ---------------------------------------------------------------- table generation
// A twiddle table in one of two forms.  pair = 0: out[i] = w, Montgomery form (the operand of fp_mul / fp_mul_wide).
// pair = 1: out[2 i] = w as a PLAIN residue, out[2 i + 1] = floor(w 2^256 / r) -- the operands of fp_mul_const (field.hpp), the
// constant-operand product the passes with CW use for every twiddle they read from a table.
__device__ __forceinline__ void tw_store(Fr* out, uint32_t i, const Fr& w_mont, uint32_t pair) {
    if (!pair) {
        fp_store(out + i, w_mont);
        return;
    }
    Fr w, q;
    fp_const_pair(w_mont, w, q);
    fp_store(out + 2 * (size_t)i, w);
    fp_store(out + 2 * (size_t)i + 1, q);
}

// out[i] = base^(i * mul)   (i < count)
__global__ void __launch_bounds__(256) k_pow_table(Fr* out, Fr base, uint32_t mul, uint32_t count, uint32_t pair) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    // base^(i*mul): exponent < 2^28 always (order of omega divides 2^28)
    tw_store(out, i, fp_pow_u32(base, i * mul), pair);
}

// out[(rho << kbits) | K] = base^((rho * K << s_log) mod n)   -- the complete inter-pass twiddle set of a pass
__global__ void __launch_bounds__(256) k_direct_table(Fr* out, Fr base, uint32_t kbits, uint32_t s_log, uint32_t log_n,
                                                      uint32_t count, uint32_t pair) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint32_t rho = i >> kbits, K = i & ((1u << kbits) - 1);
    uint32_t e = (uint32_t)(((uint64_t)rho * K) << s_log) & ((1u << log_n) - 1);
    tw_store(out, i, fp_pow_u32(base, e), pair);
}

// out[(K << bits) | rho] = base^((rho * K) mod n) (* d when `scale`): the last pass's inter-pass twiddles in load order
__global__ void __launch_bounds__(256) k_last_table(Fr* out, Fr base, uint32_t bits, uint32_t log_n, Fr d, uint32_t scale,
                                                    uint32_t pair) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;  // grid covers 2^log_n exactly (log_n >= 8)
    const uint32_t rho = i & ((1u << bits) - 1), K = i >> bits;
    const uint32_t e = (uint32_t)((uint64_t)rho * K) & ((1u << log_n) - 1);
    Fr w = fp_pow_u32(base, e);
    if (scale) w = fp_mul(w, d);
    tw_store(out, i, w, pair);
}

// out[i] = in[i] * d
__global__ void __launch_bounds__(256) k_scale_table(Fr* out, const Fr* in, Fr d, uint32_t count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) fp_store(out + i, fp_mul(fp_load(in + i), d));
}

// ---------------------------------------------------------------- the pass kernel
struct PassArgs {
    const Fr* in;
    Fr* out;
    const Fr* tw_bfly;  // R/2 entries: (w^(n/R))^e   (CW kernels: R/2 PAIRS -- plain value, quotient -- like tw_direct; see tw_store)
    const Fr* tw_lo;    // min(n, 4096) entries: w^i
    const Fr* tw_hi;    // n >> 12 entries: w^(i << 12)   (unused when n <= 4096)
    const Fr* tw_direct;  // non-null: inter-pass twiddle = tw_direct[(rho << consumed) | K] (no generation multiply)
    uint32_t direct_kmajor;  // ... = tw_direct[(K << B) | rho] instead (the last pass's table: read in load order)
    uint32_t hi_scaled;  // tw_hi already carries the uniform post-scale (1/n): never skip, no post multiply
    Fr pre3[3];         // has_pre3: x *= pre3[idx % 3] on the first-pass load (idx % 3 == 0 skipped)
    Fr post3[3];        // has_post3: y *= post3[idx % 3] on the final store
    uint32_t has_pre3, has_post3, post3_uniform;
    uint32_t log_n, B, s_log, t_log;
    uint32_t nprev;       // number of earlier passes
    uint32_t prevB[4];    // their bit widths
    uint32_t prevT[4];    // their T_q logs
    uint32_t is_last, in_len, log_c;
    // generic coset scale (coeff -> one coset of the extended domain and back): sc[i] = g^i = sc_lo[i & 4095] * sc_hi[i >> 12]
    // (sc_hi also carries the iNTT divisor).  scale_mode 1: x[i] *= sc[i] on the first pass's load (the inter-pass twiddle
    // path, which a first pass never uses, does it); 2: y[i] *= sc[i] on the final store
    const Fr* sc_lo;
    const Fr* sc_hi;
    uint32_t scale_mode;
    // several transforms of one plan in ONE launch (blockIdx.y picks the vector): a 2^20-point pass is 1024 tiles, one
    // resident round of the chip in which every workgroup waits out its own load -> stages -> store chain; with the
    // tiles of 8-16 vectors in the grid the rounds overlap (the columns of a wide witness on one coset)
    uint32_t batch;
    const Fr* in_b[NTT_BATCH_MAX];
    Fr* out_b[NTT_BATCH_MAX];
    uint32_t zskip;  // first pass of a zero-padded transform: rows rho >= R >> zskip are zero (see the load loop)
    uint32_t radix4;  // stage loop: two stages per LDS round trip (four elements per lane), launched with R / 4 * C threads
};

__device__ __forceinline__ uint32_t bitrev(uint32_t x, uint32_t bits) {
    return bits == 0 ? 0u : (__brev(x) >> (32 - bits));
}

// hi (digits k_0..k_{p-1}, k_0 most significant) -> K = sum k_q << T_q
__device__ __forceinline__ uint32_t hi_to_K(uint32_t hi, const PassArgs& a) {
    uint32_t K = 0;
    for (int q = (int)a.nprev - 1; q >= 0; q--) {
        uint32_t d = hi & ((1u << a.prevB[q]) - 1);
        hi >>= a.prevB[q];
        K |= d << a.prevT[q];
    }
    return K;
}
__device__ __forceinline__ uint32_t K_to_hi(uint32_t K, const PassArgs& a) {
    uint32_t hi = 0;
    for (uint32_t q = 0; q < a.nprev; q++) {
        uint32_t d = (K >> a.prevT[q]) & ((1u << a.prevB[q]) - 1);
        hi = (hi << a.prevB[q]) | d;
    }
    return hi;
}

__device__ __forceinline__ Fr twiddle_pow(const PassArgs& a, uint32_t e) {
    if (a.log_n <= LO_BITS) return fp_load(a.tw_lo + e);
    Fr lo = fp_load(a.tw_lo + (e & ((1u << LO_BITS) - 1)));
    Fr hi = fp_load(a.tw_hi + (e >> LO_BITS));
    return fp_mul(lo, hi);
}

extern __shared__ __attribute__((aligned(16))) uint4 h2_smem[];

// LDS tiles keep the two 16-byte halves of an element in separate planes: a wave then reads 16 B at a
// 16-B lane stride (conflict-free ds_read_b128) instead of 16 B at a 32-B stride (2-way conflicts).
__device__ __forceinline__ Fr lds_get(const uint4* lo, const uint4* hi, uint32_t i) {
    uint4 a = lo[i], b = hi[i];
    Fr r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}
__device__ __forceinline__ void lds_put(uint4* lo, uint4* hi, uint32_t i, const Fr& v) {
    lo[i] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    hi[i] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// LAZY: the lazy domain of field.hpp -- values below 4p in LDS and between the passes, products by fp_mul_wide (no final
// subtraction), bare additions; canonical residues come back at the last pass's store.  Same field elements, same output.
// FB != 0: the pass geometry as compile-time constants (B = FB bits, 4 columns per tile, no zero-padding skip, RADIX4 lanes):
// the stage loop unrolls with constant strides, the LDS planes sit at immediate offsets and the loops over a tile collapse to
// one iteration -- the same instructions on the same operands minus most of the index arithmetic.  FB = 0: everything from `a`.
// CW: the twiddles this pass reads from tables -- the butterfly twiddles in LDS and the tabulated inter-pass twiddles (tw_direct)
// -- are (plain value, quotient) pairs and multiply by fp_mul_const: 115 multiply-adds per product instead of 136 (field.hpp).
// Twiddles COMPOSED at run time (lo x hi of the two-level tables, the coset scales, pre3 / post3) stay Montgomery products.
// DP: tw_direct holds pairs too (the 2^16-entry table of a middle pass: 4 MiB, read out of L2).  The LAST pass's complete table
// stays in Montgomery form whatever CW says: as pairs it is 64 B per element streamed from HBM next to the 64 B of data, and the
// pass -- 2.1 GB per launch at 2^24 -- stopped following its instruction count (590 us either way, profiles/r6_ntt_constw.txt).
template <bool RADIX4, bool LAZY, uint32_t FB = 0, bool CW = false, bool DP = false>
__global__ void __launch_bounds__(512, 4) k_ntt_pass(PassArgs a) {
    static_assert(!CW || LAZY, "the constant-operand product returns values of the lazy domain");
    static_assert(!DP || CW, "pairs in tw_direct only next to pairs in tw_bfly");
    // x * w for a canonical twiddle w: canonical arithmetic, or any x < 2^256 -> a value below 2p
    auto tmul = [](const Fr& x, const Fr& w) -> Fr {
        if constexpr (LAZY) return fp_mul_wide(x, w);
        else return fp_mul(x, w);
    };
    const uint32_t B = FB ? FB : a.B, R = 1u << B, log_c = FB ? 2u : a.log_c, C = 1u << log_c;
    const uint32_t zskip = FB ? 0u : a.zskip;
    uint4* t_lo = h2_smem;                  // R*C low halves
    uint4* t_hi = t_lo + (R << log_c);      // R*C high halves
    uint4* w_lo = t_hi + (R << log_c);      // R/2 butterfly twiddles, low / high halves
    uint4* w_hi = w_lo + (R >> 1) + (CW ? 0 : 1);
    // CW: two more planes for the quotients, no padding: a 256 x 4 tile + 128 pairs is 40 KiB exactly, four workgroups per CU
    uint4* q_lo = w_hi + (R >> 1);
    uint4* q_hi = q_lo + (R >> 1);
    // x * (butterfly twiddle i)
    struct Tw {
        Fr w, q;
    };
    auto tw_get = [&](uint32_t i) __attribute__((always_inline)) -> Tw {
        Tw t;
        t.w = lds_get(w_lo, w_hi, i);
        if constexpr (CW) t.q = lds_get(q_lo, q_hi, i);
        return t;
    };
    auto bmul = [](const Fr& x, const Tw& t) __attribute__((always_inline)) -> Fr {
        if constexpr (CW) return fp_mul_const(x, t.w, t.q);
        else if constexpr (LAZY) return fp_mul_wide(x, t.w);
        else return fp_mul(x, t.w);
    };
    const uint32_t nthreads = FB ? ((RADIX4 ? (R >> 2) : (R >> 1)) << log_c) : blockDim.x;  // == max(R/2 * C, 1) (RADIX4: half)
    const uint32_t tid = threadIdx.x;
    const Fr* const in_p = a.batch ? a.in_b[blockIdx.y] : a.in;   // (wave-uniform: scalar loads from the kernel arguments)
    Fr* const out_p = a.batch ? a.out_b[blockIdx.y] : a.out;
    const uint32_t n_mask = (a.log_n >= 32) ? 0xffffffffu : ((1u << a.log_n) - 1);

    for (uint32_t i = tid; i < (R >> 1); i += nthreads) {
        if constexpr (CW) {
            lds_put(w_lo, w_hi, i, fp_load(a.tw_bfly + 2 * i));
            lds_put(q_lo, q_hi, i, fp_load(a.tw_bfly + 2 * i + 1));
        } else {
            lds_put(w_lo, w_hi, i, fp_load(a.tw_bfly + i));
        }
    }

    const uint32_t tile_id = blockIdx.x;
    const uint32_t total = R << log_c;

    // ---- tile geometry
    uint32_t base = 0, K_uniform = 0;
    const uint32_t S = 1u << a.s_log;
    if (!a.is_last) {
        // tiles: for each hi, for each chunk of C consecutive low positions
        uint32_t chunks_per_hi = S >> log_c;
        uint32_t hi = tile_id / chunks_per_hi, lo0 = (tile_id % chunks_per_hi) << log_c;
        base = (hi << (B + a.s_log)) + lo0;
        K_uniform = hi_to_K(hi, a);
    }

    // ---- load (+ zero pad, coset pre-scale, inter-pass twiddle), bit-reversed rows into LDS.
    // A lane's NE elements are fetched as ONE batch -- every global load (the elements, then their twiddles) is issued
    // before the first product needs one -- so a tile pays one memory latency, not one per element (the rows of an early
    // pass are 2 MiB apart: each of those loads is a DRAM page of its own).
    constexpr uint32_t NE = RADIX4 ? 4 : 2;
    for (uint32_t e0 = tid; e0 < total; e0 += NE * nthreads) {
        uint32_t rho[NE], col[NE], idx[NE], Kk[NE];
        bool live[NE];
        Fr x[NE];
#pragma unroll
        for (uint32_t q = 0; q < NE; q++) {
            const uint32_t e = e0 + q * nthreads;
            if (!a.is_last) {
                col[q] = e & (C - 1);
                rho[q] = (e >> log_c) & (R - 1);
                idx[q] = base + (rho[q] << a.s_log) + col[q];
                Kk[q] = K_uniform;
            } else {
                rho[q] = e & (R - 1);
                col[q] = (e >> B) & (C - 1);
                Kk[q] = (tile_id << log_c) + col[q];
                idx[q] = (K_to_hi(Kk[q], a) << B) + rho[q];
            }
            // Zero padding by 2^z (coeff_to_extended): the rows rho >= R >> z of the first pass are zero, so its first z
            // stages are butterflies (u, 0) -> (u, u) whatever the twiddle: each loaded element is written to the 2^z rows
            // those stages would copy it to (the low z bits of the bit-reversed row index) and the stage loop starts at z.
            live[q] = e < total && !(zskip && rho[q] >= (R >> zskip));
        }
#pragma unroll
        for (uint32_t q = 0; q < NE; q++) {
            x[q] = fp_zero<FrParams>();
            if (live[q] && idx[q] < a.in_len) x[q] = fp_load(in_p + idx[q]);
        }
        if (a.has_pre3) {
#pragma unroll
            for (uint32_t q = 0; q < NE; q++) {
                const uint32_t m = idx[q] % 3;
                Fr w;
#pragma unroll
                for (int l = 0; l < 8; l++) w.l[l] = m == 1 ? a.pre3[1].l[l] : a.pre3[2].l[l];
                if (m != 0) x[q] = tmul(x[q], w);
            }
        }
        const bool pre_scale = a.scale_mode == 1u && a.nprev == 0;
        if (a.nprev != 0 || pre_scale) {
            // omega^(rho * S * K); a unit twiddle (rho = 0 or K = 0) multiplies like any other: the tables hold it
            if (a.tw_direct != nullptr && !pre_scale) {
#pragma unroll
                for (uint32_t q0 = 0; q0 < NE; q0 += 2) {
                    if constexpr (DP) {
                        // (plain value, quotient) pairs, one at a time: two pairs in flight next to the four elements spilled
#pragma unroll
                        for (uint32_t q = q0; q < q0 + 2; q++) {
                            const size_t at = a.direct_kmajor ? ((Kk[q] << B) | rho[q]) : ((rho[q] << a.t_log) | Kk[q]);
                            const Fr w = fp_load(a.tw_direct + 2 * at), wq = fp_load(a.tw_direct + 2 * at + 1);
                            x[q] = fp_mul_const(x[q], w, wq);
                        }
                    } else {
                        Fr w[2];
#pragma unroll
                        for (uint32_t q = 0; q < 2; q++)
                            w[q] = fp_load(a.tw_direct + (a.direct_kmajor ? ((Kk[q0 + q] << B) | rho[q0 + q]) : ((rho[q0 + q] << a.t_log) | Kk[q0 + q])));
#pragma unroll
                        for (uint32_t q = 0; q < 2; q++) x[q0 + q] = tmul(x[q0 + q], w[q]);
                    }
                }
            } else if (a.log_n <= LO_BITS && !pre_scale) {
#pragma unroll
                for (uint32_t q = 0; q < NE; q++) {
                    const uint32_t ex = (uint32_t)(((uint64_t)rho[q] * Kk[q]) << a.s_log) & n_mask;
                    x[q] = tmul(x[q], fp_load(a.tw_lo + ex));
                }
            } else {
                // two at a time: 16 twiddle halves in flight next to the elements keeps the kernel within 128 VGPRs
                // (the coset pre-scale of a first pass, g^idx from its own two-level table, runs through the same code)
                const Fr* const two_lo = pre_scale ? a.sc_lo : a.tw_lo;
                const Fr* const two_hi = pre_scale ? a.sc_hi : a.tw_hi;
#pragma unroll
                for (uint32_t q0 = 0; q0 < NE; q0 += 2) {
                    Fr wl[2], wh[2];
#pragma unroll
                    for (uint32_t q = 0; q < 2; q++) {
                        const uint32_t ex = pre_scale ? (idx[q0 + q] & n_mask)
                                                      : ((uint32_t)(((uint64_t)rho[q0 + q] * Kk[q0 + q]) << a.s_log) & n_mask);
                        wl[q] = fp_load(two_lo + (ex & ((1u << LO_BITS) - 1)));
                        wh[q] = fp_load(two_hi + (ex >> LO_BITS));
                    }
#pragma unroll
                    for (uint32_t q = 0; q < 2; q++) x[q0 + q] = tmul(x[q0 + q], fp_mul(wl[q], wh[q]));
                }
            }
        }
#pragma unroll
        for (uint32_t q = 0; q < NE; q++) {
            // (row and column again from e: cheaper than keeping them in registers across the products)
            const uint32_t e = e0 + q * nthreads;
            const uint32_t r_q = a.is_last ? (e & (R - 1)) : ((e >> log_c) & (R - 1));
            const uint32_t c_q = a.is_last ? ((e >> B) & (C - 1)) : (e & (C - 1));
            if (e >= total || (zskip && r_q >= (R >> zskip))) continue;
            if (zskip) {
                for (uint32_t m = 0; m < (1u << zskip); m++) lds_put(t_lo, t_hi, ((bitrev(r_q, B) | m) << log_c) + c_q, x[q]);
            } else {
                lds_put(t_lo, t_hi, (bitrev(r_q, B) << log_c) + c_q, x[q]);
            }
        }
    }
    __syncthreads();

    // ---- B radix-2 DIT stages in LDS.  With `radix4` two consecutive stages share one round trip: a lane takes the four
    // rows p, p + h, p + 2h, p + 3h (h = 2^s, bits s and s + 1 of p clear), runs the two stage-s butterflies (one
    // twiddle, index r = p mod h, for both) and the two stage-(s+1) butterflies (indices r and r + h) in registers and
    // writes the four rows back: half the LDS instructions, address arithmetic and barriers of the stage-by-stage loop,
    // the same products on the same operands.
    uint32_t s0 = zskip;
    if constexpr (RADIX4) {
        const uint32_t nunits = total >> 2;
        auto round4 = [&](const uint32_t s) __attribute__((always_inline)) {
            const uint32_t h = 1u << s;
            const uint32_t log_per = (B - 2 + log_c) - s;  // units that share one r: 2^log_per
            const bool by_r = s != 0 && log_per >= 6 && (nthreads & 63) == 0;
            for (uint32_t t = tid; t < nunits; t += nthreads) {
                uint32_t c, r, p;
                if (by_r) {
                    r = t >> log_per;
                    const uint32_t j = t & ((1u << log_per) - 1);
                    c = j & (C - 1);
                    p = ((j >> log_c) << (s + 2)) | r;
                } else {
                    c = t & (C - 1);
                    const uint32_t b = t >> log_c;
                    r = b & (h - 1);
                    p = ((b >> s) << (s + 2)) | r;
                }
                const uint32_t i0 = (p << log_c) + c, step = h << log_c;
                Fr x0 = lds_get(t_lo, t_hi, i0), x1 = lds_get(t_lo, t_hi, i0 + step);
                Fr x2 = lds_get(t_lo, t_hi, i0 + 2 * step), x3 = lds_get(t_lo, t_hi, i0 + 3 * step);
                const bool unit = s == 0 || (by_r && r == 0);  // the twiddles of index r are 1 (wave-uniform test)
                if constexpr (LAZY) {
                    // rows below 4p: the operands of a product go in as they are, the others are brought below 2p
                    x0 = fp_lazy_red2p(x0);
                    x2 = fp_lazy_red2p(x2);
                    if (!unit) {
                        const Tw wa = tw_get(r << (B - 1 - s));
                        x1 = bmul(x1, wa);
                        x3 = bmul(x3, wa);
                    } else {
                        x1 = fp_lazy_red2p(x1);
                        x3 = fp_lazy_red2p(x3);
                    }
                    const Fr y0 = fp_lazy_add_red(x0, x1), y1 = fp_lazy_sub_red(x0, x1);   // below 2p: added to next
                    Fr y2 = fp_lazy_add(x2, x3), y3 = fp_lazy_sub(x2, x3);                  // below 4p: multiplied next
                    y2 = unit ? fp_lazy_red2p(y2) : bmul(y2, tw_get(r << (B - 2 - s)));
                    y3 = bmul(y3, tw_get((r + h) << (B - 2 - s)));
                    lds_put(t_lo, t_hi, i0, fp_lazy_add(y0, y2));
                    lds_put(t_lo, t_hi, i0 + 2 * step, fp_lazy_sub(y0, y2));
                    lds_put(t_lo, t_hi, i0 + step, fp_lazy_add(y1, y3));
                    lds_put(t_lo, t_hi, i0 + 3 * step, fp_lazy_sub(y1, y3));
                } else {
                    if (!unit) {
                        const Tw wa = tw_get(r << (B - 1 - s));
                        x1 = bmul(x1, wa);
                        x3 = bmul(x3, wa);
                    }
                    Fr y0 = fp_add(x0, x1), y1 = fp_sub(x0, x1), y2 = fp_add(x2, x3), y3 = fp_sub(x2, x3);
                    if (!unit) y2 = bmul(y2, tw_get(r << (B - 2 - s)));
                    y3 = bmul(y3, tw_get((r + h) << (B - 2 - s)));
                    lds_put(t_lo, t_hi, i0, fp_add(y0, y2));
                    lds_put(t_lo, t_hi, i0 + 2 * step, fp_sub(y0, y2));
                    lds_put(t_lo, t_hi, i0 + step, fp_add(y1, y3));
                    lds_put(t_lo, t_hi, i0 + 3 * step, fp_sub(y1, y3));
                }
            }
            __syncthreads();
        };
        if constexpr (FB != 0) {
#pragma unroll
            for (uint32_t s = 0; s + 1 < FB; s += 2) round4(s);
            s0 = FB & ~1u;
        } else {
            for (; s0 + 1 < B; s0 += 2) round4(s0);
        }
    }
    const uint32_t nbf = total >> 1;
    for (uint32_t s = s0; s < B; s++) {
        const uint32_t h = 1u << s;
        const uint32_t log_per = (B - 1 + log_c) - s;  // butterflies that share one twiddle index r: 2^log_per
        const bool by_r = s != 0 && log_per >= 6 && (nthreads & 63) == 0;
        for (uint32_t t = tid; t < nbf; t += nthreads) {
            uint32_t c, r, i;
            if (by_r) {
                // early stages: order the butterflies by twiddle index so r is uniform across a wave and the
                // r == 0 waves (twiddle 1) skip the multiplication: 1/2, 1/4, 1/8 ... of stages 1, 2, 3 ...
                r = t >> log_per;
                uint32_t j = t & ((1u << log_per) - 1);
                c = j & (C - 1);
                i = ((j >> log_c) << (s + 1)) | r;
            } else {
                c = t & (C - 1);
                uint32_t b = t >> log_c;
                r = b & (h - 1);
                i = ((b >> s) << (s + 1)) | r;
            }
            const uint32_t iu = (i << log_c) + c, iv = ((i + h) << log_c) + c;
            Fr u = lds_get(t_lo, t_hi, iu), v = lds_get(t_lo, t_hi, iv);
            const bool skip = s == 0 || (by_r && r == 0);
            if constexpr (LAZY) {
                u = fp_lazy_red2p(u);
                v = skip ? fp_lazy_red2p(v) : bmul(v, tw_get(r << (B - 1 - s)));
                lds_put(t_lo, t_hi, iu, fp_lazy_add(u, v));
                lds_put(t_lo, t_hi, iv, fp_lazy_sub(u, v));
            } else {
                if (!skip) v = bmul(v, tw_get(r << (B - 1 - s)));
                lds_put(t_lo, t_hi, iu, fp_add(u, v));
                lds_put(t_lo, t_hi, iv, fp_sub(u, v));
            }
        }
        __syncthreads();
    }

    // ---- store (+ post-scale on the final pass), batched like the loads
    for (uint32_t e0 = tid; e0 < total; e0 += NE * nthreads) {
        Fr y[NE];
        uint32_t idx[NE];
#pragma unroll
        for (uint32_t q = 0; q < NE; q++) {
            const uint32_t e = e0 + q * nthreads;
            const uint32_t c = e & (C - 1), k = (e >> log_c) & (R - 1);
            if (!a.is_last)
                idx[q] = base + (k << a.s_log) + c;
            else
                idx[q] = ((tile_id << log_c) + c) + (k << a.t_log);
            y[q] = lds_get(t_lo, t_hi, (k << log_c) + c);
        }
        if (a.is_last && a.scale_mode == 2u) {
            // one at a time: four results are live next to the two table halves and the product
#pragma unroll
            for (uint32_t q = 0; q < NE; q++) {
                const uint32_t i = idx[q] & n_mask;
                const Fr w = fp_mul(fp_load(a.sc_lo + (i & ((1u << LO_BITS) - 1))), fp_load(a.sc_hi + (i >> LO_BITS)));
                if constexpr (LAZY) y[q] = fp_reduce_once(fp_mul_wide(y[q], w));
                else y[q] = fp_mul(y[q], w);
            }
        } else if (a.is_last && a.has_post3 && !a.hi_scaled) {
#pragma unroll
            for (uint32_t q = 0; q < NE; q++) {
                const uint32_t m = idx[q] % 3;
                Fr w;
#pragma unroll
                for (int l = 0; l < 8; l++) w.l[l] = m == 0 ? a.post3[0].l[l] : (m == 1 ? a.post3[1].l[l] : a.post3[2].l[l]);
                if constexpr (LAZY) y[q] = fp_reduce_once(fp_mul_wide(y[q], w));
                else y[q] = fp_mul(y[q], w);
            }
        } else if (LAZY && a.is_last) {
            // the transform's output is canonical (an intermediate pass hands its values on below 4p: the next pass's
            // inter-pass twiddle product takes them as they are)
#pragma unroll
            for (uint32_t q = 0; q < NE; q++) y[q] = fp_lazy_canon(y[q]);
        }
#pragma unroll
        for (uint32_t q = 0; q < NE; q++)
            if (e0 + q * nthreads < total) fp_store(out_p + idx[q], y[q]);
    }
}


// ---------------------------------------------------------------- pass geometry (shared by the plan builder and the launcher)
struct PassShape {
    uint32_t log_c, threads;
    bool radix4, lazy, fixed, cw;
};
// `avail`: the columns a tile can take -- log2 of the stride (s_log) for the passes before the last, of the DFT count for the last
static PassShape pass_shape(uint32_t L, uint32_t B, uint32_t avail) {
    static const int env_logc = getenv("H2_NTT_LOGC") ? atoi(getenv("H2_NTT_LOGC")) : -1;
    // two stages per LDS round trip: four elements per lane, half the threads per tile (H2_NTT_RADIX4=0: one stage)
    static const bool radix4 = !(getenv("H2_NTT_RADIX4") && atoi(getenv("H2_NTT_RADIX4")) == 0);
    // the lazy domain (field.hpp: values below 4p between load and store, products without their final subtraction);
    // H2_NTT_LAZY=0 keeps canonical residues everywhere -- same output either way
    static const bool lazy = !(getenv("H2_NTT_LAZY") && atoi(getenv("H2_NTT_LAZY")) == 0);
    // the common pass -- 8 bits, tiles of 4 columns, nothing skipped -- has its geometry compiled in (H2_NTT_FIXED=0: generic)
    static const bool fixed = !(getenv("H2_NTT_FIXED") && atoi(getenv("H2_NTT_FIXED")) == 0);
    // tabulated twiddles as (plain value, quotient) pairs, multiplied by fp_mul_const (H2_NTT_CONSTW=0: Montgomery tables)
    static const bool constw = !(getenv("H2_NTT_CONSTW") && atoi(getenv("H2_NTT_CONSTW")) == 0);
    PassShape sh{};
    uint32_t log_c = (B < 8) ? (10 - B) : 2;  // generic LDS radix-2 kernel: tile = R rows x C columns, about 1024 elements
    if (env_logc >= 0 && B == 8) log_c = (uint32_t)env_logc;
    if (avail < log_c) log_c = avail;
    sh.log_c = log_c;
    uint32_t threads = ((1u << B) >> 1) << log_c;
    // (transforms below 2^18 are latency-bound chains of a few tiles: more lanes per tile finish them sooner)
    sh.radix4 = radix4 && B >= 2 && threads >= 128 && L >= 18;
    if (sh.radix4) threads >>= 1;
    if (threads < 64) threads = 64;
    if (threads > 512) threads = 512;
    sh.threads = threads;
    sh.lazy = lazy;
    sh.fixed = fixed && sh.radix4 && lazy && B == 8 && log_c == 2 && threads == 256;
    sh.cw = constw && sh.radix4 && lazy;
    return sh;
}

// ---------------------------------------------------------------- plans
static std::string plan_key(uint32_t log_n, const uint64_t omega[4]) {
    char buf[128];
    snprintf(buf, sizeof buf, "%u:%016llx%016llx%016llx%016llx", log_n, (unsigned long long)omega[3],
             (unsigned long long)omega[2], (unsigned long long)omega[1], (unsigned long long)omega[0]);
    return buf;
}

Fr fr_from_u64x4(const uint64_t v[4]) {
    Fr r;
    for (int i = 0; i < 4; i++) {
        r.l[2 * i] = (uint32_t)v[i];
        r.l[2 * i + 1] = (uint32_t)(v[i] >> 32);
    }
    return r;
}

void ntt_split(uint32_t log_n, std::vector<uint32_t>& bits) {
    bits.clear();
    if (log_n == 0) return;
    // as many 8-bit passes as possible, the remainder first (it needs no inter-pass twiddle).  A remainder of ONE bit would
    // be a whole sweep over memory for a single stage (2^25, the extended domain of a k = 24 proof: 1 + 8 + 8 + 8): it is
    // folded into a 9-bit pass at the END instead (8 + 8 + 9: the middle pass keeps its 2^16-entry twiddle table; 512-row
    // tiles of 4 columns, 74 KB of LDS): 2^25 4.61 -> 4.33 ms, 2^17 47 -> 41 us.  Two 9-bit passes pay up to 2^18 (9 + 9:
    // 69 -> 62 us) but not at 2^26 (8 + 9 + 9: 8.98 -> 9.04 ms), three never.
    const uint32_t rem = log_n % 8, q = log_n / 8;
    static const bool nine = !(getenv("H2_NTT_NINE") && atoi(getenv("H2_NTT_NINE")) == 0);
    if (nine && q >= rem && (rem == 1 || (rem == 2 && log_n <= 18))) {
        for (uint32_t p = 0; p < q - rem; p++) bits.push_back(8);
        for (uint32_t p = 0; p < rem; p++) bits.push_back(9);
        return;
    }
    if (rem) bits.push_back(rem);
    for (uint32_t p = 0; p < q; p++) bits.push_back(8);
}

// ---- library memory: the optional last-pass tables are budgeted per device and evicted least-recently-used first
static std::mutex g_tab_mu;                 // guards every plan's last_direct map and DeviceCtx::ntt_last_table_bytes
static std::atomic<uint64_t> g_tick{0};
static std::atomic<size_t> g_budget_override{(size_t)-1};

static size_t parse_bytes(const char* s) {
    char* end = nullptr;
    double v = strtod(s, &end);
    if (end && (*end == 'K' || *end == 'k')) v *= 1024.0;
    if (end && (*end == 'M' || *end == 'm')) v *= 1024.0 * 1024.0;
    if (end && (*end == 'G' || *end == 'g')) v *= 1024.0 * 1024.0 * 1024.0;
    return v <= 0 ? 0 : (size_t)v;
}

void ntt_set_table_budget(size_t bytes) { g_budget_override.store(bytes); }

size_t ntt_table_budget(DeviceCtx* ctx) {
    const size_t o = g_budget_override.load();
    if (o != (size_t)-1) return o;
    static const char* env = getenv("H2_NTT_TABLE_BUDGET");
    if (env) return parse_bytes(env);
    return (size_t)ctx->prop.totalGlobalMem / 32;
}

static void free_plan(NttPlan* pl) {
    if (pl->tables) (void)hipFree(pl->tables);
    for (const Fr* t : pl->tw_direct)
        if (t) (void)hipFree(const_cast<Fr*>(t));
    for (auto& kv : pl->scaled_hi)
        if (kv.second) (void)hipFree(kv.second);
    for (auto& kv : pl->scale_tabs)
        if (kv.second.ptr) (void)hipFree(kv.second.ptr);
    for (auto& kv : pl->last_direct)
        if (kv.second.ptr) (void)hipFree(kv.second.ptr);
    delete pl;
}

// Two steps so that no lock is held across the device-wide synchronisation: `ntt_detach_idle_plans` (under the caller's
// ctx->mu) unhooks every plan nobody holds and returns their bytes; `ntt_free_plans` -- with NO lock held, the plans' device
// current -- waits for the passes already launched against their tables and frees them.
size_t ntt_detach_idle_plans(DeviceCtx* ctx, std::vector<NttPlan*>& gone) {
    size_t bytes = 0;
    std::lock_guard<std::mutex> g(g_tab_mu);
    for (auto it = ctx->plans.begin(); it != ctx->plans.end();) {
        NttPlan* pl = it->second;
        bool busy = pl->users.load() != 0;
        for (auto& kv : pl->last_direct) busy = busy || kv.second.users != 0;
        if (busy) {
            ++it;
            continue;
        }
        bytes += pl->table_bytes;
        for (auto& kv : pl->last_direct) {
            bytes += kv.second.bytes;
            ctx->ntt_last_table_bytes -= kv.second.bytes;
        }
        bytes += pl->scaled_hi.size() * ((sizeof(Fr) << pl->log_n) >> LO_BITS);
        gone.push_back(pl);
        it = ctx->plans.erase(it);
    }
    return bytes;
}

void ntt_free_plans(std::vector<NttPlan*>& gone) {
    if (gone.empty()) return;
    // nobody can reach these plans any more; passes already launched against their tables finish first
    hipError_t e = hipDeviceSynchronize();
    for (NttPlan* pl : gone) free_plan(pl);
    gone.clear();
    H2_HIP(e);
}

size_t ntt_release_plans(DeviceCtx* ctx) {
    std::vector<NttPlan*> gone;
    size_t bytes = ntt_detach_idle_plans(ctx, gone);
    ntt_free_plans(gone);
    return bytes;
}

size_t ntt_plan_bytes(DeviceCtx* ctx) {
    std::lock_guard<std::mutex> g(g_tab_mu);
    size_t bytes = 0;
    for (auto& kv : ctx->plans) {
        NttPlan* pl = kv.second;
        bytes += pl->table_bytes;
        for (auto& t : pl->last_direct) bytes += t.second.bytes;
        bytes += pl->scaled_hi.size() * ((sizeof(Fr) << pl->log_n) >> LO_BITS);
    }
    return bytes;
}

// Makes room for `need` more bytes of last-pass tables on `ctx`: idle tables leave in least-recently-used order.
// Returns false when the budget cannot hold `need` even then.  Call WITHOUT g_tab_mu.
// `allow_evict` = false: only room that is already free counts.  A table is an optimisation worth ONE product per element
// per transform; evicting one costs a device-wide synchronisation and rebuilding the other a pass over n elements, so a
// key only displaces resident tables once it has missed twice (a budget of one or two tables under a proof that cycles
// through forward / inverse transforms and their divisors would otherwise rebuild a table on every call).
static bool last_table_make_room(DeviceCtx* ctx, size_t need, bool allow_evict) {
    const size_t budget = ntt_table_budget(ctx);
    if (need > budget) return false;
    std::vector<Fr*> gone;
    {
        std::lock_guard<std::mutex> g(g_tab_mu);
        if (!allow_evict) return ctx->ntt_last_table_bytes + need <= budget;
        while (ctx->ntt_last_table_bytes + need > budget) {
            NttPlan::LastTable* victim = nullptr;
            NttPlan* owner = nullptr;
            std::string vkey;
            for (auto& pk : ctx->plans)
                for (auto& kv : pk.second->last_direct)
                    if (kv.second.ptr && kv.second.users == 0 && (!victim || kv.second.last_use < victim->last_use)) {
                        victim = &kv.second;
                        owner = pk.second;
                        vkey = kv.first;
                    }
            if (!victim) break;
            gone.push_back(victim->ptr);
            ctx->ntt_last_table_bytes -= victim->bytes;
            owner->last_direct.erase(vkey);
        }
        if (ctx->ntt_last_table_bytes + need > budget && gone.empty()) return false;
    }
    if (!gone.empty()) {
        H2_HIP(hipDeviceSynchronize());  // passes launched against an evicted table have finished before it is freed
        for (Fr* t : gone) (void)hipFree(t);
    }
    std::lock_guard<std::mutex> g(g_tab_mu);
    return ctx->ntt_last_table_bytes + need <= budget;
}

PlanRef ntt_get_plan(DeviceCtx* ctx, uint32_t log_n, const uint64_t omega[4], hipStream_t stream) {
    std::string key = plan_key(log_n, omega);
    {
        std::lock_guard<std::mutex> g(g_tab_mu);
        auto it = ctx->plans.find(key);
        if (it != ctx->plans.end()) {
            it->second->users.fetch_add(1);
            it->second->last_use = ++g_tick;
            return PlanRef(it->second);
        }
    }

    NttPlan* pl = new NttPlan();
    pl->log_n = log_n;
    ntt_split(log_n, pl->bits);
    Fr w = fr_from_u64x4(omega);
    pl->w = w;
    const uint32_t n = 1u << log_n;
    uint32_t lo_count = n < (1u << LO_BITS) ? n : (1u << LO_BITS);
    uint32_t hi_count = log_n > LO_BITS ? (n >> LO_BITS) : 0;
    size_t total = lo_count + hi_count;
    std::vector<uint32_t> bf_off;
    {
        // which passes read their tabulated twiddles as (plain, quotient) pairs: a property of the pass's geometry, fixed here
        uint32_t consumed = 0;
        for (size_t p = 0; p < pl->bits.size(); p++) {
            const uint32_t B = pl->bits[p];
            const bool last = p + 1 == pl->bits.size();
            pl->cw.push_back(pass_shape(log_n, B, last ? consumed : log_n - consumed - B).cw ? 1 : 0);
            consumed += B;
        }
    }
    for (size_t p = 0; p < pl->bits.size(); p++) {
        const uint32_t b = pl->bits[p];
        bf_off.push_back((uint32_t)total);
        total += ((1u << b) >> 1 ? (1u << b) >> 1 : 1) * (pl->cw[p] ? 2u : 1u);
    }
    H2_HIP(hipMalloc(&pl->tables, total * sizeof(Fr)));
    pl->table_bytes = total * sizeof(Fr);
    pl->tw_lo = pl->tables;
    pl->tw_hi = pl->tables + lo_count;
    hipLaunchKernelGGL(k_pow_table, dim3((lo_count + 255) / 256), dim3(256), 0, stream, pl->tables, w, 1u, lo_count, 0u);
    if (hi_count)
        hipLaunchKernelGGL(k_pow_table, dim3((hi_count + 255) / 256), dim3(256), 0, stream, pl->tables + lo_count, w,
                           1u << LO_BITS, hi_count, 0u);
    for (size_t p = 0; p < pl->bits.size(); p++) {
        uint32_t R = 1u << pl->bits[p], half = R >> 1;
        pl->tw_bfly.push_back(pl->tables + bf_off[p]);
        if (half)
            hipLaunchKernelGGL(k_pow_table, dim3((half + 255) / 256), dim3(256), 0, stream, pl->tables + bf_off[p], w,
                               n >> pl->bits[p], half, (uint32_t)pl->cw[p]);
    }
    // passes whose whole inter-pass twiddle set has <= 2^16 entries get it tabulated (2 MiB, L2-resident):
    // the pass then spends one multiplication per element on twiddles instead of two
    {
        uint32_t consumed = 0;
        for (size_t p = 0; p < pl->bits.size(); p++) {
            const uint32_t B = pl->bits[p];
            const bool last = p + 1 == pl->bits.size();
            Fr* tab = nullptr;
            if (p > 0 && !last && B + consumed <= 16) {
                uint32_t cnt = 1u << (B + consumed);
                const size_t tab_bytes = (size_t)cnt * sizeof(Fr) * (pl->cw[p] ? 2 : 1);
                H2_HIP(hipMalloc(&tab, tab_bytes));
                pl->table_bytes += tab_bytes;
                hipLaunchKernelGGL(k_direct_table, dim3((cnt + 255) / 256), dim3(256), 0, stream, tab, w, consumed,
                                   log_n - consumed - B, log_n, cnt, (uint32_t)pl->cw[p]);
            }
            pl->tw_direct.push_back(tab);
            consumed += B;
        }
    }
    H2_HIP(hipGetLastError());
    // the tables are complete before the plan is published: a second caller on another stream (h2_dev_* on a different
    // torch stream, or the host API after a device-API first use) must not launch passes against tables still being
    // written.  Once per (log_n, omega) for the life of the process.
    H2_HIP(hipStreamSynchronize(stream));
    std::lock_guard<std::mutex> g(g_tab_mu);
    auto raced = ctx->plans.find(key);
    if (raced != ctx->plans.end()) {
        // another host-API slot of this device built the same plan meanwhile: keep the published one (this one's tables are
        // complete and nobody else has seen them: freed at once)
        free_plan(pl);
        raced->second->users.fetch_add(1);
        raced->second->last_use = ++g_tick;
        return PlanRef(raced->second);
    }
    pl->users.fetch_add(1);
    pl->last_use = ++g_tick;
    ctx->plans[key] = pl;
    return PlanRef(pl);
}

// Runs the transform.  `src` (in_len valid elements, zero-extended to n) -> result in `dst`.
// `tmp` is an n-element scratch; src may equal dst (then tmp must be distinct from both).
static void set_scale3(PassArgs& a, const Fr* pre3, const Fr* post3) {
    a.has_pre3 = pre3 != nullptr;
    a.has_post3 = post3 != nullptr;
    a.post3_uniform = post3 && fp_eq(post3[0], post3[1]) && fp_eq(post3[0], post3[2]);
    for (int i = 0; i < 3; i++) {
        if (pre3) a.pre3[i] = pre3[i];
        if (post3) a.post3[i] = post3[i];
    }
}

static void ntt_run_chunk(DeviceCtx* ctx, NttPlan* pl, const Fr* const* srcs, Fr* const* dsts, Fr* const* tmps, uint32_t cnt,
                          uint32_t in_len, const Fr* pre3, const Fr* post3, hipStream_t stream, const Fr* scale_tab,
                          uint32_t scale_mode);

// The two-level table of g^i (i < 2^log_n) with `d` folded into the high level, cached with the plan: the coset transforms
// of a proof use quotient_poly_degree generators per direction, again and again.
ScaleTabRef ntt_scale_table(NttPlan* pl, const Fr& g, const Fr* d, hipStream_t stream) {
    char kb[200];
    int at = 0;
    for (int l = 7; l >= 0; l--) at += snprintf(kb + at, sizeof kb - at, "%08x", g.l[l]);
    kb[at++] = d ? '*' : '.';
    if (d)
        for (int l = 7; l >= 0; l--) at += snprintf(kb + at, sizeof kb - at, "%08x", d->l[l]);
    const std::string key(kb, (size_t)at);
    {
        std::lock_guard<std::mutex> lk(pl->mu);
        auto it = pl->scale_tabs.find(key);
        if (it != pl->scale_tabs.end()) {
            it->second.users++;
            it->second.last_use = ++pl->scale_clock;
            return ScaleTabRef(pl, &it->second);
        }
    }
    // built OUTSIDE the plan's lock (the other host-API slot of the device keeps transforming meanwhile); the device
    // block is owned by a guard until it is published
    const uint32_t n = 1u << pl->log_n;
    const uint32_t lo_count = n < (1u << LO_BITS) ? n : (1u << LO_BITS);
    const uint32_t hi_count = pl->log_n > LO_BITS ? (n >> LO_BITS) : 1u;
    const size_t bytes = ((size_t)(1u << LO_BITS) + hi_count) * sizeof(Fr);
    struct Block {
        Fr* p = nullptr;
        ~Block() {
            if (p) (void)hipFree(p);
        }
    } block;
    H2_HIP(hipMalloc((void**)&block.p, bytes));
    hipLaunchKernelGGL(k_pow_table, dim3((lo_count + 255) / 256), dim3(256), 0, stream, block.p, g, 1u, lo_count, 0u);
    Fr* hi = block.p + (1u << LO_BITS);
    hipLaunchKernelGGL(k_pow_table, dim3((hi_count + 255) / 256), dim3(256), 0, stream, hi, g, 1u << LO_BITS, hi_count, 0u);
    if (d) hipLaunchKernelGGL(k_scale_table, dim3((hi_count + 255) / 256), dim3(256), 0, stream, hi, hi, *d, hi_count);
    H2_HIP(hipGetLastError());
    H2_HIP(hipStreamSynchronize(stream));  // complete before other streams can find it (once per generator)
    std::vector<Fr*> evicted;
    size_t evicted_bytes = 0;
    NttPlan::ScaleTab* entry = nullptr;
    {
        std::lock_guard<std::mutex> lk(pl->mu);
        auto it = pl->scale_tabs.find(key);
        if (it != pl->scale_tabs.end()) {  // another caller built the same table meanwhile: theirs stays, ours goes with `block`
            it->second.users++;
            it->second.last_use = ++pl->scale_clock;
            return ScaleTabRef(pl, &it->second);
        }
        while (pl->scale_tabs.size() >= NttPlan::SCALE_TABS_MAX) {
            auto lru = pl->scale_tabs.end();
            for (auto jt = pl->scale_tabs.begin(); jt != pl->scale_tabs.end(); ++jt)
                if (jt->second.users == 0 && (lru == pl->scale_tabs.end() || jt->second.last_use < lru->second.last_use)) lru = jt;
            if (lru == pl->scale_tabs.end()) break;  // every table is held: over the cap for now
            evicted.push_back(lru->second.ptr);
            evicted_bytes += lru->second.bytes;
            pl->scale_tabs.erase(lru);
        }
        NttPlan::ScaleTab& e = pl->scale_tabs[key];
        e.ptr = block.p;
        e.bytes = bytes;
        e.users = 1;
        e.last_use = ++pl->scale_clock;
        block.p = nullptr;
        entry = &e;
    }
    {
        std::lock_guard<std::mutex> g2(g_tab_mu);
        pl->table_bytes += bytes;
        pl->table_bytes -= std::min(pl->table_bytes, evicted_bytes);
    }
    for (Fr* q : evicted) (void)hipFree(q);  // (waits for the passes already launched against it)
    return ScaleTabRef(pl, entry);
}

void ntt_run(DeviceCtx* ctx, NttPlan* pl, const Fr* src, Fr* dst, Fr* tmp, uint32_t in_len, const Fr* pre3,
             const Fr* post3, hipStream_t stream, const Fr* scale_tab, uint32_t scale_mode) {
    ntt_run_many(ctx, pl, &src, &dst, &tmp, 1, in_len, pre3, post3, stream, scale_tab, scale_mode);
}

// `count` transforms of one plan (same size, root, scales): chunks of NTT_BATCH_MAX vectors per launch.  tmps[i]: the
// scratch of vector i (distinct per vector of a chunk; needed when the plan has >= 2 passes).
void ntt_run_many(DeviceCtx* ctx, NttPlan* pl, const Fr* const* srcs, Fr* const* dsts, Fr* const* tmps, size_t count,
                  uint32_t in_len, const Fr* pre3, const Fr* post3, hipStream_t stream, const Fr* scale_tab,
                  uint32_t scale_mode) {
    for (size_t c0 = 0; c0 < count; c0 += NTT_BATCH_MAX) {
        const uint32_t cnt = (uint32_t)std::min<size_t>(NTT_BATCH_MAX, count - c0);
        ntt_run_chunk(ctx, pl, srcs + c0, dsts + c0, tmps + c0, cnt, in_len, pre3, post3, stream, scale_tab, scale_mode);
    }
}

static void ntt_run_chunk(DeviceCtx* ctx, NttPlan* pl, const Fr* const* srcs, Fr* const* dsts, Fr* const* tmps, uint32_t cnt,
                          uint32_t in_len, const Fr* pre3, const Fr* post3, hipStream_t stream, const Fr* scale_tab,
                          uint32_t scale_mode) {
    const Fr* const src = srcs[0];
    Fr* const dst = dsts[0];
    Fr* const tmp = tmps[0];
    const uint32_t L = pl->log_n;
    if (L == 0) {
        // n = 1: X[0] = x[0] (times post3[0])
        PassArgs a{};
        a.in = src; a.out = dst; a.tw_bfly = pl->tables; a.tw_lo = pl->tw_lo; a.tw_hi = pl->tw_hi;
        set_scale3(a, pre3, post3); a.log_n = 0; a.B = 0; a.s_log = 0; a.t_log = 0; a.nprev = 0;
        a.is_last = 1; a.in_len = in_len; a.log_c = 0;
        a.sc_lo = scale_tab; a.sc_hi = scale_tab ? scale_tab + (1u << LO_BITS) : nullptr; a.scale_mode = scale_tab ? scale_mode : 0u;
        if (cnt > 1) {
            a.batch = cnt;
            for (uint32_t i = 0; i < cnt; i++) { a.in_b[i] = srcs[i]; a.out_b[i] = dsts[i]; }
        }
        hipLaunchKernelGGL((k_ntt_pass<false, false>), dim3(1, cnt), dim3(64), 4 * sizeof(Fr), stream, a);
        H2_HIP(hipGetLastError());
        return;
    }
    const size_t P = pl->bits.size();
    // buffer chain: pass 0 reads src; intermediate passes run in place on `work`; last pass writes dst.
    // With P == 1 the single (last) pass goes src -> dst through LDS (safe in place: one tile per DFT...
    // but tiles of other DFTs do not exist when P == 1, so src == dst is fine).
    Fr* work = (P >= 2) ? tmp : nullptr;
    uint32_t consumed = 0;
    for (size_t p = 0; p < P; p++) {
        PassArgs a{};
        const uint32_t B = pl->bits[p];
        const bool last = (p + 1 == P);
        a.in = (p == 0) ? src : work;
        a.out = last ? dst : work;
        if (cnt > 1) {
            a.batch = cnt;
            for (uint32_t i = 0; i < cnt; i++) {
                Fr* const work_i = (P >= 2) ? tmps[i] : nullptr;
                a.in_b[i] = (p == 0) ? srcs[i] : work_i;
                a.out_b[i] = last ? dsts[i] : work_i;
            }
        }
        a.tw_bfly = pl->tw_bfly[p];
        a.tw_lo = pl->tw_lo;
        a.tw_hi = pl->tw_hi;
        a.tw_direct = pl->tw_direct[p];
        set_scale3(a, (p == 0) ? pre3 : nullptr, last ? post3 : nullptr);
        a.log_n = L;
        a.B = B;
        a.s_log = L - consumed - B;
        a.t_log = consumed;
        a.nprev = (uint32_t)p;
        uint32_t t = 0;
        for (size_t q = 0; q < p; q++) {
            a.prevB[q] = pl->bits[q];
            a.prevT[q] = t;
            t += pl->bits[q];
        }
        a.is_last = last ? 1 : 0;
        a.in_len = (p == 0) ? in_len : (1u << L);
        a.sc_lo = scale_tab;
        a.sc_hi = scale_tab ? scale_tab + (1u << LO_BITS) : nullptr;
        a.scale_mode = (scale_tab && ((scale_mode == 1u && p == 0) || (scale_mode == 2u && last))) ? scale_mode : 0u;
        if (p == 0 && !last && in_len && in_len < (1u << L) && (in_len & (in_len - 1)) == 0 && getenv("H2_NTT_NO_ZSKIP") == nullptr) {
            uint32_t z = 0;
            while ((in_len << z) < (1u << L)) z++;  // padded by 2^z
            a.zskip = z < B ? z : B;               // in_len = (R >> z) * S rows exactly when z <= B
        }
        if (last && p > 0 && a.post3_uniform && L > LO_BITS) {
            // iNTT: fold the divisor into the high twiddle table used by the last pass's inter-pass twiddles
            char key[80];
            snprintf(key, sizeof key, "%08x%08x%08x%08x%08x%08x%08x%08x", post3[0].l[7], post3[0].l[6], post3[0].l[5],
                     post3[0].l[4], post3[0].l[3], post3[0].l[2], post3[0].l[1], post3[0].l[0]);
            Fr* scaled = nullptr;
            {
                std::lock_guard<std::mutex> g(pl->mu);
                auto it = pl->scaled_hi.find(key);
                if (it == pl->scaled_hi.end()) {
                    uint32_t cnt = (1u << L) >> LO_BITS;
                    H2_HIP(hipMalloc(&scaled, cnt * sizeof(Fr)));
                    hipLaunchKernelGGL(k_scale_table, dim3((cnt + 255) / 256), dim3(256), 0, stream, scaled, pl->tw_hi,
                                       post3[0], cnt);
                    H2_HIP(hipStreamSynchronize(stream));  // complete before other streams can find it (once per divisor)
                    pl->scaled_hi[key] = scaled;
                } else {
                    scaled = it->second;
                }
            }
            a.tw_hi = scaled;
            a.hi_scaled = 1;
        }
        // The last pass of a large transform reads its inter-pass twiddles from a complete table (32 B x n, streamed in the
        // order of its loads) instead of composing each from two: one product per element instead of two, on a pass that
        // is bound by VALU issue and has the HBM time to spare (2^24: 1.84 -> see DESIGN 3.2).  One table per divisor
        // folded into it, inside the per-device budget (ntt_table_budget: the least recently used idle table leaves
        // first); H2_NTT_LAST_TABLE=0, no room in the budget or a failed allocation leave the lo x hi form.
        static const bool last_table = !(getenv("H2_NTT_LAST_TABLE") && atoi(getenv("H2_NTT_LAST_TABLE")) == 0);
        static const uint32_t last_table_max = getenv("H2_NTT_LAST_TABLE_MAX_LOG") ? (uint32_t)atoi(getenv("H2_NTT_LAST_TABLE_MAX_LOG")) : 26u;
        NttPlan::LastTable* used_table = nullptr;
        // the pin is dropped when this pass has been launched -- or when anything on the way there throws (H2_HIP): a leaked
        // pin would keep the table from ever being evicted and its plan from ever being released
        struct Pin {
            NttPlan::LastTable* t = nullptr;
            ~Pin() {
                if (t) {
                    std::lock_guard<std::mutex> g(g_tab_mu);
                    t->users--;
                }
            }
        } pinned;
        if (last && p > 0 && last_table && L >= 18 && L <= last_table_max) {
            const bool scaled = a.hi_scaled != 0;
            std::string key;
            if (scaled) {
                char kb[80];
                snprintf(kb, sizeof kb, "%08x%08x%08x%08x%08x%08x%08x%08x", post3[0].l[7], post3[0].l[6], post3[0].l[5],
                         post3[0].l[4], post3[0].l[3], post3[0].l[2], post3[0].l[1], post3[0].l[0]);
                key = kb;
            }
            const size_t bytes = sizeof(Fr) << L;   // (Montgomery form also under CW: see k_ntt_pass's DP)
            auto pin = [&]() -> NttPlan::LastTable* {  // with g_tab_mu held
                auto it = pl->last_direct.find(key);
                if (it == pl->last_direct.end() || it->second.ptr == nullptr) return nullptr;
                it->second.users++;
                it->second.last_use = ++g_tick;
                return &it->second;
            };
            bool may_evict = false;
            {
                std::lock_guard<std::mutex> g(g_tab_mu);
                used_table = pin();
                if (!used_table) may_evict = ++pl->last_misses[key] >= 2;   // see last_table_make_room
            }
            pinned.t = used_table;
            if (!used_table && last_table_make_room(ctx, bytes, may_evict)) {
                // built outside the lock (a table is 0.5 .. 2 GiB of powers); a second builder of the same table loses
                Fr* tab = nullptr;
                if (hipMalloc(&tab, bytes) != hipSuccess) {
                    (void)hipGetLastError();  // no room on the device: this transform composes its twiddles
                    tab = nullptr;
                } else {
                    hipLaunchKernelGGL(k_last_table, dim3((1u << L) / 256), dim3(256), 0, stream, tab, pl->w, B, L,
                                       scaled ? post3[0] : pl->w, scaled ? 1u : 0u, 0u);
                    H2_HIP(hipStreamSynchronize(stream));  // complete before other streams can find it
                }
                Fr* loser = nullptr;
                {
                    std::lock_guard<std::mutex> g(g_tab_mu);
                    used_table = pin();
                    if (used_table) {
                        loser = tab;
                    } else if (tab) {
                        NttPlan::LastTable& e = pl->last_direct[key];
                        e.ptr = tab;
                        e.bytes = bytes;
                        ctx->ntt_last_table_bytes += bytes;
                        pl->last_misses[key] = 0;  // evicted later, it has to miss twice again before it displaces others
                        used_table = pin();
                    }
                }
                if (loser) (void)hipFree(loser);
                pinned.t = used_table;
            }
            if (used_table != nullptr) {
                a.tw_direct = used_table->ptr;
                a.direct_kmajor = 1;
            }
        }
        {
            const PassShape sh = pass_shape(L, B, last ? consumed : a.s_log);
            const bool cw = pl->cw[p] != 0;   // (== sh.cw: the plan's tables were built for it)
            a.log_c = sh.log_c;
            a.radix4 = sh.radix4 ? 1u : 0u;
            const uint32_t R = 1u << B, C = 1u << sh.log_c, threads = sh.threads;
            const uint32_t ntiles = (1u << L) / (R * C);
            // tile planes + butterfly twiddles: R/2 values and a pad, or (CW) R/2 pairs exactly -- 40 KiB for the common pass
            const size_t lds = cw ? ((size_t)R * C + R) * sizeof(Fr) : ((size_t)R * C + (R >> 1) + 2) * sizeof(Fr);
            if (lds > 64 * 1024) {  // beyond the default dynamic LDS limit: raise it once
                static bool raised[64] = {};  // per device
                const int dev = ctx->device;
                if (dev < 0 || dev >= 64 || !raised[dev]) {
                    H2_HIP(hipFuncSetAttribute((const void*)k_ntt_pass<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                    H2_HIP(hipFuncSetAttribute((const void*)k_ntt_pass<true, true, 0, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                    H2_HIP(hipFuncSetAttribute((const void*)k_ntt_pass<true, true, 0, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                    H2_HIP(hipFuncSetAttribute((const void*)k_ntt_pass<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                    H2_HIP(hipFuncSetAttribute((const void*)k_ntt_pass<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                    H2_HIP(hipFuncSetAttribute((const void*)k_ntt_pass<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                    if (dev >= 0 && dev < 64) raised[dev] = true;
                }
            }
            const bool dp = cw && !last;      // pairs in tw_direct: the middle passes' tables (a first pass has none)
            if (cw && sh.fixed && a.zskip == 0 && dp)
                hipLaunchKernelGGL((k_ntt_pass<true, true, 8, true, true>), dim3(ntiles, cnt), dim3(threads), lds, stream, a);
            else if (cw && sh.fixed && a.zskip == 0)
                hipLaunchKernelGGL((k_ntt_pass<true, true, 8, true, false>), dim3(ntiles, cnt), dim3(threads), lds, stream, a);
            else if (cw && dp)
                hipLaunchKernelGGL((k_ntt_pass<true, true, 0, true, true>), dim3(ntiles, cnt), dim3(threads), lds, stream, a);
            else if (cw)
                hipLaunchKernelGGL((k_ntt_pass<true, true, 0, true, false>), dim3(ntiles, cnt), dim3(threads), lds, stream, a);
            else if (sh.fixed && a.zskip == 0)
                hipLaunchKernelGGL((k_ntt_pass<true, true, 8>), dim3(ntiles, cnt), dim3(threads), lds, stream, a);
            else if (a.radix4 && sh.lazy)
                hipLaunchKernelGGL((k_ntt_pass<true, true>), dim3(ntiles, cnt), dim3(threads), lds, stream, a);
            else if (a.radix4)
                hipLaunchKernelGGL((k_ntt_pass<true, false>), dim3(ntiles, cnt), dim3(threads), lds, stream, a);
            else if (sh.lazy)
                hipLaunchKernelGGL((k_ntt_pass<false, true>), dim3(ntiles, cnt), dim3(threads), lds, stream, a);
            else
                hipLaunchKernelGGL((k_ntt_pass<false, false>), dim3(ntiles, cnt), dim3(threads), lds, stream, a);
        }
        // (`pinned` unpins here: launched -- an eviction from here on synchronises the device before it frees)
        consumed += B;
    }
    H2_HIP(hipGetLastError());
}

}  // namespace h2
