// poly.hip -- batched polynomial arithmetic as fused elementwise kernels (HBM-bound:
// 32 B per operand per element, one 2 x dwordx4 access per lane).
//
// Replaces the ec-gpu-gen kernels named at the reference's launch sites:
//   eval_mul_c / eval_sum_c / eval_sum / eval_mul / eval_lctheta / eval_lcbeta / eval_addgamma /
//   eval_constant  (plonk/evaluation_gpu.rs:148-163,202-217,246-259,279-305,560,579-585,622-669)
//   batch_mont / batch_unmont  (arithmetic.rs:235-241,280-286)
// and the CPU loops Polynomial +,-,*scalar (poly.rs:191-257) and
// divide_by_vanishing_poly (poly/domain.rs:354-373).
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include <vector>
#include "common.hpp"
#include "poly.hpp"

namespace h2 {

__device__ __forceinline__ size_t rot_index(size_t i, int32_t rot, size_t size) {
    // (i + rot) mod size for |rot| < size   (get_rotation_idx, plonk/evaluation.rs:40-42)
    long long v = (long long)i + rot;
    if (v < 0) v += (long long)size;
    if (v >= (long long)size) v -= (long long)size;
    return (size_t)v;
}

template <int OP>
__global__ void __launch_bounds__(256) k_eval_op(Fr* res, const Fr* l, const Fr* r, int32_t l_rot, int32_t r_rot,
                                                 size_t size, Fr c) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < size; i += stride) {
        Fr lv, rv, out;
        if (OP != H2_OP_CONSTANT) lv = fp_load(l + rot_index(i, l_rot, size));
        if (OP == H2_OP_SUM || OP == H2_OP_MUL || OP == H2_OP_SUB || OP == H2_OP_LCTHETA || OP == H2_OP_LCBETA)
            rv = fp_load(r + rot_index(i, r_rot, size));
        if (OP == H2_OP_MUL_C) out = fp_mul(lv, c);
        else if (OP == H2_OP_SUM_C || OP == H2_OP_ADDGAMMA) out = fp_add(lv, c);
        else if (OP == H2_OP_SUM) out = fp_add(lv, rv);
        else if (OP == H2_OP_MUL) out = fp_mul(lv, rv);
        else if (OP == H2_OP_SUB) out = fp_sub(lv, rv);
        else if (OP == H2_OP_LCTHETA) out = fp_add(fp_mul(lv, c), rv);
        else if (OP == H2_OP_LCBETA) out = fp_mul(fp_add(lv, c), rv);
        else out = c;
        fp_store(res + i, out);
    }
}

// One element per lane, the whole range as ONE grid: workgroups are dispatched in index order, so the chip works on a
// moving window of neighbouring DRAM pages.  A capped grid whose workgroups stride across the whole operand keeps every
// page of it open at once: measured with plain 16-byte kernels (tools/membench.hip, 1 GiB operands) 2 reads + 1 write run
// at 6.1 TB/s as one grid and at 4.8-5.6 TB/s as 4-32 grid-striding workgroups per CU (writes alone: 6.9 vs 4.4 TB/s).
// The kernels keep their grid-stride loops for ranges beyond 2^31 workgroups.
static unsigned grid_for(size_t n) {
    size_t blocks = (n + 255) / 256;
    if (blocks > 0x7fffffffull) blocks = 0x7fffffffull;
    if (blocks == 0) blocks = 1;
    return (unsigned)blocks;
}

static Fr fr_host(const uint64_t v[4]) {
    Fr r;
    for (int i = 0; i < 4; i++) {
        r.l[2 * i] = (uint32_t)v[i];
        r.l[2 * i + 1] = (uint32_t)(v[i] >> 32);
    }
    return r;
}

int eval_op_launch(int op, Fr* res, const Fr* l, const Fr* r, int32_t l_rot, int32_t r_rot, size_t size,
                   const uint64_t c[4], hipStream_t stream) {
    if (size == 0) return H2_OK;
    bool need_l = op != H2_OP_CONSTANT;
    bool need_r = op == H2_OP_SUM || op == H2_OP_MUL || op == H2_OP_SUB || op == H2_OP_LCTHETA || op == H2_OP_LCBETA;
    bool need_c = !(op == H2_OP_SUM || op == H2_OP_MUL || op == H2_OP_SUB);
    if (!res || (need_l && !l) || (need_r && !r) || (need_c && !c)) {
        set_last_error("h2_eval_op: missing operand");
        return H2_ERR_INVALID;
    }
    // in-place is only defined for an un-rotated operand (evaluation_gpu.rs:631-639)
    if ((res == l && need_l && l_rot != 0) || (res == r && need_r && r_rot != 0)) {
        set_last_error("h2_eval_op: result aliases a rotated operand");
        return H2_ERR_INVALID;
    }
    // normalise rotations into (-size, size)
    long long sz = (long long)size;
    l_rot = (int32_t)(((long long)l_rot % sz));
    r_rot = (int32_t)(((long long)r_rot % sz));
    Fr cv = c ? fr_host(c) : Fr{};
    dim3 g(grid_for(size)), b(256);
#define H2_CASE(OPC) \
    case OPC: hipLaunchKernelGGL(k_eval_op<OPC>, g, b, 0, stream, res, l, r, l_rot, r_rot, size, cv); break;
    switch (op) {
        H2_CASE(H2_OP_MUL_C)
        H2_CASE(H2_OP_SUM_C)
        H2_CASE(H2_OP_SUM)
        H2_CASE(H2_OP_MUL)
        H2_CASE(H2_OP_SUB)
        H2_CASE(H2_OP_LCTHETA)
        H2_CASE(H2_OP_LCBETA)
        H2_CASE(H2_OP_ADDGAMMA)
        H2_CASE(H2_OP_CONSTANT)
        default: set_last_error("h2_eval_op: unknown op"); return H2_ERR_INVALID;
    }
#undef H2_CASE
    H2_HIP(hipGetLastError());
    return H2_OK;
}

// a[i] *= t[i % t_len]; t_len is a power of two (2^(extended_k - k)), table read through L1/L2
__global__ void __launch_bounds__(256) k_divide_by_vanishing(Fr* a, size_t size, const Fr* t, size_t t_mask) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < size; i += stride)
        fp_store(a + i, fp_mul(fp_load(a + i), fp_load(t + (i & t_mask))));
}

int divide_by_vanishing_launch(Fr* a, size_t size, const Fr* t, size_t t_len, hipStream_t stream) {
    if (size == 0) return H2_OK;
    if (!a || !t || t_len == 0 || (t_len & (t_len - 1)) != 0) {
        set_last_error("h2_divide_by_vanishing_poly: t_len must be a non-zero power of two");
        return H2_ERR_INVALID;
    }
    hipLaunchKernelGGL(k_divide_by_vanishing, dim3(grid_for(size)), dim3(256), 0, stream, a, size, t, t_len - 1);
    H2_HIP(hipGetLastError());
    return H2_OK;
}

template <bool TO_MONT>
__global__ void __launch_bounds__(256) k_batch_mont(Fr* a, size_t n) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        Fr v = fp_load(a + i);
        fp_store(a + i, TO_MONT ? fp_to_mont(v) : fp_from_mont(v));
    }
}

int batch_mont_launch(Fr* a, size_t n, bool to_mont, hipStream_t stream) {
    if (n == 0) return H2_OK;
    if (to_mont)
        hipLaunchKernelGGL(k_batch_mont<true>, dim3(grid_for(n)), dim3(256), 0, stream, a, n);
    else
        hipLaunchKernelGGL(k_batch_mont<false>, dim3(grid_for(n)), dim3(256), 0, stream, a, n);
    H2_HIP(hipGetLastError());
    return H2_OK;
}


// ---------------------------------------------------------------- compact witness columns
// A witness column whose values fit 64 bits (booleans, bytes, 16-bit limbs, 48-bit products: most of a zkWasm-shaped trace)
// crosses PCIe as 8 bytes per cell instead of 32 and is widened here to canonical 4 x u64 scalars: dst[i] = {src[i], 0, 0, 0}.
__global__ void __launch_bounds__(256) k_widen_u64(const uint64_t* src, size_t n, uint4* dst) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    // one lane per 16-byte half of an element: the stores of a wave cover 1 KiB contiguously
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < 2 * n; t += stride) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if ((t & 1) == 0) {
            const uint64_t x = src[t >> 1];
            v.x = (uint32_t)x;
            v.y = (uint32_t)(x >> 32);
        }
        dst[t] = v;
    }
}

int widen_u64_launch(const uint64_t* src, size_t n, Fr* dst, hipStream_t stream) {
    if (n == 0) return H2_OK;
    hipLaunchKernelGGL(k_widen_u64, dim3(grid_for(2 * n)), dim3(256), 0, stream, src, n, (uint4*)dst);
    H2_HIP(hipGetLastError());
    return H2_OK;
}

// ---------------------------------------------------------------- find_max_scalar_bits for a group of columns
// plonk/prover.rs:237-254 takes the maximum of a column and its bit length; the bit length of the maximum is the bit
// length of the OR of all values, and an OR needs no ordering: one launch ORs the limbs of up to 16 canonical columns
// (blockIdx.y = column) into 8 words each.
static constexpr int MAXBITS_COLS = 16;
struct MaxBitsArgs {
    const Fr* col[MAXBITS_COLS];
};

__global__ void __launch_bounds__(256) k_or_limbs(MaxBitsArgs a, size_t n, uint32_t* words) {
    __shared__ uint32_t sh[8];
    const Fr* col = a.col[blockIdx.y];
    if (threadIdx.x < 8) sh[threadIdx.x] = 0;
    __syncthreads();
    uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const Fr v = fp_load(col + i);
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] |= v.l[k];
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) acc[k] |= __shfl_xor(acc[k], off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (acc[k]) atomicOr(&sh[k], acc[k]);
    }
    __syncthreads();
    if (threadIdx.x < 8 && sh[threadIdx.x]) atomicOr(&words[blockIdx.y * 8 + threadIdx.x], sh[threadIdx.x]);
}

int max_scalar_bits_launch(const Fr* const* d_cols, size_t count, size_t n, uint32_t* d_words, uint32_t* out_bits,
                           hipStream_t stream) {
    if (count == 0) return H2_OK;
    H2_HIP(hipMemsetAsync(d_words, 0, count * 8 * sizeof(uint32_t), stream));
    if (n) {
        unsigned blocks = (unsigned)std::min<size_t>((n + 2047) / 2048, 256);  // 8 global atomics per workgroup
        for (size_t c0 = 0; c0 < count; c0 += MAXBITS_COLS) {
            MaxBitsArgs a{};
            const size_t m = std::min<size_t>(MAXBITS_COLS, count - c0);
            for (size_t j = 0; j < m; j++) a.col[j] = d_cols[c0 + j];
            hipLaunchKernelGGL(k_or_limbs, dim3(blocks, (unsigned)m), dim3(256), 0, stream, a, n, d_words + c0 * 8);
        }
        H2_HIP(hipGetLastError());
    }
    std::vector<uint32_t> words(count * 8);
    H2_HIP(hipMemcpyAsync(words.data(), d_words, count * 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    H2_HIP(hipStreamSynchronize(stream));
    for (size_t c = 0; c < count; c++) {
        uint32_t bits = 0;
        for (int k = 7; k >= 0; k--)
            if (words[c * 8 + k]) {
                bits = 32 * k + (32 - __builtin_clz(words[c * 8 + k]));
                break;
            }
        out_bits[c] = bits;
    }
    return H2_OK;
}

// ---------------------------------------------------------------- eval_polynomial (Horner) on device
// arithmetic.rs:707-735 evaluates p(x) by per-thread Horner over contiguous chunks + powers of x.  Here a
// workgroup folds 4096 coefficients: lane t runs Horner in x^256 over a[t], a[t+256], ... (coalesced loads),
// then an 8-level LDS tree combines the 256 lanes with x, x^2, x^4, ...  The per-workgroup values form a
// 4096-times shorter polynomial in x^4096, folded again by the same kernel until one element is left.
struct EvalPolyArgs {
    const Fr* in;
    Fr* out;
    size_t n;
    Fr x256;     // x^256
    Fr xpow[8];  // x^(2^l), l = 0..7
};

__device__ __forceinline__ void eval_poly_block(const EvalPolyArgs& a, uint4* sh_lo, uint4* sh_hi) {
    const uint32_t t = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * 4096;
    Fr acc = fp_zero<FrParams>();
#pragma unroll 1
    for (int j = 15; j >= 0; j--) {
        size_t i = base + (size_t)j * 256 + t;
        Fr c = (i < a.n) ? fp_load(a.in + i) : fp_zero<FrParams>();
        acc = fp_add(fp_mul(acc, a.x256), c);
    }
    // tree: after level l, lane t (multiple of 2^(l+1)) holds sum_{u < 2^(l+1)} v[t+u] x^u
    for (int l = 0; l < 8; l++) {
        sh_lo[t] = make_uint4(acc.l[0], acc.l[1], acc.l[2], acc.l[3]);
        sh_hi[t] = make_uint4(acc.l[4], acc.l[5], acc.l[6], acc.l[7]);
        __syncthreads();
        const uint32_t step = 1u << l;
        if ((t & (2 * step - 1)) == 0) {
            uint4 lo = sh_lo[t + step], hi = sh_hi[t + step];
            Fr o;
            o.l[0] = lo.x; o.l[1] = lo.y; o.l[2] = lo.z; o.l[3] = lo.w;
            o.l[4] = hi.x; o.l[5] = hi.y; o.l[6] = hi.z; o.l[7] = hi.w;
            acc = fp_add(acc, fp_mul(o, a.xpow[l]));
        }
        __syncthreads();
    }
    if (t == 0) fp_store(a.out + blockIdx.x, acc);
}

__global__ void __launch_bounds__(256) k_eval_poly(EvalPolyArgs a) {
    __shared__ uint4 sh_lo[256], sh_hi[256];
    eval_poly_block(a, sh_lo, sh_hi);
}

// blockIdx.y = evaluation: one level of EVERY evaluation of a batch in one launch (table in device memory)
__global__ void __launch_bounds__(256) k_eval_poly_multi(const EvalPolyArgs* table) {
    __shared__ uint4 sh_lo[256], sh_hi[256];
    __shared__ EvalPolyArgs a;
    const uint32_t* src = (const uint32_t*)(table + blockIdx.y);
    for (uint32_t k = threadIdx.x; k < sizeof(EvalPolyArgs) / 4; k += 256) ((uint32_t*)&a)[k] = src[k];
    __syncthreads();
    if ((size_t)blockIdx.x * 4096 >= a.n) return;
    eval_poly_block(a, sh_lo, sh_hi);
}

// result (host) = sum_i poly[i] * x^i.  d_tmp: ceil(n / 4096) + ceil(n / 4096^2) + 2 elements of scratch.
// enqueue the levels of one evaluation; the value lands in d_tmp's last level, whose address is returned
static Fr* eval_polynomial_enqueue(const Fr* d_poly, size_t n, const uint64_t point[4], Fr* d_tmp, hipStream_t stream) {
    Fr x = fr_host(point);
    const Fr* in = d_poly;
    Fr* dst = d_tmp;
    size_t cnt = n;
    for (;;) {
        EvalPolyArgs a;
        a.in = in;
        a.out = dst;
        a.n = cnt;
        Fr p = x;
        for (int l = 0; l < 8; l++) {
            a.xpow[l] = p;
            p = fp_sqr(p);
        }
        a.x256 = p;  // x^256
        size_t blocks = (cnt + 4095) / 4096;
        hipLaunchKernelGGL(k_eval_poly, dim3((unsigned)blocks), dim3(256), 0, stream, a);
        if (blocks == 1) break;
        // next level: polynomial in x^4096 over the block values
        for (int l = 0; l < 4; l++) p = fp_sqr(p);  // x^256 -> x^4096
        x = p;
        in = dst;
        dst = dst + blocks;
        cnt = blocks;
    }
    return dst;
}

int eval_polynomial_launch(const Fr* d_poly, size_t n, const uint64_t point[4], Fr* d_tmp, uint64_t out[4],
                           hipStream_t stream) {
    if (n == 0) {
        memset(out, 0, 32);
        return H2_OK;
    }
    Fr* dst = eval_polynomial_enqueue(d_poly, n, point, d_tmp, stream);
    H2_HIP(hipGetLastError());
    H2_HIP(hipMemcpyAsync(out, dst, 32, hipMemcpyDeviceToHost, stream));
    H2_HIP(hipStreamSynchronize(stream));
    return H2_OK;
}

// `count` evaluations (polynomial j at point j), one read-back and one synchronisation for all of them: the prover
// evaluates every committed polynomial at x, omega x, ... (plonk/prover.rs:700-790, a rayon par_iter over
// eval_polynomial_st there).  Every level of the fold is ONE launch over all evaluations (a single evaluation is a
// latency-bound chain on a quarter of the chip; sixty of them back to back were 2 ms of a 36 ms proof).
// d_tmp: eval_polynomial_batch_tmp_bytes(count, n).
int eval_polynomial_batch_launch(const Fr* const* d_polys, size_t count, size_t n, const uint64_t* points, Fr* d_tmp,
                                 uint64_t* out, hipStream_t stream) {
    if (count == 0) return H2_OK;
    if (n == 0) {
        memset(out, 0, 32 * count);
        return H2_OK;
    }
    const size_t per = eval_polynomial_tmp_elems(n);
    Fr* d_out = d_tmp + count * per;                        // packed results
    EvalPolyArgs* d_table = (EvalPolyArgs*)(d_out + count);  // one table per level
    std::vector<std::vector<EvalPolyArgs>> levels;
    std::vector<Fr> x(count);
    std::vector<const Fr*> in(count);
    std::vector<Fr*> dst(count);
    for (size_t j = 0; j < count; j++) {
        x[j] = fr_host(points + 4 * j);
        in[j] = d_polys[j];
        dst[j] = d_tmp + j * per;
    }
    size_t cnt = n;
    for (;;) {
        const size_t blocks = (cnt + 4095) / 4096;
        std::vector<EvalPolyArgs> table(count);
        for (size_t j = 0; j < count; j++) {
            EvalPolyArgs& a = table[j];
            a.in = in[j];
            a.out = blocks == 1 ? d_out + j : dst[j];
            a.n = cnt;
            Fr p = x[j];
            for (int l = 0; l < 8; l++) {
                a.xpow[l] = p;
                p = fp_sqr(p);
            }
            a.x256 = p;
            for (int l = 0; l < 4; l++) p = fp_sqr(p);  // x^256 -> x^4096: the next level's variable
            x[j] = p;
            in[j] = dst[j];
            dst[j] += blocks;
        }
        EvalPolyArgs* d_level = d_table + levels.size() * count;
        levels.push_back(std::move(table));
        H2_HIP(hipMemcpyAsync(d_level, levels.back().data(), count * sizeof(EvalPolyArgs), hipMemcpyHostToDevice, stream));
        for (size_t j0 = 0; j0 < count; j0 += 32768)
            hipLaunchKernelGGL(k_eval_poly_multi, dim3((unsigned)blocks, (unsigned)std::min<size_t>(32768, count - j0)),
                               dim3(256), 0, stream, d_level + j0);
        if (blocks == 1) break;
        cnt = blocks;
    }
    H2_HIP(hipGetLastError());
    H2_HIP(hipMemcpyAsync(out, d_out, 32 * count, hipMemcpyDeviceToHost, stream));
    H2_HIP(hipStreamSynchronize(stream));  // also keeps the level tables alive until their uploads have been consumed
    return H2_OK;
}

size_t eval_polynomial_batch_tmp_bytes(size_t count, size_t n) {
    size_t levels = 1, c = n ? n : 1;
    while ((c = (c + 4095) / 4096) > 1) levels++;
    return count * (eval_polynomial_tmp_elems(n ? n : 1) + 1) * sizeof(Fr) + (levels + 1) * count * sizeof(EvalPolyArgs) + 64;
}

size_t eval_polynomial_tmp_elems(size_t n) {
    size_t total = 2, c = n;
    while (c > 1) {
        c = (c + 4095) / 4096;
        total += c;
        if (c == 1) break;
    }
    return total;
}

// ---------------------------------------------------------------- batch_invert on device
// arithmetic.rs:840-844 (`parallelize` + ff::BatchInvert per chunk).  Montgomery's trick per lane over a
// strided set of 8..64 elements (coalesced across lanes); zeros stay zero.  3 multiplications per element
// plus one field inversion per workgroup (fp_inv: the binary extended GCD of field.hpp -- the inverting wave's lanes
// all hold the same value, so its branches are uniform: ~105 us instead of the ~230 us of the a^(r-2) chain).
// One inversion per WORKGROUP: the lanes' chain products are multiplied up by two LDS scans (prefix and suffix, 8 steps
// each), wave 0 inverts the workgroup's total, and lane t recovers the inverse of its own product as
// total^-1 * (product of the lanes before it) * (product of the lanes after it) -- 18 multiplications per lane instead
// of a private inversion (which made two thirds of this kernel's work).
__device__ __forceinline__ void binv_put(uint4* lo, uint4* hi, uint32_t i, const Fr& v) {
    lo[i] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    hi[i] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}
__device__ __forceinline__ Fr binv_get(const uint4* lo, const uint4* hi, uint32_t i) {
    const uint4 x = lo[i], y = hi[i];
    Fr r;
    r.l[0] = x.x; r.l[1] = x.y; r.l[2] = x.z; r.l[3] = x.w;
    r.l[4] = y.x; r.l[5] = y.y; r.l[6] = y.z; r.l[7] = y.w;
    return r;
}

__global__ void __launch_bounds__(256) k_batch_invert(Fr* a, Fr* prefix, size_t n, size_t nthreads) {
    __shared__ uint4 sh_lo[256], sh_hi[256];
    const uint32_t tid = threadIdx.x;
    const size_t t = (size_t)blockIdx.x * blockDim.x + tid;
    const bool active = t < nthreads;
    Fr acc = fp_one<FrParams>();
    size_t last = t;
    if (active) {
        for (size_t i = t; i < n; i += nthreads) {
            Fr v = fp_load(a + i);
            fp_store(prefix + i, acc);
            if (!fp_is_zero(v)) acc = fp_mul(acc, v);
            last = i;
        }
    }
    // before = product of the chain products of lanes < tid, after = of lanes > tid (Hillis-Steele, inclusive then shifted)
    Fr incl = acc;
    for (uint32_t off = 1; off < 256; off <<= 1) {
        binv_put(sh_lo, sh_hi, tid, incl);
        __syncthreads();
        if (tid >= off) incl = fp_mul(incl, binv_get(sh_lo, sh_hi, tid - off));
        __syncthreads();
    }
    binv_put(sh_lo, sh_hi, tid, incl);
    __syncthreads();
    const Fr before = tid ? binv_get(sh_lo, sh_hi, tid - 1) : fp_one<FrParams>();
    const Fr total = binv_get(sh_lo, sh_hi, 255);
    __syncthreads();
    Fr sfx = acc;
    for (uint32_t off = 1; off < 256; off <<= 1) {
        binv_put(sh_lo, sh_hi, tid, sfx);
        __syncthreads();
        if (tid + off < 256) sfx = fp_mul(sfx, binv_get(sh_lo, sh_hi, tid + off));
        __syncthreads();
    }
    binv_put(sh_lo, sh_hi, tid, sfx);
    __syncthreads();
    const Fr after = tid < 255 ? binv_get(sh_lo, sh_hi, tid + 1) : fp_one<FrParams>();
    __syncthreads();
    if (tid < 64) {  // one wave inverts (its lanes all hold `total`), the others wait at the barrier
        const Fr tinv = fp_inv(total);
        if (tid == 0) binv_put(sh_lo, sh_hi, 0, tinv);
    }
    __syncthreads();
    if (!active) return;
    Fr inv = fp_mul(fp_mul(binv_get(sh_lo, sh_hi, 0), before), after);
    for (size_t i = last;; i -= nthreads) {
        Fr v = fp_load(a + i);
        if (!fp_is_zero(v)) {
            fp_store(a + i, fp_mul(inv, fp_load(prefix + i)));
            inv = fp_mul(inv, v);
        }
        if (i < nthreads) break;
    }
}

int batch_invert_launch(Fr* d_a, Fr* d_tmp, size_t n, hipStream_t stream) {
    if (n == 0) return H2_OK;
    // elements per lane: 64 when that still fills the chip, down to 8 for small inputs (the chain of 3 multiplications
    // per element is pure latency there; the shared inversion costs a lane 18 multiplications whatever the chunk)
    size_t chunk = 64;
    while (chunk > 8 && n / chunk < 65536) chunk /= 2;
    size_t nthreads = (n + chunk - 1) / chunk;
    if (nthreads < 256) nthreads = n < 256 ? n : 256;
    hipLaunchKernelGGL(k_batch_invert, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, stream, d_a, d_tmp, n,
                       nthreads);
    H2_HIP(hipGetLastError());
    return H2_OK;
}

// ---------------------------------------------------------------- linear combination of polynomials
// res[i] = sum_j c_j * p_j[i]: the GWC / SHPLONK batching `poly_batch = poly_batch * v + p` (gwc/prover.rs:39-151,
// shplonk/prover.rs:110-209) with the powers of v supplied by the caller; one pass over every input instead of one
// eval_mul_c + eval_sum launch per polynomial.
static constexpr int LINCOMB_MAX = 8;
struct LincombArgs {
    const Fr* p[LINCOMB_MAX];
    Fr c[LINCOMB_MAX];
    Fr* res;
    size_t size;
    int count, accumulate;
};

__global__ void __launch_bounds__(256) k_lincomb(LincombArgs a) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.size; i += stride) {
        Fr acc = a.accumulate ? fp_load(a.res + i) : fp_zero<FrParams>();
#pragma unroll
        for (int j = 0; j < LINCOMB_MAX; j++)
            if (j < a.count) acc = fp_add(acc, fp_mul(fp_load(a.p[j] + i), a.c[j]));
        fp_store(a.res + i, acc);
    }
}

int lincomb_launch(Fr* res, const Fr* const* polys, const uint64_t* coeffs, size_t count, size_t size, hipStream_t stream) {
    if (size == 0) return H2_OK;
    if (count == 0) {
        H2_HIP(hipMemsetAsync(res, 0, size * sizeof(Fr), stream));
        return H2_OK;
    }
    for (size_t j0 = 0; j0 < count; j0 += LINCOMB_MAX) {
        LincombArgs a{};
        a.count = (int)std::min<size_t>(LINCOMB_MAX, count - j0);
        for (int j = 0; j < a.count; j++) {
            a.p[j] = polys[j0 + j];
            a.c[j] = fr_host(coeffs + 4 * (j0 + j));
            if (a.p[j] == res && (j0 + j) != 0) {
                set_last_error("h2_dev_lincomb: the result may only alias the first input");
                return H2_ERR_INVALID;
            }
        }
        a.res = res;
        a.size = size;
        a.accumulate = j0 != 0;
        hipLaunchKernelGGL(k_lincomb, dim3(grid_for(size)), dim3(256), 0, stream, a);
    }
    H2_HIP(hipGetLastError());
    return H2_OK;
}

// ---------------------------------------------------------------------------------------------
// a[i] *= g^i (the generalisation of distribute_powers_zeta, poly/domain.rs:382-398, to an arbitrary generator): with
// g = zeta * extended_omega^j it turns a coefficient vector into the input of the n-point NTT that evaluates it on coset
// j of the extended domain -- the unit of the coset-sharded multi-GPU proof.  Every lane raises g
// to its first index once and then steps by g^256 over 8 strided (coalesced) elements: ~5.5 products per element.
__global__ void __launch_bounds__(256) k_distribute_powers(Fr* a, size_t n, Fr g, Fr g_step) {
    const size_t base = (size_t)blockIdx.x * (256 * 8) + threadIdx.x;
    if (base >= n) return;
    Fr w = fp_pow_u32(g, (uint32_t)base);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const size_t i = base + (size_t)k * 256;
        if (i < n) {
            fp_store(a + i, fp_mul(fp_load(a + i), w));
            w = fp_mul(w, g_step);
        }
    }
}

int distribute_powers_launch(Fr* a, size_t n, const uint64_t g[4], hipStream_t stream) {
    if (n == 0) return H2_OK;
    if (n > ((size_t)1 << 28)) {
        set_last_error("h2_dev_distribute_powers: n exceeds 2^28");
        return H2_ERR_INVALID;
    }
    const Fr gv = fr_host(g);
    hipLaunchKernelGGL(k_distribute_powers, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, stream, a, n, gv,
                       fp_pow_u32(gv, 256));
    H2_HIP(hipGetLastError());
    return H2_OK;
}

// ---------------------------------------------------------------------------------------------
// Permutation argument, the elementwise parts around the grand-product scan:
//   keygen   permutation/keygen.rs:197-238   sigma_col[j] = DELTA^{c} * omega^{r} for mapping[col][j] = (c, r)
//   prover   permutation/prover.rs:89-128    per column of a set:
//              den[i] *= beta * sigma[i] + gamma + value[i]
//              num[i] *= DELTA^{col} * omega^{i} * beta + gamma + value[i]
// omega^i: every lane raises omega to its first index once (~35 products) and then steps by omega^256 over PT_ITEMS
// strided (coalesced) elements: 32 of them, so that the power costs ~1 product per element next to the 2-4 of the terms.
static constexpr int PT_ITEMS = 32;

__global__ void __launch_bounds__(256) k_perm_sigma(Fr* out, const uint32_t* map_col, const uint32_t* map_row, size_t n,
                                                    Fr delta, Fr omega) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr v = fp_mul(fp_pow_u32(delta, map_col[i]), fp_pow_u32(omega, map_row[i]));
    fp_store(out + i, v);
}

struct PermTermsArgs {
    Fr *num, *den;
    const Fr *value, *sigma;
    size_t n;
    Fr beta, gamma;
    Fr delta_beta;  // DELTA^{col} * beta
    Fr omega, omega_step;  // omega, omega^256
    int first;      // 1: overwrite num / den, 0: multiply into them
};

__global__ void __launch_bounds__(256) k_perm_terms(PermTermsArgs a) {
    const size_t base = (size_t)blockIdx.x * (256 * PT_ITEMS) + threadIdx.x;
    if (base >= a.n) return;
    Fr w = fp_mul(fp_pow_u32(a.omega, (uint32_t)base), a.delta_beta);
#pragma unroll 2
    for (int k = 0; k < PT_ITEMS; k++) {
        size_t i = base + (size_t)k * 256;
        if (i >= a.n) break;
        Fr v = fp_add(fp_load(a.value + i), a.gamma);
        Fr nu = fp_add(w, v);
        Fr de = fp_add(fp_mul(a.beta, fp_load(a.sigma + i)), v);
        if (!a.first) {
            nu = fp_mul(nu, fp_load(a.num + i));
            de = fp_mul(de, fp_load(a.den + i));
        }
        fp_store(a.num + i, nu);
        fp_store(a.den + i, de);
        w = fp_mul(w, a.omega_step);
    }
}

int perm_sigma_launch(Fr* out, const uint32_t* map_col, const uint32_t* map_row, size_t n, const uint64_t delta[4],
                      const uint64_t omega[4], hipStream_t stream) {
    if (n == 0) return H2_OK;
    hipLaunchKernelGGL(k_perm_sigma, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, out, map_col, map_row, n,
                       fr_host(delta), fr_host(omega));
    H2_HIP(hipGetLastError());
    return H2_OK;
}

int perm_terms_launch(Fr* num, Fr* den, const Fr* value, const Fr* sigma, size_t n, const uint64_t beta[4],
                      const uint64_t gamma[4], const uint64_t delta_pow[4], const uint64_t omega[4], int first,
                      hipStream_t stream) {
    if (n == 0) return H2_OK;
    PermTermsArgs a{};
    a.num = num;
    a.den = den;
    a.value = value;
    a.sigma = sigma;
    a.n = n;
    a.beta = fr_host(beta);
    a.gamma = fr_host(gamma);
    a.delta_beta = fp_mul(fr_host(delta_pow), a.beta);
    a.omega = fr_host(omega);
    a.omega_step = fp_pow_u32(a.omega, 256);
    a.first = first;
    const size_t per = 256 * PT_ITEMS;
    hipLaunchKernelGGL(k_perm_terms, dim3((unsigned)((n + per - 1) / per)), dim3(256), 0, stream, a);
    H2_HIP(hipGetLastError());
    return H2_OK;
}


// ---------------------------------------------------------------------------------------------
// The vanishing argument's blinding polynomial (vanishing/prover.rs:47-61: a parallel fill from thread_rng, every
// coefficient `Scalar::random` = 512 random bits reduced modulo r).  Here: element i = ChaCha20 block i under a 256-bit
// key (counter = i, nonce = 0; the key comes from OS entropy in halo2-gpu-specific_amd/rng.py, or from the seeded test
// stream), its 64 output bytes cut into two 253-bit integers lo (words 0..7) and hi (words 8..15), value = lo + 2^253 hi
// mod r (506 random bits: bias < 2^-252), stored in Montgomery form.  rng.py holds the host twin.
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
#define H2_QR(a, b, c, d)                                                  \
    a += b; d ^= a; d = rotl32(d, 16); c += d; b ^= c; b = rotl32(b, 12);  \
    a += b; d ^= a; d = rotl32(d, 8);  c += d; b ^= c; b = rotl32(b, 7)

struct ChaChaKey {
    uint32_t w[8];
};

__global__ void __launch_bounds__(256) k_random_fr(ChaChaKey key, size_t n, Fr* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t s[16], x[16];
    s[0] = 0x61707865u; s[1] = 0x3320646eu; s[2] = 0x79622d32u; s[3] = 0x6b206574u;  // "expand 32-byte k"
#pragma unroll
    for (int j = 0; j < 8; j++) s[4 + j] = key.w[j];
    s[12] = (uint32_t)i; s[13] = (uint32_t)((uint64_t)i >> 32); s[14] = 0; s[15] = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = s[j];
    for (int round = 0; round < 10; round++) {
        H2_QR(x[0], x[4], x[8], x[12]);
        H2_QR(x[1], x[5], x[9], x[13]);
        H2_QR(x[2], x[6], x[10], x[14]);
        H2_QR(x[3], x[7], x[11], x[15]);
        H2_QR(x[0], x[5], x[10], x[15]);
        H2_QR(x[1], x[6], x[11], x[12]);
        H2_QR(x[2], x[7], x[8], x[13]);
        H2_QR(x[3], x[4], x[9], x[14]);
    }
    Fr lo, hi;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        lo.l[j] = x[j] + s[j];
        hi.l[j] = x[8 + j] + s[8 + j];
    }
    lo.l[7] &= 0x1fffffffu;
    hi.l[7] &= 0x1fffffffu;
    Fr rr, k253;  // R^2 mod r and 2^253 R^2 mod r: canonical integer -> Montgomery form of lo + 2^253 hi
    constexpr uint32_t K253[8] = {0x3697e008u, 0x0bd29b1cu, 0xc39f76d7u, 0xa5491397u,
                                  0x9433f9fdu, 0x912798ccu, 0x6ff98cafu, 0x019f0b29u};
#pragma unroll
    for (int j = 0; j < 8; j++) {
        rr.l[j] = FrParams::RR[j];
        k253.l[j] = K253[j];
    }
    fp_store(out + i, fp_add(fp_mul(lo, rr), fp_mul(hi, k253)));
}
#undef H2_QR

int random_fr_launch(const uint8_t key[32], size_t n, uint64_t* d_out, hipStream_t stream) {
    if (n == 0) return H2_OK;
    ChaChaKey k;
    for (int j = 0; j < 8; j++)
        k.w[j] = (uint32_t)key[4 * j] | ((uint32_t)key[4 * j + 1] << 8) | ((uint32_t)key[4 * j + 2] << 16) | ((uint32_t)key[4 * j + 3] << 24);
    hipLaunchKernelGGL(k_random_fr, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, k, n, (Fr*)d_out);
    H2_HIP(hipGetLastError());
    return H2_OK;
}


}  // namespace h2
