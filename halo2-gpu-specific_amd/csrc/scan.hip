// scan.hip -- field-element prefix scans: the serial O(n) recurrences the reference runs on one CPU
// core between its GPU calls.
//   kate_division   arithmetic.rs:754-773   q[i] = a[i+1] + b*q[i+1]          (an affine recurrence, from the top)
//   grand products  permutation/prover.rs:89-165, shuffle/prover.rs, logup: z[i+1] = z[i] * f[i]
// Three-phase blocked scan: (1) a workgroup scans 1024 consecutive elements (4 per lane, staged through
// LDS so the global accesses stay coalesced; the 256 lane aggregates are combined by a Hillis-Steele scan
// whose multipliers b^(4*2^d) are compile-time-known powers passed by the host), (2) the per-workgroup
// aggregates are scanned by the same kernel recursively, (3) every workgroup but the first applies its
// predecessor's aggregate.  ~4.5 field multiplications per element.
#include <cstring>
#include <vector>

#include "common.hpp"
#include "scan.hpp"

namespace h2 {

static constexpr int SC_ITEMS = 4;
static constexpr int SC_BLOCK = 256 * SC_ITEMS;  // elements per workgroup

struct ScanArgs {
    const Fr* in;
    Fr* out;
    Fr* aggr;       // per-workgroup aggregates (inclusive value at the workgroup's last element)
    size_t n;       // elements to scan
    // index maps: logical element j reads in[in_rev ? in_top - j : j], writes out[out_rev ? out_top - j : j + out_shift]
    size_t in_top, out_top;
    int in_rev, out_rev, out_shift;
    int affine;     // 0: r_j = r_{j-1} * a_j ; 1: r_j = a_j + b * r_{j-1} ; 2: r_j = r_{j-1} + a_j
    int has_init;   // modes 0 / 2: a_0 is multiplied by / added to `init` on load
    Fr init;
    Fr bp[SC_ITEMS + 1];  // affine: b^1 .. b^4 in bp[1..4]
    Fr bstep[8];          // affine: b^(4 * 2^d), d = 0..7
};

__device__ __forceinline__ Fr lds_ld(const uint4* lo, const uint4* hi, uint32_t i) {
    uint4 a = lo[i], b = hi[i];
    Fr r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}
__device__ __forceinline__ void lds_st(uint4* lo, uint4* hi, uint32_t i, const Fr& v) {
    lo[i] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    hi[i] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

__global__ void __launch_bounds__(256) k_scan_local(ScanArgs a) {
    __shared__ uint4 s_lo[SC_BLOCK], s_hi[SC_BLOCK];
    const uint32_t t = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * SC_BLOCK;
    const Fr neutral = a.affine ? fp_zero<FrParams>() : fp_one<FrParams>();
    // coalesced load into LDS (logical order)
#pragma unroll
    for (int k = 0; k < SC_ITEMS; k++) {
        size_t j = base + (size_t)k * 256 + t;
        Fr v = neutral;
        if (j < a.n) {
            v = fp_load(a.in + (a.in_rev ? a.in_top - j : j));
            if (a.has_init && j == 0) v = a.affine == 2 ? fp_add(v, a.init) : fp_mul(v, a.init);
        }
        lds_st(s_lo, s_hi, (uint32_t)(k * 256 + t), v);
    }
    __syncthreads();
    // lane-local inclusive scan over 4 consecutive elements
    Fr loc[SC_ITEMS];
    Fr run = neutral;
#pragma unroll
    for (int k = 0; k < SC_ITEMS; k++) {
        Fr v = lds_ld(s_lo, s_hi, t * SC_ITEMS + k);
        run = a.affine == 1 ? fp_add(v, fp_mul(a.bp[1], run)) : a.affine == 2 ? fp_add(run, v) : fp_mul(run, v);
        loc[k] = run;
    }
    __syncthreads();
    // Hillis-Steele scan of the 256 lane aggregates (reusing the first 256 LDS slots)
    Fr agg = run;
    for (int d = 0; d < 8; d++) {
        lds_st(s_lo, s_hi, t, agg);
        __syncthreads();
        const uint32_t off = 1u << d;
        if (t >= off) {
            Fr o = lds_ld(s_lo, s_hi, t - off);
            agg = a.affine == 1 ? fp_add(agg, fp_mul(a.bstep[d], o)) : a.affine == 2 ? fp_add(agg, o) : fp_mul(agg, o);
        }
        __syncthreads();
    }
    lds_st(s_lo, s_hi, t, agg);
    __syncthreads();
    const bool has_carry = t > 0;
    Fr carry = has_carry ? lds_ld(s_lo, s_hi, t - 1) : neutral;
    __syncthreads();
    // final values of this lane, back through LDS for a coalesced store
#pragma unroll
    for (int k = 0; k < SC_ITEMS; k++) {
        Fr v = loc[k];
        if (has_carry)
            v = a.affine == 1 ? fp_add(v, fp_mul(a.bp[k + 1], carry)) : a.affine == 2 ? fp_add(v, carry) : fp_mul(v, carry);
        lds_st(s_lo, s_hi, t * SC_ITEMS + k, v);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SC_ITEMS; k++) {
        size_t j = base + (size_t)k * 256 + t;
        if (j < a.n) {
            Fr v = lds_ld(s_lo, s_hi, (uint32_t)(k * 256 + t));
            fp_store(a.out + (a.out_rev ? a.out_top - j : j + a.out_shift), v);
        }
    }
    if (t == 255 && a.aggr != nullptr) fp_store(a.aggr + blockIdx.x, agg);
}

// phase 3: elements of workgroup blk >= 1 absorb the inclusive aggregate of workgroups 0..blk-1
struct FixArgs {
    Fr* out;
    const Fr* aggr;     // scanned aggregates
    const Fr* bpow;     // affine: bpow[p] = b^(p+1), p < SC_BLOCK
    size_t n, out_top;
    int out_rev, out_shift, affine;
};

__global__ void __launch_bounds__(256) k_scan_fix(FixArgs a) {
    const size_t blk = (size_t)blockIdx.x + 1;
    const Fr carry = fp_load(a.aggr + (blk - 1));
#pragma unroll
    for (int k = 0; k < SC_ITEMS; k++) {
        uint32_t p = k * 256 + threadIdx.x;
        size_t j = blk * SC_BLOCK + p;
        if (j >= a.n) continue;
        Fr* dst = a.out + (a.out_rev ? a.out_top - j : j + a.out_shift);
        Fr v = fp_load(dst);
        v = a.affine == 1 ? fp_add(v, fp_mul(fp_load(a.bpow + p), carry)) : a.affine == 2 ? fp_add(v, carry) : fp_mul(v, carry);
        fp_store(dst, v);
    }
}

// bpow[p] = b^(p+1)
__global__ void __launch_bounds__(256) k_bpow(Fr* out, Fr b, uint32_t count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) fp_store(out + i, fp_pow_u32(b, i + 1));
}

static Fr fr_from(const uint64_t v[4]) {
    Fr r;
    for (int i = 0; i < 4; i++) {
        r.l[2 * i] = (uint32_t)v[i];
        r.l[2 * i + 1] = (uint32_t)(v[i] >> 32);
    }
    return r;
}

size_t scan_tmp_elems(size_t n) {
    size_t total = 16, c = n;
    while (c > SC_BLOCK) {
        c = (c + SC_BLOCK - 1) / SC_BLOCK;
        total += c + SC_BLOCK;  // aggregates + the b^(p+1) table of that level
    }
    return total;
}

// Recursive driver.  `b` is only used by the affine scan.  tmp: scan_tmp_elems(n) elements.
static void scan_run(ScanArgs a, const Fr& b, Fr* tmp, hipStream_t stream) {
    const size_t n = a.n;
    if (n == 0) return;
    const size_t blocks = (n + SC_BLOCK - 1) / SC_BLOCK;
    Fr* bpow = nullptr;
    if (a.affine == 1) {
        Fr p = b;
        for (int k = 1; k <= SC_ITEMS; k++) {
            a.bp[k] = p;
            p = fp_mul(p, b);
        }
        Fr s = a.bp[SC_ITEMS];  // b^4
        for (int d = 0; d < 8; d++) {
            a.bstep[d] = s;
            s = fp_sqr(s);
        }
        // s = b^(4 * 256) = b^1024: the multiplier of the next level
        if (blocks > 1) {
            bpow = tmp;
            tmp += SC_BLOCK;
            hipLaunchKernelGGL(k_bpow, dim3(SC_BLOCK / 256), dim3(256), 0, stream, bpow, b, (uint32_t)SC_BLOCK);
        }
        a.aggr = blocks > 1 ? tmp : nullptr;
        hipLaunchKernelGGL(k_scan_local, dim3((unsigned)blocks), dim3(256), 0, stream, a);
        if (blocks > 1) {
            ScanArgs up{};
            up.in = tmp;
            up.out = tmp;
            up.n = blocks;
            up.affine = 1;
            scan_run(up, s, tmp + blocks, stream);
        }
    } else {
        a.aggr = blocks > 1 ? tmp : nullptr;
        hipLaunchKernelGGL(k_scan_local, dim3((unsigned)blocks), dim3(256), 0, stream, a);
        if (blocks > 1) {
            ScanArgs up{};
            up.in = tmp;
            up.out = tmp;
            up.n = blocks;
            up.affine = a.affine;
            scan_run(up, b, tmp + blocks, stream);
        }
    }
    if (blocks > 1) {
        FixArgs f{};
        f.out = a.out;
        f.aggr = tmp;
        f.bpow = bpow;
        f.n = n;
        f.out_top = a.out_top;
        f.out_rev = a.out_rev;
        f.out_shift = a.out_shift;
        f.affine = a.affine;
        hipLaunchKernelGGL(k_scan_fix, dim3((unsigned)(blocks - 1)), dim3(256), 0, stream, f);
    }
}

// kate_division (arithmetic.rs:754-773): q(X) = a(X) / (X - b), a has n coefficients, q has n - 1.
// q[n-2] = a[n-1]; q[i] = a[i+1] + b*q[i+1].  With j = n-2-i: r_j = a[n-1-j] + b*r_{j-1}.
int kate_division_launch(const Fr* d_a, size_t n, const uint64_t b[4], Fr* d_q, Fr* d_tmp, hipStream_t stream) {
    if (n < 2) return H2_OK;
    ScanArgs s{};
    s.in = d_a;
    s.out = d_q;
    s.n = n - 1;
    s.in_rev = 1;
    s.in_top = n - 1;
    s.out_rev = 1;
    s.out_top = n - 2;
    s.affine = 1;
    scan_run(s, fr_from(b), d_tmp, stream);
    H2_HIP(hipGetLastError());
    return H2_OK;
}

// z[0] = init; z[i] = z[i-1] * f[i-1], i < n   (the grand-product columns: permutation/prover.rs:151-160)
int prefix_product_launch(const Fr* d_f, size_t n, const uint64_t init[4], Fr* d_z, Fr* d_tmp, hipStream_t stream) {
    if (n == 0) return H2_OK;
    Fr i0 = fr_from(init);
    H2_HIP(hipMemcpyAsync(d_z, &i0, sizeof(Fr), hipMemcpyHostToDevice, stream));
    if (n > 1) {
        ScanArgs s{};
        s.in = d_f;
        s.out = d_z;
        s.n = n - 1;
        s.out_shift = 1;
        s.affine = 0;
        s.has_init = 1;
        s.init = i0;
        scan_run(s, i0, d_tmp, stream);
    }
    H2_HIP(hipGetLastError());
    H2_HIP(hipStreamSynchronize(stream));  // `i0` is a stack temporary of this frame
    return H2_OK;
}

// z[0] = init; z[i] = z[i-1] + f[i-1], i < n   (the logup grand sums: logup/prover.rs:353-367)
int prefix_sum_launch(const Fr* d_f, size_t n, const uint64_t init[4], Fr* d_z, Fr* d_tmp, hipStream_t stream) {
    if (n == 0) return H2_OK;
    Fr i0 = fr_from(init);
    H2_HIP(hipMemcpyAsync(d_z, &i0, sizeof(Fr), hipMemcpyHostToDevice, stream));
    if (n > 1) {
        ScanArgs s{};
        s.in = d_f;
        s.out = d_z;
        s.n = n - 1;
        s.out_shift = 1;
        s.affine = 2;
        s.has_init = 1;
        s.init = i0;
        scan_run(s, i0, d_tmp, stream);
    }
    H2_HIP(hipGetLastError());
    H2_HIP(hipStreamSynchronize(stream));  // `i0` is a stack temporary of this frame
    return H2_OK;
}

}  // namespace h2
