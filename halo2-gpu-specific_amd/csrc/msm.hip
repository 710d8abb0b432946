// msm.hip -- Pippenger multi-scalar multiplication over BN254 G1 for gfx950.
//
// Replaces ec-gpu-gen's `SingleMultiexpKernel::multiexp_bound` (called from
// /root/reference/halo2_proofs/src/arithmetic.rs:334-367) ; the CPU twin that fixes the
// semantics is `multiexp_serial` (arithmetic.rs:20-108): result = sum_i scalar_i * base_i.
// The output is a group element, so window size, digit form and summation order are free
// (results are compared after affine normalisation, like the reference's own `==`).
//
// Pipeline (one stream, no host round trip until the final 1-2 KB read-back):
//   k_digits      scalar -> canonical (one Montgomery mul), signed c-bit digits, bucket histogram
//   k_scan        exclusive scans: entries per bucket, 32-entry segments per bucket
//   k_scatter     counting sort of (point index, sign) by (window, bucket)
//   k_acc_seg     one thread per (bucket, segment): <= 32 mixed XYZZ additions        [hot loop]
//   k_finish      per bucket: fold its segment partials (serial when few, else queued)
//   k_finish_heavy one workgroup per heavy bucket: strided fold + LDS tree
//   k_reduce      per window: sum_b (b+1) * B_b by chunked running sums + small scalar mul + LDS tree
//   host          adds the <= W*G partial window sums and runs the W*c doublings (Horner)
// Splitting buckets into fixed 32-entry segments keeps the hot loop load-balanced for the
// skewed digit distributions real witness columns have (boolean / small-valued columns put
// most points into a handful of buckets -- SURVEY.md section 7 "hard parts (ii)").
#include <algorithm>
#include <array>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "common.hpp"
#include "ec.hpp"
#include "msm.hpp"

namespace h2 {

static constexpr uint32_t SEG = 32;          // entries per accumulation segment
static constexpr uint32_t FINISH_SERIAL = 8; // partials a single thread folds in k_finish
static constexpr uint32_t KEY_INVALID = 0xffffffffu;
static constexpr uint32_t SIGN_BIT = 0x80000000u;
static constexpr uint32_t REDUCE_T = 256;    // threads per k_reduce workgroup
static constexpr uint32_t REDUCE_M = 8;      // buckets per k_reduce thread

struct MsmShape {
    uint32_t c, W, nb, nbt, G;  // window bits, windows, buckets/window, total buckets, reduce groups/window
    size_t n, entries, max_items;
    // scratch offsets (bytes)
    size_t off_keys, off_sorted, off_counts, off_starts, off_cursor, off_segstarts, off_heavy, off_partials,
        off_buckets, off_winpart, total;
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static MsmShape msm_shape(size_t n, uint32_t max_bits) {
    MsmShape s{};
    if (max_bits > 254) max_bits = 254;
    if (max_bits == 0) max_bits = 1;
    s.n = n;
    // cost model: W * (n + 3 * 2^(c-1)) group additions
    double best = 1e300;
    uint32_t best_c = 4;
    for (uint32_t c = 2; c <= 20; c++) {
        uint32_t W = (max_bits + 1 + c - 1) / c;
        double cost = (double)W * ((double)n + 3.0 * (double)(1u << (c - 1)));
        if (cost < best) {
            best = cost;
            best_c = c;
        }
    }
    if (const char* env = getenv("H2_MSM_WINDOW")) {
        int v = atoi(env);
        if (v >= 2 && v <= 20) best_c = (uint32_t)v;
    }
    s.c = best_c;
    s.W = (max_bits + 1 + s.c - 1) / s.c;
    s.nb = 1u << (s.c - 1);
    s.nbt = s.W * s.nb;
    uint32_t per_group = REDUCE_T * REDUCE_M;
    s.G = (s.nb + per_group - 1) / per_group;
    s.entries = n * s.W;
    s.max_items = s.entries / SEG + s.nbt + 1;
    size_t o = 0;
    auto take = [&](size_t bytes) {
        size_t r = o;
        o = align_up(o + bytes, 256);
        return r;
    };
    s.off_keys = take(s.entries * 4);
    s.off_sorted = take(s.entries * 4);
    s.off_counts = take(((size_t)s.nbt + 2) * 4);
    s.off_starts = take(((size_t)s.nbt + 2) * 4);
    s.off_cursor = take(((size_t)s.nbt + 2) * 4);
    s.off_segstarts = take(((size_t)s.nbt + 2) * 4);
    s.off_heavy = take(((size_t)s.nbt + 2) * 4);
    s.off_partials = take(s.max_items * sizeof(XYZZ));
    s.off_buckets = take((size_t)s.nbt * sizeof(XYZZ));
    s.off_winpart = take((size_t)s.W * s.G * sizeof(XYZZ));
    s.total = o;
    return s;
}

size_t msm_scratch_bytes(size_t n, uint32_t max_bits) { return msm_shape(n, max_bits).total; }
void msm_shape_query(size_t n, uint32_t max_bits, uint32_t* c, uint32_t* windows, uint32_t* buckets_per_window) {
    MsmShape s = msm_shape(n, max_bits);
    if (c) *c = s.c;
    if (windows) *windows = s.W;
    if (buckets_per_window) *buckets_per_window = s.nb;
}

// ---------------------------------------------------------------- k_digits
__global__ void __launch_bounds__(256) k_digits(const Fr* scalars, size_t n, uint32_t c, uint32_t W, uint32_t nb,
                                                uint32_t max_bits, uint32_t* keys, uint32_t* counts) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr s = fp_from_mont(fp_load(scalars + i));  // canonical little-endian integer (to_repr, arithmetic.rs:21)
    // keep only the low max_bits bits (multiexp_bound contract)
#pragma unroll
    for (int k = 0; k < 8; k++) {
        int lo_bit = 32 * k;
        if ((int)max_bits <= lo_bit)
            s.l[k] = 0;
        else if ((int)max_bits < lo_bit + 32)
            s.l[k] &= (1u << (max_bits - lo_bit)) - 1;
    }
    const uint32_t mask = (1u << c) - 1, half = 1u << (c - 1);
    uint64_t buf = 0;
    int nbits = 0;
    uint32_t w = 0, carry = 0;
    auto emit = [&](uint32_t raw) {
        raw += carry;
        uint32_t neg = 0, mag = raw;
        if (raw > half) {  // digit = raw - 2^c  (negative)
            mag = (1u << c) - raw;
            neg = SIGN_BIT;
            carry = 1;
        } else {
            carry = 0;
        }
        uint32_t out = KEY_INVALID;
        if (mag != 0) {
            uint32_t key = w * nb + (mag - 1);
            atomicAdd(&counts[key], 1u);
            out = key | neg;
        }
        keys[(size_t)w * n + i] = out;
        w++;
    };
#pragma unroll
    for (int k = 0; k < 8; k++) {
        buf |= (uint64_t)s.l[k] << nbits;
        nbits += 32;
        while (nbits >= (int)c && w < W) {
            emit((uint32_t)buf & mask);
            buf >>= c;
            nbits -= c;
        }
    }
    while (w < W) {
        emit((uint32_t)buf & mask);
        buf >>= c;
    }
}

// ---------------------------------------------------------------- k_scan (single workgroup)
// starts[b] = sum_{b'<b} counts[b'];  segstarts[b] = sum_{b'<b} ceil(counts[b']/SEG);  cursor = starts.
// Arrays have nbt + 1 entries (the last holds the totals).
__global__ void __launch_bounds__(1024) k_scan(const uint32_t* counts, uint32_t nbt, uint32_t* starts,
                                               uint32_t* cursor, uint32_t* segstarts) {
    __shared__ uint32_t sh_a[1024], sh_b[1024];
    __shared__ uint32_t base_a, base_b;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) {
        base_a = 0;
        base_b = 0;
    }
    __syncthreads();
    const uint32_t ITEMS = 4, CHUNK = 1024 * ITEMS;
    for (uint32_t c0 = 0; c0 < nbt + 1; c0 += CHUNK) {
        uint32_t va[ITEMS], vb[ITEMS], sa = 0, sb = 0;
#pragma unroll
        for (uint32_t k = 0; k < ITEMS; k++) {
            uint32_t idx = c0 + tid * ITEMS + k;
            uint32_t cnt = (idx < nbt) ? counts[idx] : 0;
            va[k] = sa;
            vb[k] = sb;
            sa += cnt;
            sb += (cnt + SEG - 1) / SEG;
        }
        sh_a[tid] = sa;
        sh_b[tid] = sb;
        __syncthreads();
        // Hillis-Steele inclusive scan over the 1024 thread totals
        for (uint32_t off = 1; off < 1024; off <<= 1) {
            uint32_t xa = 0, xb = 0;
            if (tid >= off) {
                xa = sh_a[tid - off];
                xb = sh_b[tid - off];
            }
            __syncthreads();
            sh_a[tid] += xa;
            sh_b[tid] += xb;
            __syncthreads();
        }
        uint32_t ea = base_a + sh_a[tid] - sa, eb = base_b + sh_b[tid] - sb;  // exclusive prefix of this thread
#pragma unroll
        for (uint32_t k = 0; k < ITEMS; k++) {
            uint32_t idx = c0 + tid * ITEMS + k;
            if (idx <= nbt) {
                starts[idx] = ea + va[k];
                cursor[idx] = ea + va[k];
                segstarts[idx] = eb + vb[k];
            }
        }
        __syncthreads();
        if (tid == 1023) {
            base_a += sh_a[1023];
            base_b += sh_b[1023];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- k_scatter
__global__ void __launch_bounds__(256) k_scatter(const uint32_t* keys, size_t n, uint32_t* cursor, uint32_t* sorted) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t w = blockIdx.y;
    uint32_t key = keys[(size_t)w * n + i];
    if (key == KEY_INVALID) return;
    uint32_t pos = atomicAdd(&cursor[key & ~SIGN_BIT], 1u);
    sorted[pos] = (uint32_t)i | (key & SIGN_BIT);
}

// ---------------------------------------------------------------- k_acc_seg (hot loop)
__global__ void __launch_bounds__(256) k_acc_seg(const Affine* bases, const uint32_t* sorted, const uint32_t* starts,
                                                 const uint32_t* segstarts, uint32_t nbt, XYZZ* partials) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t total_items = segstarts[nbt];
    if (t >= total_items) return;
    // bucket = largest b with segstarts[b] <= t  (binary search; the table is L2-resident)
    uint32_t lo = 0, hi = nbt;  // invariant: segstarts[lo] <= t < segstarts[hi]
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (segstarts[mid] <= t)
            lo = mid;
        else
            hi = mid;
    }
    uint32_t b = lo, j = t - segstarts[b];
    uint32_t e0 = starts[b] + j * SEG, e1 = starts[b + 1];
    if (e1 > e0 + SEG) e1 = e0 + SEG;
    XYZZ acc = xyzz_identity();
    for (uint32_t e = e0; e < e1; e++) {
        uint32_t ref = sorted[e];
        Affine p = affine_load(bases + (ref & ~SIGN_BIT));
        acc = xyzz_madd(acc, p, (ref & SIGN_BIT) != 0);
    }
    xyzz_store(partials + t, acc);
}

// ---------------------------------------------------------------- k_finish / k_finish_heavy
__global__ void __launch_bounds__(256) k_finish(const XYZZ* partials, const uint32_t* segstarts, uint32_t nbt,
                                                XYZZ* buckets, uint32_t* heavy_list, uint32_t* heavy_count) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nbt) return;
    uint32_t p0 = segstarts[b], p1 = segstarts[b + 1];
    uint32_t np = p1 - p0;
    if (np > FINISH_SERIAL) {
        heavy_list[atomicAdd(heavy_count, 1u)] = b;
        return;
    }
    XYZZ acc = xyzz_identity();
    if (np >= 1) acc = xyzz_load(partials + p0);
    for (uint32_t p = p0 + 1; p < p1; p++) acc = xyzz_add(acc, xyzz_load(partials + p));
    xyzz_store(buckets + b, acc);
}

__global__ void __launch_bounds__(256) k_finish_heavy(const XYZZ* partials, const uint32_t* segstarts,
                                                      const uint32_t* heavy_list, const uint32_t* heavy_count,
                                                      XYZZ* buckets) {
    __shared__ XYZZ sh[256];
    const uint32_t tid = threadIdx.x;
    uint32_t nheavy = *heavy_count;
    for (uint32_t h = blockIdx.x; h < nheavy; h += gridDim.x) {
        uint32_t b = heavy_list[h];
        uint32_t p0 = segstarts[b], p1 = segstarts[b + 1];
        XYZZ acc = xyzz_identity();
        for (uint32_t p = p0 + tid; p < p1; p += 256) acc = xyzz_add(acc, xyzz_load(partials + p));
        sh[tid] = acc;
        __syncthreads();
        for (uint32_t off = 128; off >= 1; off >>= 1) {
            if (tid < off) sh[tid] = xyzz_add(sh[tid], sh[tid + off]);
            __syncthreads();
        }
        if (tid == 0) xyzz_store(buckets + b, sh[0]);
        __syncthreads();
    }
}

// ---------------------------------------------------------------- k_reduce
// window w, group g: sum over this group's buckets of (b + 1) * B_b   (b = index inside the window)
__global__ void __launch_bounds__(REDUCE_T) k_reduce(const XYZZ* buckets, uint32_t nb, uint32_t G, XYZZ* winpart) {
    __shared__ XYZZ sh[REDUCE_T];
    const uint32_t tid = threadIdx.x, g = blockIdx.x, w = blockIdx.y;
    const XYZZ* B = buckets + (size_t)w * nb;
    uint32_t k0 = (g * REDUCE_T + tid) * REDUCE_M;
    XYZZ res = xyzz_identity();
    if (k0 < nb) {
        uint32_t k1 = k0 + REDUCE_M;
        if (k1 > nb) k1 = nb;
        XYZZ running = xyzz_identity(), acc = xyzz_identity();
        for (uint32_t b = k1; b-- > k0;) {  // summation by parts (arithmetic.rs:98-106)
            running = xyzz_add(running, xyzz_load(B + b));
            acc = xyzz_add(acc, running);
        }
        // acc = sum (b - k0 + 1) * B_b ;  running = sum B_b
        res = acc;
        if (k0 != 0 && !xyzz_is_identity(running)) res = xyzz_add(res, xyzz_mul_u32(running, k0));
    }
    sh[tid] = res;
    __syncthreads();
    for (uint32_t off = REDUCE_T / 2; off >= 1; off >>= 1) {
        if (tid < off) sh[tid] = xyzz_add(sh[tid], sh[tid + off]);
        __syncthreads();
    }
    if (tid == 0) xyzz_store(winpart + (size_t)w * G + g, sh[0]);
}

// ---------------------------------------------------------------- synthetic bases (bench / tests)
// n deterministic G1 points by try-and-increment: x = mix(seed, i), y = (x^3 + 3)^((q+1)/4) when that
// is a square root (q = 3 mod 4).  Cofactor 1: every curve point is in G1.  Not part of the prover
// path; it exists so bench.py can build its workload without touching the CPU oracle.
__device__ __forceinline__ uint64_t splitmix64(uint64_t& x) {
    uint64_t z = (x += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

__global__ void __launch_bounds__(256) k_random_points(uint64_t seed, size_t n, Affine* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // (q + 1) / 4, little-endian u32 limbs
    const uint32_t E[8] = {0xb61f3f52u, 0x4f082305u, 0x5a1c72a3u, 0x65e05aa4u,
                           0xa0605617u, 0x6e14116du, 0xb84c680au, 0x0c19139cu};
    uint64_t st = seed ^ (0xd1342543de82ef95ull * (uint64_t)(i + 1));
    Fq x;
    for (int k = 0; k < 4; k++) {
        uint64_t v = splitmix64(st);
        x.l[2 * k] = (uint32_t)v;
        x.l[2 * k + 1] = (uint32_t)(v >> 32);
    }
    x.l[7] &= 0x1fffffffu;  // < 2^253 < q: a valid Montgomery residue
    Fq three = fp_add(fp_add(fp_one<FqParams>(), fp_one<FqParams>()), fp_one<FqParams>());
    for (;;) {
        Fq rhs = fp_add(fp_mul(fp_sqr(x), x), three);
        Fq y = fp_one<FqParams>();
        for (int bit = 253; bit >= 0; bit--) {
            y = fp_sqr(y);
            if ((E[bit >> 5] >> (bit & 31)) & 1) y = fp_mul(y, rhs);
        }
        if (fp_eq(fp_sqr(y), rhs)) {
            if (splitmix64(st) & 1) y = fp_neg(y);
            fp_store(&out[i].x, x);
            fp_store(&out[i].y, y);
            return;
        }
        x = fp_add(x, fp_one<FqParams>());
    }
}

int random_points_launch(uint64_t seed, size_t n, uint64_t* d_out, hipStream_t stream) {
    if (n == 0) return H2_OK;
    hipLaunchKernelGGL(k_random_points, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, seed, n,
                       (Affine*)d_out);
    H2_HIP(hipGetLastError());
    return H2_OK;
}

// ---------------------------------------------------------------- drivers
void msm_identity(uint64_t out_xyz[12]) {
    Jacobian j = xyzz_to_jacobian(xyzz_identity());
    memcpy(out_xyz, &j, 96);
}

static void msm_launch(const MsmShape& s, const Fr* d_scalars, const Affine* d_bases, uint32_t max_bits, char* scratch,
                       hipStream_t stream) {
    uint32_t* keys = (uint32_t*)(scratch + s.off_keys);
    uint32_t* sorted = (uint32_t*)(scratch + s.off_sorted);
    uint32_t* counts = (uint32_t*)(scratch + s.off_counts);
    uint32_t* starts = (uint32_t*)(scratch + s.off_starts);
    uint32_t* cursor = (uint32_t*)(scratch + s.off_cursor);
    uint32_t* segstarts = (uint32_t*)(scratch + s.off_segstarts);
    uint32_t* heavy = (uint32_t*)(scratch + s.off_heavy);  // [0] = count, [1..] = list
    XYZZ* partials = (XYZZ*)(scratch + s.off_partials);
    XYZZ* buckets = (XYZZ*)(scratch + s.off_buckets);
    XYZZ* winpart = (XYZZ*)(scratch + s.off_winpart);

    H2_HIP(hipMemsetAsync(counts, 0, ((size_t)s.nbt + 2) * 4, stream));
    H2_HIP(hipMemsetAsync(heavy, 0, 4, stream));
    unsigned nblk = (unsigned)((s.n + 255) / 256);
    hipLaunchKernelGGL(k_digits, dim3(nblk), dim3(256), 0, stream, d_scalars, s.n, s.c, s.W, s.nb,
                       max_bits > 254 ? 254u : max_bits, keys, counts);
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, stream, counts, s.nbt, starts, cursor, segstarts);
    hipLaunchKernelGGL(k_scatter, dim3(nblk, s.W), dim3(256), 0, stream, keys, s.n, cursor, sorted);
    unsigned iblk = (unsigned)((s.max_items + 255) / 256);
    hipLaunchKernelGGL(k_acc_seg, dim3(iblk), dim3(256), 0, stream, d_bases, sorted, starts, segstarts, s.nbt, partials);
    hipLaunchKernelGGL(k_finish, dim3((s.nbt + 255) / 256), dim3(256), 0, stream, partials, segstarts, s.nbt, buckets,
                       heavy + 1, heavy);
    hipLaunchKernelGGL(k_finish_heavy, dim3(1024), dim3(256), 0, stream, partials, segstarts, heavy + 1, heavy, buckets);
    hipLaunchKernelGGL(k_reduce, dim3(s.G, s.W), dim3(REDUCE_T), 0, stream, buckets, s.nb, s.G, winpart);
    H2_HIP(hipGetLastError());
}

// host tail: add the G partials of each window, then Horner over the windows
static void msm_host_tail(const MsmShape& s, const std::vector<XYZZ>& winpart, uint64_t out_xyz[12]) {
    XYZZ acc = xyzz_identity();
    for (int w = (int)s.W - 1; w >= 0; w--) {
        for (uint32_t k = 0; k < s.c; k++) acc = xyzz_double(acc);
        XYZZ ws = xyzz_identity();
        for (uint32_t g = 0; g < s.G; g++) ws = xyzz_add(ws, winpart[(size_t)w * s.G + g]);
        acc = xyzz_add(acc, ws);
    }
    Jacobian j = xyzz_to_jacobian(acc);
    memcpy(out_xyz, &j, 96);
}

int msm_device(DeviceCtx* ctx, const Fr* d_scalars, const uint64_t* d_bases, size_t n, uint32_t max_bits,
               void* d_scratch, size_t scratch_bytes, uint64_t* out_xyz, hipStream_t stream) {
    if (n == 0 || max_bits == 0) {
        msm_identity(out_xyz);
        return H2_OK;
    }
    if (n > 0x7fffffffu) {
        set_last_error("h2 msm: n must be < 2^31");
        return H2_ERR_INVALID;
    }
    MsmShape s = msm_shape(n, max_bits);
    if (!d_scratch || scratch_bytes < s.total) {
        set_last_error("h2 msm: scratch too small (see h2_msm_scratch_bytes)");
        return H2_ERR_INVALID;
    }
    msm_launch(s, d_scalars, (const Affine*)d_bases, max_bits, (char*)d_scratch, stream);
    std::vector<XYZZ> winpart((size_t)s.W * s.G);
    H2_HIP(hipMemcpyAsync(winpart.data(), (char*)d_scratch + s.off_winpart, winpart.size() * sizeof(XYZZ),
                          hipMemcpyDeviceToHost, stream));
    H2_HIP(hipStreamSynchronize(stream));
    msm_host_tail(s, winpart, out_xyz);
    return H2_OK;
}

int msm_host_resident_scalars(DeviceCtx* ctx, const Fr* d_scalars, const uint64_t* bases, size_t n, uint32_t max_bits,
                              uint64_t out_xyz[12]) {
    Affine* d_bases = (Affine*)ctx->buf_c.get(n * sizeof(Affine));
    H2_HIP(hipMemcpyAsync(d_bases, bases, n * sizeof(Affine), hipMemcpyHostToDevice, ctx->stream));
    size_t sb = msm_scratch_bytes(n, max_bits);
    void* scratch = ctx->msm_scratch.get(sb);
    return msm_device(ctx, d_scalars, (const uint64_t*)d_bases, n, max_bits, scratch, sb, out_xyz, ctx->stream);
}

int msm_host(DeviceCtx* ctx, const uint64_t* scalars, const uint64_t* bases, size_t n, uint32_t max_bits,
             uint64_t out_xyz[12]) {
    Fr* d_s = (Fr*)ctx->buf_d.get(n * sizeof(Fr));
    H2_HIP(hipMemcpyAsync(d_s, scalars, n * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    return msm_host_resident_scalars(ctx, d_s, bases, n, max_bits, out_xyz);
}

// gpu_multiexp_bound (arithmetic.rs:413-440): ceil(n / N_GPU) contiguous chunks, one leased
// device each (par_chunks), partial points folded on the host (`reduce(|acc, x| acc + x)`).
int msm_host_multi(const uint64_t* scalars, const uint64_t* bases, size_t n, uint32_t max_bits, uint64_t out_xyz[12]) {
    int n_gpu = device_count();
    if (n_gpu <= 0) throw HipError{hipErrorNoDevice, "no HIP device visible", __FILE__, __LINE__};
    size_t part_len = (n + n_gpu - 1) / n_gpu;
    size_t nparts = (n + part_len - 1) / part_len;
    std::vector<std::array<uint64_t, 12>> parts(nparts);
    std::vector<int> rcs(nparts, H2_OK);
    std::vector<std::string> errs(nparts);
    std::vector<std::thread> th;
    for (size_t p = 0; p < nparts; p++) {
        th.emplace_back([&, p] {
            size_t lo = p * part_len, len = std::min(part_len, n - lo);
            rcs[p] = guarded([&] {
                DeviceLease lease;
                return msm_host(lease.ctx, scalars + 4 * lo, bases + 8 * lo, len, max_bits, parts[p].data());
            });
            if (rcs[p] != H2_OK) errs[p] = get_last_error();
        });
    }
    for (auto& t : th) t.join();
    for (size_t p = 0; p < nparts; p++)
        if (rcs[p] != H2_OK) {
            set_last_error(errs[p]);
            return rcs[p];
        }
    g1_sum_host(parts[0].data(), nparts, out_xyz);
    return H2_OK;
}

// host fold of `count` Jacobian points (12 x u64 each) -- the `reduce(|acc, x| acc + x)` of
// arithmetic.rs:434 and the local add after an all-gather of per-rank partial points.
// Jacobian (X, Y, Z) -> XYZZ (X, Y, Z^2, Z^3).
void g1_sum_host(const uint64_t* points, size_t count, uint64_t out_xyz[12]) {
    XYZZ acc = xyzz_identity();
    for (size_t p = 0; p < count; p++) {
        Jacobian j;
        memcpy(&j, points + 12 * p, 96);
        XYZZ q;
        q.x = j.x;
        q.y = j.y;
        q.zz = fp_sqr(j.z);
        q.zzz = fp_mul(q.zz, j.z);
        acc = xyzz_add(acc, q);
    }
    Jacobian j = xyzz_to_jacobian(acc);
    memcpy(out_xyz, &j, 96);
}

}  // namespace h2
