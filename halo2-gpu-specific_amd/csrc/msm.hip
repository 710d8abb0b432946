// msm.hip -- Pippenger multi-scalar multiplication over BN254 G1 for gfx950.
//
// Replaces ec-gpu-gen's `SingleMultiexpKernel::multiexp_bound` (called from
// /root/reference/halo2_proofs/src/arithmetic.rs:334-367) ; the CPU twin that fixes the
// semantics is `multiexp_serial` (arithmetic.rs:20-108): result = sum_i scalar_i * base_i.
// The output is a group element, so window size, digit form and summation order are free
// (results are compared after affine normalisation, like the reference's own `==`).
//
// Pipeline (one stream, no host round trip until the final 1-2 KB read-back):
//   k_sample      64 scalars -> host: does one value dominate the column? (it then gets its own window, see k_digits)
//   k_digits      scalar -> canonical (one Montgomery mul), signed digits of BALANCED width (c or c - 1 bits), keys, and
//                 the histogram of (window, bucket >> lo_bits) partitions; narrow columns: rows cut into ranges = windows
//   k_scan_parts / k_partition / k_bucket_sort   two-level counting sort of (point index, sign) by (window, bucket); over a
//                 shifted-base table k_partition is k_part_a + k_part_b: two passes that reorder their tile in LDS first
//   k_acc_slice   one thread per fixed-size SLICE of the sorted list: exactly S mixed XYZZ
//                 additions per lane whatever the bucket sizes are; a partial sum is emitted at
//                 every bucket boundary inside the slice                              [hot loop]
//   k_finish      a QUAD per bucket folds its slice partials (<= 32), heavier buckets are queued
//   k_finish_mid  queued buckets with <= 512 partials: one wave each (strided fold + shuffle tree)
//   k_finish_heavy / _heavy2   the rest: ceil(P / 256) workgroups per bucket (strided fold + quad tree), then their fold
//   k_reduce      per window: sum_b (b+1) * B_b by chunked running sums on quads + small scalar mul + quad tree; the last
//                 workgroup of a window folds its groups
//   host          W window sums -> Horner (W additions, max_bits + 1 doublings); the dominant scalar's multiplication
// Everything after k_acc_slice runs on quads (ec_quad.hpp): four lanes per chain of dependent additions.
// Slicing the *sorted list* evenly (instead of giving each bucket to a thread) keeps every lane of
// the hot loop busy for any digit distribution: Poisson-sized buckets of a uniform MSM as well as
// the skewed columns real witnesses have (boolean / small-valued columns put most points into a
// handful of buckets -- SURVEY.md section 7 "hard parts (ii)").  The partial of (bucket b, slice s)
// lives in slot b + s: along the sorted list either b or s grows at every emission, so slots are
// unique, a bucket's partials are contiguous, and no second prefix sum is needed.
#include <algorithm>
#include <array>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>

#include "common.hpp"
#include "ec_quad.hpp"
#include "msm.hpp"

namespace h2 {

static constexpr uint32_t FINISH_SERIAL = 32; // partials a single quad folds in k_finish
static constexpr uint32_t SORT_T = 1024;      // threads of a k_bucket_sort workgroup (one partition each)
static constexpr uint32_t HEAVY_SPLIT = 64;   // workgroups sharing one heavy bucket in k_finish_heavy
static constexpr uint32_t MID_MAX = 512;      // partials of a bucket one wave folds in k_finish_mid
static constexpr uint32_t KEY_INVALID = 0xffffffffu;
static constexpr uint32_t SIGN_BIT = 0x80000000u;
static constexpr uint32_t REDUCE_T = 512;    // threads per k_reduce workgroup = 128 quads (two waves per SIMD)
static constexpr uint32_t REDUCE_QM = 32;    // consecutive buckets a quad walks (fewer when that still fits the chip)
static constexpr uint32_t PART_T = 2048;      // entries per k_partition workgroup (256 threads x 8)
static constexpr uint32_t PART_AB_T = 8192;       // entries per workgroup of the two-level partition passes (256 threads x 32)
static constexpr uint32_t PARTA_GROUPS_MAX = 128; // level-A groups (2^a_bits <= 2^7)
static constexpr uint32_t PART_T_TABLE = 16384;  // ... over a shifted-base table with many partitions (256 threads x 64)

struct MsmShape {
    uint32_t c, W, nb, nbt, G;  // window bits, digit windows, buckets/window, total buckets, points exported per window (1)
    uint32_t RG, qm;            // k_reduce: workgroups per window (folded on the device by the last one to finish) and
                                // consecutive buckets per quad
    uint32_t Wt;                // windows the pipeline runs: R * W, + 1 when a dominant scalar has its own window (see k_digits)
    uint32_t Wk;                // key arrays k_digits writes (n keys each): W, + 1 with a dominant scalar; cols * Wc when fused
    uint32_t wfull;             // windows 0 .. wfull - 1 are c bits wide, the rest c - 1 (see msm_shape)
    uint32_t R, range_shift;    // row ranges: rows i >> range_shift = r use windows r * W .. r * W + W - 1 (see msm_shape)
    uint32_t cols, Wc;          // fused multi-column shape: `cols` columns x Wc = W + 1 windows each (cols = 0: one MSM)
    uint32_t tab;               // 1: shifted-base table (see ShiftTable): the W digits of a scalar share ONE window of nb buckets
    size_t tab_stride;          // points per level of the table: digit w of row i adds table[w * tab_stride + i]
    size_t off_coltab;          // fused: per-column scalar pointers (8 B) and dominant values (32 B)
    uint32_t log_s;             // slice length S = 2^log_s entries
    uint32_t lo_bits, hi_bits;  // bucket id = hi (partition inside the window) : lo (bin inside the partition)
    uint32_t np;                // partitions = W << hi_bits
    size_t n, entries, max_items;
    // scratch offsets (bytes)
    size_t off_keys, off_sorted, off_tmp, off_pcount, off_pbase, off_pcursor, off_starts, off_heavy, off_partials,
        off_buckets, off_winpart, off_rcount, off_bflags, off_tmpa, total;
    uint32_t two_level;         // table shape: partition in two coalesced passes (k_part_a / k_part_b), a_bits + b_bits = hi_bits
    uint32_t a_bits;
    // table shape, reduce by bit planes (k_reduce_chunks / k_reduce_planes): `planes` points leave the device per MSM --
    // P_0 .. P_(planes-2) and the sum of the chunks' own weighted sums -- in winpart[Wt ..]; 0 = the classic k_reduce
    uint32_t planes, chunks, plane_l, plane_seg;
    size_t off_planes;          // the chunk sums R[chunks], then the plane partials [planes][plane_seg], then `planes` counters
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static MsmShape msm_shape(size_t n, uint32_t max_bits, bool hot, uint32_t cols = 0) {
    MsmShape s{};
    if (max_bits > 254) max_bits = 254;
    if (max_bits == 0) max_bits = 1;
    s.n = n;
    // cost model: W * (n + 3 * 2^(c-1)) group additions
    double best = 1e300;
    uint32_t best_c = 4;
    // windows wider than 17 bits lose more in the bucket-proportional kernels than they save in additions (measured at
    // 2^22 / 2^24: c = 17 8.4 / 28.8 ms, c = 18 13.7 / 46.3 ms, c = 20 - / 40.7 ms: k_bucket_sort's partitions get small
    // and numerous, k_reduce walks 2^c * W buckets), so the model only ranks c <= 17
    for (uint32_t c = 2; c <= 17; c++) {
        uint32_t W = (max_bits + 1 + c - 1) / c;
        double cost = (double)W * ((double)n + 3.0 * (double)(1u << (c - 1)));
        if (cost < best) {
            best = cost;
            best_c = c;
        }
    }
    if (const char* env = getenv("H2_MSM_WINDOW")) {
        int v = atoi(env);
        if (v >= 2 && v <= 18) best_c = (uint32_t)v;  // k_reduce folds <= 128 groups per window
    }
    s.c = best_c;
    s.W = (max_bits + 1 + s.c - 1) / s.c;
    // Balanced windows.  max_bits + 1 bits cut into W digits of c bits leave the top window whatever remains -- 7 bits
    // at c = 13, 2 bits at c = 12, 5 at c = 10: a window whose n digits share a handful of buckets, i.e. one sort
    // partition (one workgroup) and nothing but heavy buckets.  The same W windows get (max_bits + 1) / W bits each
    // instead, the first `wfull` of them one more: widths c and c - 1, every window with >= 2^(c-2) buckets.
    {
        const uint32_t T = max_bits + 1, base = T / s.W, rem = T % s.W;
        s.c = rem ? base + 1 : base;
        s.wfull = rem ? rem : s.W;
    }
    s.Wt = s.W + (hot ? 1u : 0u);
    s.R = 1;
    s.range_shift = 31;
    s.cols = cols;
    // every fused column keeps a slot for its dominant-scalar window -- except in shapes of one or two windows, where that
    // window is never used (see msm_device) and would only double the finish / reduce work
    s.Wc = s.W + ((cols && s.W <= 2) ? 0u : 1u);
    if (cols) s.Wt = cols * s.Wc;
    s.Wk = s.Wt;
    s.nb = 1u << (s.c - 1);
    // Row ranges.  A narrow column (booleans, bytes, a 10-bit opcode) has ONE window of a few buckets: one or a handful
    // of sort partitions (= workgroups) for all n entries, thousands of entries per bucket, everything after the sort on
    // the heavy-bucket path.  Its rows are cut into R ranges that act as separate windows -- range r of window w is
    // window r * W + w, with its own buckets and partitions -- and the host adds the R sums of each window.
    if (!cols && n >= (1u << 15)) {
        const uint32_t lo0 = (s.c - 1 < 8) ? (s.c - 1) : 8;
        const uint32_t base_np = s.W << (s.c - 1 - lo0);
        uint32_t R = 1;
        // ... until there are >= 128 partitions and a bucket holds <= 2048 entries on average (k_finish_mid's range of
        // slice partials even when half the buckets go unused, as with booleans), while the buckets stay few against
        // the entries and a range keeps >= 2048 rows
        while ((R * base_np < 128 || n / ((size_t)R * s.W * s.nb) > 2048) && (size_t)2 * R * s.W * s.nb * 4 <= n &&
               (n / (2 * R)) >= 2048)
            R *= 2;
        if (R > 1) {
            uint32_t shift = 11;
            while (((size_t)1 << shift) * R < n) shift++;
            s.range_shift = shift;
            s.R = (uint32_t)((n + ((size_t)1 << shift) - 1) >> shift);
            s.Wt = s.R * s.W + (hot ? 1u : 0u);
        }
    }
    s.nbt = s.Wt * s.nb;
    // k_reduce is a chain of 2 * qm additions per quad: the shortest chains whose workgroups still fit the chip once
    s.qm = REDUCE_QM;
    for (uint32_t qm = 4; qm < REDUCE_QM; qm *= 2) {
        uint32_t rg = (s.nb + REDUCE_T / 4 * qm - 1) / (REDUCE_T / 4 * qm);
        static const size_t reduce_wgs = getenv("H2_MSM_REDUCE_WGS") ? (size_t)std::max(1, atoi(getenv("H2_MSM_REDUCE_WGS"))) : 256;
        if ((size_t)s.Wt * rg <= reduce_wgs && rg <= REDUCE_T / 4) {  // one workgroup per CU: a CU with two runs both at half speed
            s.qm = qm;
            break;
        }
    }
    if (const char* env = getenv("H2_MSM_REDUCE_QM")) {
        int v = atoi(env);
        if (v >= 1 && v <= 64 && (s.nb + REDUCE_T / 4 * v - 1) / (REDUCE_T / 4 * v) <= REDUCE_T / 4) s.qm = (uint32_t)v;
    }
    uint32_t per_group = REDUCE_T / 4 * s.qm;
    s.RG = (s.nb + per_group - 1) / per_group;
    s.G = 1;
    s.entries = n * s.Wk;
    // slice length: aim at >= 2^18 slices (one resident round of the chip at 4 waves/SIMD), 8 <= S <= 64
    s.log_s = 6;
    while (s.log_s > 3 && (s.entries >> s.log_s) < (1u << 18)) s.log_s--;
    // large MSMs: keep the partials per bucket low (k_finish folds <= FINISH_SERIAL of them per thread, more go through
    // the heavy-bucket path) by lengthening the slices while at least 2^20 of them remain
    while (s.log_s < 10 && (s.entries >> (s.log_s + 1)) >= (1u << 20) && ((n / s.nb) >> s.log_s) > 2) s.log_s++;
    if (const char* env = getenv("H2_MSM_SLICE_LOG")) {
        int v = atoi(env);
        if (v >= 1 && v <= 10) s.log_s = (uint32_t)v;
    }
    // two-level counting sort: the low bits of a bucket id are resolved inside LDS (k_bucket_sort), the
    // high bits select one of 2^hi_bits partitions per window (k_partition); W << hi_bits <= 16384 so
    // the per-workgroup partition histogram of k_digits fits in 64 KiB of LDS
    s.lo_bits = (s.c - 1 < 8) ? (s.c - 1) : 8;
    while (((size_t)s.Wt << (s.c - 1 - s.lo_bits)) > 16384) s.lo_bits++;
    s.hi_bits = s.c - 1 - s.lo_bits;
    s.np = s.Wt << s.hi_bits;
    s.max_items = (s.entries >> s.log_s) + s.nbt + 2;
    size_t o = 0;
    auto take = [&](size_t bytes) {
        size_t r = o;
        o = align_up(o + bytes, 256);
        return r;
    };
    s.off_keys = take(s.entries * 4);
    s.off_sorted = take(s.entries * 4);
    s.off_tmp = take(s.entries * 8);
    s.off_pcount = take(((size_t)s.np + 2) * 4);
    s.off_pbase = take(((size_t)s.np + 2) * 4);
    s.off_pcursor = take(((size_t)s.np + 2) * 4);
    s.off_starts = take(((size_t)s.nbt + 2) * 4);
    s.off_heavy = take(((size_t)s.nbt + 2) * 4);
    s.off_partials = take(s.max_items * sizeof(XYZZ));
    s.off_buckets = take((size_t)s.nbt * sizeof(XYZZ));
    s.off_winpart = take((size_t)s.Wt * (1 + s.RG) * sizeof(XYZZ));  // window sums, then the RG group partials of each
    s.off_rcount = take((size_t)s.Wt * 4);
    s.off_bflags = take(n / 256 + 4);  // one byte per 256-row block: the block is one entry of the dominant-value bucket
    s.off_coltab = take((size_t)(cols ? cols : 1) * 64);
    s.total = o;
    return s;
}

// ---------------------------------------------------------------- shifted-base tables
// A base table that is committed against again and again (the SRS: `params.g`, `params.g_lagrange`,
// poly/commitment.rs:148-170) can be expanded ONCE into T[j][i] = [2^(o_j)] P_i, o_j = the bit offset of digit j
// (h2_dev_bases_precompute; D x n x 64 B -- 12 GiB for 2^24 points, which is what 288 GB of HBM are for).  Digit j of
// scalar i then adds T[j][i] into bucket |d| - 1 of a bucket set SHARED by all digits:
//     sum_i s_i P_i = sum_i sum_j d_ij [2^(o_j)] P_i = sum_b (b + 1) * (sum of the +-T[j][i] with |d_ij| = b + 1),
// so the number of buckets no longer grows with the number of windows and the windows can be much wider than 17 bits:
// 12 digits of 22 / 21 bits instead of 15 of 17 at 2^24 (a fifth of the additions gone), one k_reduce window instead
// of 15-20 (the latency-bound tail of small MSMs), no Horner on the host.  Semantics unchanged (arithmetic.rs:20-108).
struct ShiftTable {
    const Affine* table;  // level 0 = a copy of the bases
    size_t n;             // points per level
    uint32_t D, c, wfull; // digits; digit j is c bits wide for j < wfull, c - 1 after (balanced cut of 255 bits)
    const Affine* blocks; // sums of BLOCK_ROWS consecutive bases (see "dominant value over whole blocks"); null in a view
                          // that does not start on a block boundary
};
static constexpr uint32_t BLOCK_ROWS = 256;           // = the rows a k_digits workgroup takes per step
static constexpr uint32_t BLOCK_INDEX0 = 0x40000000u; // point index of block sum 0 in the sorted entries
static constexpr uint32_t BLOCK_KEY = 0x40000000u;    // key of a row that stands for its whole block (bucket 0)

// digits a scalar below 2^max_bits needs: the signed top digit must not carry out, i.e. o_(j+1) >= max_bits + 1
static uint32_t table_digits_used(const ShiftTable& t, uint32_t max_bits) {
    uint32_t off = 0;
    for (uint32_t j = 0; j < t.D; j++) {
        off += t.c - (j >= t.wfull ? 1u : 0u);
        if (off >= max_bits + 1) return j + 1;
    }
    return t.D;
}

static MsmShape msm_shape_table(size_t n, uint32_t max_bits, bool hot, const ShiftTable& t) {
    MsmShape s{};
    if (max_bits > 254) max_bits = 254;
    if (max_bits == 0) max_bits = 1;
    s.n = n;
    s.tab = 1;
    s.tab_stride = t.n;
    s.c = t.c;
    s.wfull = t.wfull;
    s.W = table_digits_used(t, max_bits);
    s.Wk = s.W + (hot ? 1u : 0u);
    s.Wt = 1 + (hot ? 1u : 0u);
    s.R = 1;
    s.range_shift = 31;
    s.Wc = s.W + 1;
    s.nb = 1u << (s.c - 1);
    // k_reduce: every quad pays a ~(c + c / 2)-step scalar multiplication to lift its chunk, so the chunks are as long
    // as still leaves a workgroup per CU (<= 64 buckets: a chain of 128 additions)
    s.qm = 4;
    while (s.qm < 64 && (s.nb + REDUCE_T / 4 * s.qm - 1) / (REDUCE_T / 4 * s.qm) > 256) s.qm *= 2;
    if (const char* env = getenv("H2_MSM_REDUCE_QM")) {
        int v = atoi(env);
        if (v >= 1 && v <= 64) s.qm = (uint32_t)v;
    }
    const uint32_t per_group = REDUCE_T / 4 * s.qm;
    s.RG = (s.nb + per_group - 1) / per_group;
    s.G = 1;
    s.entries = n * s.Wk;
    s.log_s = 6;
    while (s.log_s > 3 && (s.entries >> s.log_s) < (1u << 18)) s.log_s--;
    while (s.log_s < 10 && (s.entries >> (s.log_s + 1)) >= (1u << 20) && ((s.entries / s.nb) >> s.log_s) > 2) s.log_s++;
    if (const char* env = getenv("H2_MSM_SLICE_LOG")) {
        int v = atoi(env);
        if (v >= 1 && v <= 10) s.log_s = (uint32_t)v;
    }
    // sort: 2^8 bins per k_bucket_sort workgroup -- measured at 2^24 with 21-bit bucket ids: 2^8 bins 1.6 ms, 2^10 3.5 ms,
    // 2^12 5.1 ms (placing the bins in sweeps of 256 does not change that: profiles/r2_msm_experiments.txt), k_partition
    // over the remaining 2^13 / 2^11 / 2^9 partitions 3.0 / 2.5 / 2.0 ms -- so up to 2^13 partitions for k_partition
    s.lo_bits = (s.c - 1 < 8) ? (s.c - 1) : 8;
    while (s.c - 1 - s.lo_bits > 13 && s.lo_bits < 12) s.lo_bits++;
    if (const char* env = getenv("H2_MSM_TABLE_LO")) {
        int v = atoi(env);
        if (v >= 4 && v <= 12 && (uint32_t)v <= s.c - 1 && s.c - 1 - (uint32_t)v <= 13) s.lo_bits = (uint32_t)v;
    }
    s.hi_bits = s.c - 1 - s.lo_bits;
    // the dominant-scalar window only ever fills its bucket 0: it gets ONE partition (2^lo_bits buckets) behind window 0's
    s.np = (1u << s.hi_bits) + (hot ? 1u : 0u);
    s.nbt = s.nb + (hot ? (1u << s.lo_bits) : 0u);
    s.max_items = (s.entries >> s.log_s) + s.nbt + 2;
    size_t o = 0;
    auto take = [&](size_t bytes) {
        size_t r = o;
        o = align_up(o + bytes, 256);
        return r;
    };
    s.off_keys = take(s.entries * 4);
    s.off_sorted = take(s.entries * 4);
    s.off_tmp = take(s.entries * 8);
    s.off_pcount = take(((size_t)s.np + 2) * 4);
    s.off_pbase = take(((size_t)s.np + 2) * 4);
    s.off_pcursor = take(((size_t)s.np + 2) * 4);
    s.off_starts = take(((size_t)s.nbt + 2) * 4);
    s.off_heavy = take(((size_t)s.nbt + 2) * 4);
    s.off_partials = take(s.max_items * sizeof(XYZZ));
    s.off_buckets = take((size_t)s.nbt * sizeof(XYZZ));
    // Reduce by bit planes (below, k_reduce_chunks): for power-of-two chunk lengths and at least a workgroup of chunks
    static const bool planes_on = !(getenv("H2_MSM_REDUCE_PLANES") && atoi(getenv("H2_MSM_REDUCE_PLANES")) == 0);
    s.planes = 0;
    if (planes_on && (s.qm & (s.qm - 1)) == 0 && s.nb >= 4 * s.qm) {
        s.chunks = (s.nb + s.qm - 1) / s.qm;
        uint32_t nbits = 0;
        while ((1u << nbits) < s.chunks) nbits++;
        s.planes = nbits + 1;
        // the planes' workgroups together fill three quarters of the chip: plane_seg workgroups per plane, plane_l chunk sums
        // per quad.  Measured (profiles/r5_msm_plane_wgs_sweep.txt): MORE workgroups -- shorter chains per quad -- lose (512:
        // +3..12 % on a single MSM, the segment fold and the launch grow), 128-256 tie for a single MSM and the smaller grids
        // leave more of the chip to the next column's accumulation in a batch (2^16 0.25 -> 0.23 ms per MSM, 2^18 0.54 -> 0.53)
        static const uint32_t plane_wgs = getenv("H2_MSM_PLANE_WGS") ? (uint32_t)std::max(1, atoi(getenv("H2_MSM_PLANE_WGS"))) : 192u;
        s.plane_seg = std::max<uint32_t>(1, plane_wgs / s.planes);
        const uint32_t half = nbits ? (1u << (nbits - 1)) : 1u;
        while (s.plane_seg > 1 && (s.plane_seg - 1) * (REDUCE_T / 4) >= half) s.plane_seg--;   // no empty workgroups
        s.plane_l = (half + s.plane_seg * (REDUCE_T / 4) - 1) / (s.plane_seg * (REDUCE_T / 4));
    }
    s.off_winpart = take(((size_t)s.Wt * (1 + s.RG) + s.planes) * sizeof(XYZZ));
    s.off_rcount = take((size_t)s.Wt * 4);
    s.off_planes = s.planes ? take(((size_t)s.chunks + (size_t)s.planes * s.plane_seg) * sizeof(XYZZ) + (size_t)s.planes * 4 + 64) : 0;
    s.off_bflags = take(n / 256 + 4);  // one byte per 256-row block: the block is one entry of the dominant-value bucket
    s.off_coltab = take(64);
    // two-level partition (H2_MSM_TWO_LEVEL=0: the single pass): the entries pass through a second 8-byte-per-entry buffer
    static const bool two_level = !(getenv("H2_MSM_TWO_LEVEL") && atoi(getenv("H2_MSM_TWO_LEVEL")) == 0);
    static const uint32_t two_level_min_hi = getenv("H2_MSM_TWO_LEVEL_MIN_HI") ? (uint32_t)atoi(getenv("H2_MSM_TWO_LEVEL_MIN_HI")) : 8u;
    s.two_level = (two_level && s.hi_bits >= two_level_min_hi && s.hi_bits >= 2 && s.hi_bits <= 14) ? 1u : 0u;
    s.a_bits = s.hi_bits / 2;
    s.off_tmpa = s.two_level ? take(s.entries * 8 + (PARTA_GROUPS_MAX * 3 + 4) * 4) : 0;
    s.total = o;
    return s;
}

namespace {
std::mutex g_tab_mu;
std::map<const uint64_t*, ShiftTable> g_tables;  // device base pointer -> its table
}  // namespace

// table whose bases contain [d_bases, d_bases + n): a view that starts at d_bases's row
static bool table_find(const uint64_t* d_bases, size_t n, ShiftTable* out) {
    if (const char* env = getenv("H2_MSM_NO_TABLE"))
        if (env[0] == '1') return false;
    std::lock_guard<std::mutex> g(g_tab_mu);
    auto it = g_tables.upper_bound(d_bases);
    if (it == g_tables.begin()) return false;
    --it;
    if (d_bases + 8 * n > it->first + 8 * it->second.n) return false;
    ShiftTable t = it->second;
    const size_t off = (size_t)(d_bases - it->first) / 8;
    t.table += off;
    t.blocks = (t.blocks && off % BLOCK_ROWS == 0) ? t.blocks + off / BLOCK_ROWS : nullptr;
    *out = t;
    return true;
}

// ... and is it worth using for this bound?
static bool table_lookup(const uint64_t* d_bases, size_t n, uint32_t max_bits, ShiftTable* out) {
    // below 2^15 scalars the windowed pipeline (and its fused groups) is as fast: 2^14 0.41 vs 0.44 ms
    size_t min_n = (size_t)1 << 15;
    if (const char* env = getenv("H2_MSM_TABLE_MIN_N")) min_n = (size_t)atoll(env);
    if (n < min_n) return false;
    ShiftTable t;
    if (!table_find(d_bases, n, &t)) return false;
    // narrow columns: the plain pipeline's row ranges and fused groups are built for them, and a table saves nothing
    static const bool force = getenv("H2_MSM_TABLE_FORCE") != nullptr;  // experiment: tables of more digits than windows
    if (!force && table_digits_used(t, max_bits) >= msm_shape(n, max_bits, false).W) return false;
    *out = t;
    return true;
}

int bases_forget(const uint64_t* d_bases);
// drops the table whose bases contain d_bases (its bases no longer hold the points it was built from)
static void table_drop_containing(const uint64_t* d_bases) {
    const uint64_t* key = nullptr;
    {
        std::lock_guard<std::mutex> g(g_tab_mu);
        auto it = g_tables.upper_bound(d_bases);
        if (it == g_tables.begin()) return;
        --it;
        if (d_bases >= it->first + 8 * it->second.n) return;
        key = it->first;
    }
    bases_forget(key);
}

// sized for the larger of the two shapes (with the extra window of a dominant scalar), table forms included
size_t msm_scratch_bytes(size_t n, uint32_t max_bits) {
    size_t need = std::max(msm_shape(n, max_bits, true).total, msm_shape(n, max_bits, false).total);
    std::lock_guard<std::mutex> g(g_tab_mu);
    for (const auto& kv : g_tables)
        if (kv.second.n >= n)   // (both forms: with the dominant-scalar array the slices can come out longer and the partials fewer)
            need = std::max(need, std::max(msm_shape_table(n, max_bits, true, kv.second).total, msm_shape_table(n, max_bits, false, kv.second).total));
    return need;
}
// scratch that lets h2_dev_msm_batch(_ex) fuse `count` columns of bound `max_bits` over one base table
size_t msm_batch_scratch_bytes(size_t n, uint32_t max_bits, size_t count);

void msm_shape_query(size_t n, uint32_t max_bits, uint32_t* c, uint32_t* windows, uint32_t* buckets_per_window) {
    MsmShape s = msm_shape(n, max_bits, false);
    if (c) *c = s.c;
    if (windows) *windows = s.W;
    if (buckets_per_window) *buckets_per_window = s.nb;
}

// ---------------------------------------------------------------- k_digits
// One thread per scalar (grid-stride): Montgomery -> canonical, signed c-bit digits, keys[w][i] =
// bucket | sign (KEY_INVALID for a zero digit), and the histogram of (window, bucket >> lo_bits)
// partitions, privatised in LDS and flushed once per workgroup.
//
// Dominant scalar.  Committed columns are often constant over most rows -- a grand-product / grand-sum column over
// the padding rows of a circuit, a default value -- and Pippenger would pay W additions for each of those rows.  When
// sampling finds a value v on >= 1/4 of the rows (`hot_on`), the rows holding v contribute no digits at all; instead
// they enter bucket 0 of one extra window (index W), whose sum E = sum of their points is multiplied by v once on the
// host: sum_i s_i P_i = sum_{s_i != v} s_i P_i + v * E.  One addition per such row instead of W.
extern __shared__ __attribute__((aligned(16))) uint32_t h2_msm_smem[];

// Fused multi-column form (col_scalars != nullptr): blockIdx.y is the column; its scalars, its dominant value and its
// flag come from the tables, its W + 1 windows start at window blockIdx.y * (W + 1), and `np` counts the partitions of
// ONE column -- the rest of the pipeline just sees more windows over the same bases.
__global__ void __launch_bounds__(256) k_digits(const Fr* scalars, size_t n, uint32_t c, uint32_t W, uint32_t nb,
                                                uint32_t max_bits, uint32_t lo_bits, uint32_t hi_bits, uint32_t np,
                                                uint32_t* keys, uint32_t* pcount, int hot_on, Fr hot,
                                                const Fr* const* col_scalars, const Fr* col_hot, uint64_t col_hot_mask,
                                                uint32_t range_shift, uint32_t R, uint32_t wfull, uint32_t tabmode,
                                                size_t block_rows_end, uint8_t* block_flags) {
    // tabmode (shifted-base table): every digit's bucket belongs to window 0, the dominant-scalar window is window 1
    // block_rows_end (dominant value over whole blocks): rows below it lie in complete blocks of BLOCK_ROWS bases whose
    // SUMS are tabulated next to the bases -- the padding rows of a circuit make a grand-product column one value over
    // millions of consecutive rows: a block of 256 rows all holding the dominant value enters its bucket as ONE point
    // (row 0 of the block carries BLOCK_KEY, the other 255 nothing) instead of 256.
    bool hot_slot = hot_on != 0;  // does key array W exist?
    if (col_scalars != nullptr) {
        const uint32_t col = blockIdx.y, Wc = np >> hi_bits;  // windows (= key arrays) per column: W or W + 1
        scalars = col_scalars[col];
        hot_on = (int)((col_hot_mask >> col) & 1);
        if (hot_on) hot = fp_load(col_hot + col);
        hot_slot = Wc > W;
        keys += (size_t)col * Wc * n;
        pcount += (size_t)col * np;
    }
    uint32_t* hist = h2_msm_smem;
    for (uint32_t k = threadIdx.x; k < np; k += blockDim.x) hist[k] = 0;
    __syncthreads();
    for (size_t base = (size_t)blockIdx.x * blockDim.x; base < n; base += (size_t)gridDim.x * blockDim.x) {
        const size_t i = base + threadIdx.x;
        const bool valid = i < n;
        const Fr raw = valid ? fp_load(scalars + i) : fp_zero<FrParams>();
        if (hot_slot) {  // uniform branch; a fused column of more than two windows always owns the extra window
            const bool is_hot = valid && hot_on && fp_eq(raw, hot);
            bool whole_block = false;
            if (block_rows_end) whole_block = __syncthreads_and(is_hot ? 1 : 0) != 0 && base + BLOCK_ROWS <= block_rows_end;
            if (whole_block) {
                // the W digit keys of these rows are neither written nor read: k_partition skips flagged blocks
                if (threadIdx.x == 0) {
                    atomicAdd(&hist[(tabmode ? 1u : R * W) << hi_bits], 1u);
                    block_flags[base / BLOCK_ROWS] = 1;
                }
                keys[(size_t)W * n + i] = threadIdx.x == 0 ? BLOCK_KEY : KEY_INVALID;
                continue;
            } else {
                const uint64_t m = __ballot(is_hot);
                if (m && (int)(threadIdx.x & 63) == __ffsll((unsigned long long)m) - 1)
                    atomicAdd(&hist[(tabmode ? 1u : R * W) << hi_bits], (uint32_t)__popcll(m));  // partition 0 of the extra window
                if (valid) keys[(size_t)W * n + i] = is_hot ? 0u : KEY_INVALID;
            }
            if (is_hot) {
                for (uint32_t w = 0; w < W; w++) keys[(size_t)w * n + i] = KEY_INVALID;
                continue;
            }
        }
        if (!valid) continue;
        Fr s = fp_from_mont(raw);  // canonical little-endian integer (to_repr, arithmetic.rs:21)
        // keep only the low max_bits bits (multiexp_bound contract)
#pragma unroll
        for (int k = 0; k < 8; k++) {
            int lo_bit = 32 * k;
            if ((int)max_bits <= lo_bit)
                s.l[k] = 0;
            else if ((int)max_bits < lo_bit + 32)
                s.l[k] &= (1u << (max_bits - lo_bit)) - 1;
        }
        uint64_t buf = 0;
        int nbits = 0;
        uint32_t w = 0, carry = 0;
        const uint32_t vw0 = (uint32_t)(i >> range_shift) * W;  // first window of this row's range
        const uint32_t wstep = tabmode ? 0u : 1u;
        auto emit = [&](uint32_t raw, uint32_t cw) {  // cw: width of window w
            raw += carry;
            uint32_t neg = 0, mag = raw;
            if (raw > (1u << (cw - 1))) {  // digit = raw - 2^cw  (negative)
                mag = (1u << cw) - raw;
                neg = SIGN_BIT;
                carry = 1;
            } else {
                carry = 0;
            }
            uint32_t out = KEY_INVALID;
            if (mag != 0) {
                uint32_t bucket = mag - 1;
                out = bucket | neg;
                atomicAdd(&hist[((vw0 + w * wstep) << hi_bits) + (bucket >> lo_bits)], 1u);
            }
            keys[(size_t)w * n + i] = out;
            w++;
        };
#pragma unroll
        for (int k = 0; k < 8; k++) {
            buf |= (uint64_t)s.l[k] << nbits;
            nbits += 32;
            for (;;) {
                const uint32_t cw = c - (w >= wfull ? 1u : 0u);
                if (nbits < (int)cw || w >= W) break;
                emit((uint32_t)buf & ((1u << cw) - 1), cw);
                buf >>= cw;
                nbits -= cw;
            }
        }
        while (w < W) {
            const uint32_t cw = c - (w >= wfull ? 1u : 0u);
            emit((uint32_t)buf & ((1u << cw) - 1), cw);
            buf >>= cw;
        }
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < np; k += blockDim.x) {
        uint32_t v = hist[k];
        if (v) atomicAdd(&pcount[k], v);
    }
}

// ---------------------------------------------------------------- k_scan_parts (single workgroup)
__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t* sh, uint32_t* total) {
    // inclusive scan inside each wave64 with shuffles, then across the 4 waves through LDS
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t y = __shfl_up(x, off, 64);
        if (lane >= (uint32_t)off) x += y;
    }
    if (lane == 63) sh[wave] = x;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < wave; w++) base += sh[w];
    *total = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return base + x - v;
}

// the same over the SORT_T = 1024 threads (16 waves) of k_bucket_sort
__device__ __forceinline__ uint32_t block_exclusive_scan_1024(uint32_t v, uint32_t* sh, uint32_t* total) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t y = __shfl_up(x, off, 64);
        if (lane >= (uint32_t)off) x += y;
    }
    if (lane == 63) sh[wave] = x;
    __syncthreads();
    uint32_t base = 0, all = 0;
#pragma unroll
    for (uint32_t w = 0; w < 16; w++) {
        const uint32_t t = sh[w];
        if (w < wave) base += t;
        all += t;
    }
    *total = all;
    __syncthreads();
    return base + x - v;
}

// pbase[p] = sum_{p' < p} pcount[p'] (p <= np), pcursor = pbase, starts[nbt] = total entries
__global__ void __launch_bounds__(256) k_scan_parts(const uint32_t* pcount, uint32_t np, uint32_t* pbase,
                                                    uint32_t* pcursor, uint32_t* starts, uint32_t nbt) {
    __shared__ uint32_t sh[4];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t c0 = 0; c0 < np + 1; c0 += 256) {
        uint32_t idx = c0 + threadIdx.x;
        uint32_t v = idx < np ? pcount[idx] : 0;
        uint32_t total;
        uint32_t ex = block_exclusive_scan_256(v, sh, &total);
        if (idx <= np) {
            pbase[idx] = carry + ex;
            pcursor[idx] = carry + ex;
        }
        __syncthreads();
        if (threadIdx.x == 0) carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) starts[nbt] = carry;
}

// ---------------------------------------------------------------- k_partition (sort pass A)
// Workgroup = PART_T consecutive keys of one window.  Local ranks come from LDS atomics; each
// non-empty partition costs ONE global atomic per workgroup (space reservation), and the entries of a
// partition land in a contiguous run, so the 8-byte stores of a workgroup merge into full lines.
// TILE = entries per workgroup (256 threads x TILE / 256): 2048 for the windowed shape (256 partitions per window: runs of
// 8); 16384 over a table, whose 2^13 partitions would otherwise cost a global atomic for nearly every entry
template <uint32_t TILE>
__global__ void __launch_bounds__(256) k_partition(const uint32_t* keys, size_t n, uint32_t lo_bits, uint32_t hi_bits,
                                                   uint32_t* pcursor, uint2* tmp, uint32_t range_shift, uint32_t W,
                                                   uint32_t R, uint32_t tab_stride, const uint8_t* block_flags, uint32_t w_first) {
    uint32_t* cnt = h2_msm_smem;               // 2^hi_bits local counters, then the reserved bases
    const uint32_t nparts = 1u << hi_bits, w = blockIdx.y + w_first;
    // window of these PART_T rows: ranges are multiples of PART_T rows; key array W (if present) is the dominant-scalar
    // window, which is not cut into ranges.  Shifted-base table (tab_stride != 0): digit array w feeds window 0 with the
    // points of table level w; the dominant-scalar array feeds window 1 with the bases themselves (level 0)
    uint32_t vw = (R > 1) ? (w == W ? R * W : (uint32_t)(((size_t)blockIdx.x * TILE) >> range_shift) * W + w) : w;
    uint32_t level_base = 0;
    if (tab_stride) {
        vw = (w == W) ? 1u : 0u;
        level_base = (w == W) ? 0u : w * tab_stride;
    }
    for (uint32_t k = threadIdx.x; k < nparts; k += blockDim.x) cnt[k] = 0;
    const uint32_t ITEMS = TILE / 256;
    uint32_t key[TILE / 256], rank[TILE / 256];
    const size_t i0 = (size_t)blockIdx.x * TILE;
    // All of a lane's keys are requested before the first one is used (the loop used to wait for each key, and for the
    // block flag in front of it, in turn: 2 x ITEMS memory latencies per tile).  Rows beyond n read the last row and are
    // discarded; the flags of the tile's blocks (one per 256 rows, the same for every lane) go through LDS.
    uint32_t* sflag = cnt + nparts;   // ITEMS words: block k of the tile is flagged
    if (threadIdx.x < ITEMS) {
        const size_t row0 = i0 + (size_t)threadIdx.x * 256;   // BLOCK_ROWS == 256 == the stride of k
        sflag[threadIdx.x] = (block_flags != nullptr && w != W && row0 < n) ? block_flags[row0 / BLOCK_ROWS] : 0u;
    }
#pragma unroll
    for (uint32_t k = 0; k < ITEMS; k++) {
        const size_t i = i0 + k * 256 + threadIdx.x;
        key[k] = keys[(size_t)w * n + (i < n ? i : n - 1)];
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < ITEMS; k++) {
        const size_t i = i0 + k * 256 + threadIdx.x;
        // rows of a block that entered the dominant-value bucket as one point have no digit keys (k_digits)
        if (i >= n || sflag[k] != 0) key[k] = KEY_INVALID;
        rank[k] = 0;
        if (key[k] != KEY_INVALID) rank[k] = atomicAdd(&cnt[(key[k] & ~(SIGN_BIT | BLOCK_KEY)) >> lo_bits], 1u);
    }
    __syncthreads();
    // reserve the tile's run in every partition; a lane owns nparts / 256 of them and its reservations go out eight at a
    // time (one returning atomic at a time was up to 32 round trips to memory per tile)
    for (uint32_t k0 = threadIdx.x; k0 < nparts; k0 += 8 * blockDim.x) {
        uint32_t v[8], b[8];
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) v[j] = (k0 + j * blockDim.x < nparts) ? cnt[k0 + j * blockDim.x] : 0u;
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) {
            const uint32_t k = k0 + j * blockDim.x;
            b[j] = 0;
            if (TILE >= 16384) {   // nearly every partition receives entries from a tile this large: no branch, no wait
                if (k < nparts) b[j] = atomicAdd(&pcursor[(vw << hi_bits) + k], v[j]);
            } else if (v[j]) {
                b[j] = atomicAdd(&pcursor[(vw << hi_bits) + k], v[j]);
            }
        }
#pragma unroll
        for (uint32_t j = 0; j < 8; j++)
            if (k0 + j * blockDim.x < nparts) cnt[k0 + j * blockDim.x] = b[j];
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < ITEMS; k++) {
        if (key[k] == KEY_INVALID) continue;
        uint32_t bucket = key[k] & ~(SIGN_BIT | BLOCK_KEY);
        const uint32_t row = (uint32_t)(i0 + k * 256 + threadIdx.x);
        // a row that stands for its whole block of BLOCK_ROWS bases (k_digits) refers to the block's tabulated sum
        const uint32_t i = (key[k] & BLOCK_KEY) ? BLOCK_INDEX0 + row / BLOCK_ROWS : level_base + row;
        tmp[cnt[bucket >> lo_bits] + rank[k]] = make_uint2(i | (key[k] & SIGN_BIT), bucket & ((1u << lo_bits) - 1));
    }
}

// ---------------------------------------------------------------- two-level partition over a table
// k_partition leaves 2 x 10^8 separate 8-byte stores at 2^24 (a tile of 16384 keys has two entries per partition, written
// by different lanes at different times).  Here the 2^hi_bits partitions are reached in two passes of 2^a_bits groups and
// 2^b_bits partitions per group; each pass reorders its tile in LDS first, so a lane's neighbours write neighbouring
// entries and a run leaves the CU as whole cache lines.  Entries between the passes: (reference | sign, bucket).
// aux (after the entries of tmpa): cursor_a[G], tile_start[G + 1] (tiles of pass B per group, prefix sums).
__global__ void __launch_bounds__(PARTA_GROUPS_MAX) k_part_init(const uint32_t* pbase, uint32_t a_bits, uint32_t b_bits,
                                                                uint32_t* cursor_a, uint32_t* tile_start) {
    __shared__ uint32_t tiles[PARTA_GROUPS_MAX];
    const uint32_t G = 1u << a_bits, a = threadIdx.x;
    if (a < G) {
        const uint32_t gs = pbase[a << b_bits], ge = pbase[(a + 1) << b_bits];
        cursor_a[a] = gs;
        tiles[a] = (ge - gs + PART_AB_T - 1) / PART_AB_T;
    }
    __syncthreads();
    if (a == 0) {
        uint32_t run = 0;
        for (uint32_t k = 0; k < G; k++) {
            tile_start[k] = run;
            run += tiles[k];
        }
        tile_start[G] = run;
    }
}

// reorder the tile's `total` staged entries group by group and write them out: consecutive staged entries of one group go
// to consecutive addresses
__device__ __forceinline__ void part_ab_flush(const uint2* stage, uint32_t total, const uint32_t* loff, const uint32_t* gbase,
                                              uint32_t shift, uint32_t mask, bool strip, uint32_t lo_mask, uint2* out) {
    for (uint32_t idx = threadIdx.x; idx < total; idx += 256) {
        uint2 e = stage[idx];
        const uint32_t g = (e.y >> shift) & mask;
        const uint32_t dst = gbase[g] + (idx - loff[g]);
        if (strip) e.y &= lo_mask;
        out[dst] = e;
    }
}

// exclusive scan of hist[0..bins) (bins <= 128, two waves) -> loff, total; reserves the tile's run in every group (every
// lane of the workgroup calls this: it contains a barrier)
__device__ __forceinline__ void part_ab_reserve(const uint32_t* hist, uint32_t* loff, uint32_t* gbase, uint32_t bins,
                                                uint32_t* cursors, uint32_t* total_out, uint32_t* wsum) {
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    uint32_t v = 0, x = 0;
    if (t < 128) {
        v = t < bins ? hist[t] : 0u;
        x = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t y = __shfl_up(x, off, 64);
            if (lane >= (uint32_t)off) x += y;
        }
        if (lane == 63) wsum[wave] = x;
    }
    __syncthreads();
    if (t < 128) {
        const uint32_t base = wave ? wsum[0] : 0u;
        if (t < bins) {
            loff[t] = base + x - v;
            gbase[t] = v ? atomicAdd(&cursors[t], v) : 0u;
        }
        if (t == 127) *total_out = base + x;
    }
}

__global__ void __launch_bounds__(256) k_part_a(const uint32_t* keys, size_t n, uint32_t lo_bits, uint32_t b_bits,
                                                uint32_t a_bits, uint32_t* cursor_a, uint2* tmpa, uint32_t tab_stride,
                                                const uint8_t* block_flags) {
    constexpr uint32_t ITEMS = PART_AB_T / 256;
    __shared__ uint32_t hist[PARTA_GROUPS_MAX], loff[PARTA_GROUPS_MAX], gbase[PARTA_GROUPS_MAX], sflag[ITEMS], total_sh, wsum[2];
    uint2* stage = (uint2*)h2_msm_smem;  // PART_AB_T entries
    const uint32_t w = blockIdx.y, G = 1u << a_bits;
    const size_t i0 = (size_t)blockIdx.x * PART_AB_T;
    const uint32_t level_base = w * tab_stride;
    if (threadIdx.x < G) hist[threadIdx.x] = 0;
    if (threadIdx.x < ITEMS) {
        const size_t row0 = i0 + (size_t)threadIdx.x * 256;
        sflag[threadIdx.x] = (block_flags != nullptr && row0 < n) ? block_flags[row0 / BLOCK_ROWS] : 0u;
    }
    uint32_t key[ITEMS], rank[ITEMS];
#pragma unroll
    for (uint32_t k = 0; k < ITEMS; k++) {
        const size_t i = i0 + k * 256 + threadIdx.x;
        key[k] = keys[(size_t)w * n + (i < n ? i : n - 1)];
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < ITEMS; k++) {
        const size_t i = i0 + k * 256 + threadIdx.x;
        if (i >= n || sflag[k] != 0) key[k] = KEY_INVALID;
        rank[k] = 0;
        if (key[k] != KEY_INVALID) rank[k] = atomicAdd(&hist[((key[k] & ~(SIGN_BIT | BLOCK_KEY)) >> lo_bits) >> b_bits], 1u);
    }
    __syncthreads();
    part_ab_reserve(hist, loff, gbase, G, cursor_a, &total_sh, wsum);
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < ITEMS; k++) {
        if (key[k] == KEY_INVALID) continue;
        const uint32_t bucket = key[k] & ~(SIGN_BIT | BLOCK_KEY);
        const uint32_t row = (uint32_t)(i0 + k * 256 + threadIdx.x);
        const uint32_t ref = (key[k] & BLOCK_KEY) ? BLOCK_INDEX0 + row / BLOCK_ROWS : level_base + row;
        stage[loff[(bucket >> lo_bits) >> b_bits] + rank[k]] = make_uint2(ref | (key[k] & SIGN_BIT), bucket);
    }
    __syncthreads();
    part_ab_flush(stage, total_sh, loff, gbase, lo_bits + b_bits, G - 1, false, 0, tmpa);
}

__global__ void __launch_bounds__(256) k_part_b(const uint2* tmpa, const uint32_t* pbase, uint32_t lo_bits, uint32_t b_bits,
                                                uint32_t a_bits, const uint32_t* tile_start, uint32_t* pcursor, uint2* tmp) {
    constexpr uint32_t ITEMS = PART_AB_T / 256;
    __shared__ uint32_t hist[PARTA_GROUPS_MAX], loff[PARTA_GROUPS_MAX], gbase[PARTA_GROUPS_MAX], total_sh, group_sh, wsum[2];
    uint2* stage = (uint2*)h2_msm_smem;
    const uint32_t G = 1u << a_bits, P = 1u << b_bits;
    if (blockIdx.x >= tile_start[G]) return;
    if (threadIdx.x < G && tile_start[threadIdx.x] <= blockIdx.x && blockIdx.x < tile_start[threadIdx.x + 1]) group_sh = threadIdx.x;
    if (threadIdx.x < P) hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t a = group_sh;
    const uint32_t gs = pbase[a << b_bits], ge = pbase[(a + 1) << b_bits];
    const uint32_t e0 = gs + (blockIdx.x - tile_start[a]) * PART_AB_T;
    uint2 ent[ITEMS];
    uint32_t rank[ITEMS];
#pragma unroll
    for (uint32_t k = 0; k < ITEMS; k++) {
        const uint32_t e = e0 + k * 256 + threadIdx.x;
        ent[k] = tmpa[e < ge ? e : ge - 1];
    }
#pragma unroll
    for (uint32_t k = 0; k < ITEMS; k++) {
        const uint32_t e = e0 + k * 256 + threadIdx.x;
        rank[k] = e < ge ? atomicAdd(&hist[(ent[k].y >> lo_bits) & (P - 1)], 1u) : 0xffffffffu;
    }
    __syncthreads();
    part_ab_reserve(hist, loff, gbase, P, pcursor + (a << b_bits), &total_sh, wsum);
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < ITEMS; k++)
        if (rank[k] != 0xffffffffu) stage[loff[(ent[k].y >> lo_bits) & (P - 1)] + rank[k]] = ent[k];
    __syncthreads();
    part_ab_flush(stage, total_sh, loff, gbase, lo_bits, P - 1, true, (1u << lo_bits) - 1, tmp);
}

// ---------------------------------------------------------------- k_bucket_sort (sort pass B)
// One workgroup per partition: histogram of the low bucket bits in LDS, exclusive scan -> the start
// offset of every bucket of the partition (written to `starts`), then the entries are placed.
// A partition far above the average size means a skewed column (a witness column that is mostly one value:
// every such scalar lands in one bucket): its entries would serialise on one LDS counter, so the counter updates
// are aggregated per wave -- lanes holding the key of the first active lane are served by one atomic, four times over
// (a witness column of three or four distinct values is then served completely), and only what is left falls back to
// per-lane atomics.
static constexpr int SKEW_ROUNDS = 4;  // distinct keys of a wave served by one atomic each (a column of 3-4 values: all of them)
__device__ __forceinline__ uint32_t lds_count_aggregated(uint32_t* bins, uint32_t key, bool valid) {
    const uint32_t lane = threadIdx.x & 63;
    uint64_t active = __ballot(valid);
    uint32_t result = 0;
#pragma unroll
    for (int round = 0; round < SKEW_ROUNDS; round++) {
        if (active == 0) break;
        const int leader = __ffsll((unsigned long long)active) - 1;
        const uint32_t k = __shfl(key, leader, 64);
        const uint64_t same = __ballot(valid && key == k) & active;
        uint32_t base = 0;
        if ((int)lane == leader) base = atomicAdd(&bins[k], (uint32_t)__popcll(same));
        base = __shfl(base, leader, 64);
        if ((same >> lane) & 1) result = base + (uint32_t)__popcll(same & ((1ull << lane) - 1));
        active &= ~same;
    }
    if ((active >> lane) & 1) result = atomicAdd(&bins[key], 1u);
    return result;
}

__global__ void __launch_bounds__(SORT_T) k_bucket_sort(const uint2* tmp, const uint32_t* pbase, uint32_t lo_bits,
                                                        uint32_t hi_bits, uint32_t nb, uint32_t skew_threshold,
                                                        uint32_t hot_partition, uint32_t hot_wc, uint64_t hot_mask,
                                                        uint32_t* starts, uint32_t* sorted, uint32_t stage_cap) {
    uint32_t* bins = h2_msm_smem;  // 2^lo_bits counters, reused as cursors
    __shared__ uint32_t sh[16];
    const uint32_t nbins = 1u << lo_bits, p = blockIdx.x;
    const uint32_t e0 = pbase[p], e1 = pbase[p + 1];
    bool is_hot_partition = p == hot_partition;
    if (hot_wc) {  // fused shape: the dominant-scalar window of column j is window j * hot_wc + hot_wc - 1
        const uint32_t w = p >> hi_bits;
        is_hot_partition = (p & ((1u << hi_bits) - 1)) == 0 && (w % hot_wc) == hot_wc - 1 && ((hot_mask >> (w / hot_wc)) & 1);
    }
    if (is_hot_partition) {
        // the dominant scalar's window: every entry sits in bin 0, so there is nothing to sort -- k_copy_hot moves the
        // references with the whole chip instead of this one workgroup; only the bucket starts are written here
        const uint32_t w = p >> hi_bits, hi = p & ((1u << hi_bits) - 1);
        for (uint32_t b = threadIdx.x; b < nbins; b += SORT_T) starts[(size_t)w * nb + ((hi << lo_bits) | b)] = b ? e1 : e0;
        return;
    }
    const bool skewed = (e1 - e0) > skew_threshold;  // uniform over the workgroup
    for (uint32_t k = threadIdx.x; k < nbins; k += SORT_T) bins[k] = 0;
    __syncthreads();
    if (!skewed) {
        // (eight loads in flight per lane: one at a time, a partition of 24 000 entries was 24 memory latencies per pass)
        for (uint32_t e = e0 + threadIdx.x; e < e1; e += 8 * SORT_T) {
            uint32_t key[8];
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) key[j] = tmp[min(e + j * SORT_T, e1 - 1)].y;
#pragma unroll
            for (uint32_t j = 0; j < 8; j++)
                if (e + j * SORT_T < e1) atomicAdd(&bins[key[j]], 1u);
        }
    } else {
        for (uint32_t e = e0 + threadIdx.x; e < e1 + (SORT_T - 1); e += 4 * SORT_T) {  // whole waves stay in the loop
            uint32_t key[4];
#pragma unroll
            for (int j = 0; j < 4; j++) key[j] = (e + j * SORT_T < e1) ? tmp[e + j * SORT_T].y : 0;
#pragma unroll
            for (int j = 0; j < 4; j++) lds_count_aggregated(bins, key[j], e + j * SORT_T < e1);
        }
    }
    __syncthreads();
    // exclusive scan of the nbins (<= 4096) counters: (nbins / SORT_T) consecutive bins per thread
    const uint32_t per = (nbins + SORT_T - 1) / SORT_T;
    uint32_t local[4] = {0, 0, 0, 0}, sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        if (k < per) {
            uint32_t b = threadIdx.x * per + k;
            uint32_t v = b < nbins ? bins[b] : 0;
            local[k] = sum;
            sum += v;
        }
    }
    __syncthreads();
    // bucket index of bin b of partition p: window w = p >> hi_bits, bucket = ((p & (2^hi_bits - 1)) << lo_bits) | b
    const uint32_t w = p >> hi_bits, hi = p & ((1u << hi_bits) - 1);
    {
        uint32_t total;
        uint32_t ex = block_exclusive_scan_1024(sum, sh, &total);
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            if (k < per) {
                uint32_t b = threadIdx.x * per + k;
                if (b < nbins) {
                    uint32_t st = e0 + ex + local[k];
                    bins[b] = st;
                    starts[(size_t)w * nb + ((hi << lo_bits) | b)] = st;
                }
            }
        }
    }
    __syncthreads();
    if (!skewed) {
        // a partition that fits the staging area is put in order in LDS and leaves as whole lines: placed directly, its
        // entries are 4-byte stores at 2^lo_bits different frontiers (64 transactions per wave and store)
        uint32_t* stage = bins + nbins;
        const bool staged = (e1 - e0) <= stage_cap;   // uniform over the workgroup
        for (uint32_t e = e0 + threadIdx.x; e < e1; e += 8 * SORT_T) {
            uint2 v[8];
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) v[j] = tmp[min(e + j * SORT_T, e1 - 1)];
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) {
                if (e + j * SORT_T >= e1) continue;
                const uint32_t pos = atomicAdd(&bins[v[j].y], 1u);
                if (staged)
                    stage[pos - e0] = v[j].x;
                else
                    sorted[pos] = v[j].x;
            }
        }
        if (staged) {
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < e1 - e0; i += SORT_T) sorted[e0 + i] = stage[i];
        }
    } else {
        for (uint32_t e = e0 + threadIdx.x; e < e1 + (SORT_T - 1); e += 4 * SORT_T) {
            uint2 v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = (e + j * SORT_T < e1) ? tmp[e + j * SORT_T] : make_uint2(0, 0);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const bool valid = e + j * SORT_T < e1;
                uint32_t slot = lds_count_aggregated(bins, v[j].y, valid);
                if (valid) sorted[slot] = v[j].x;
            }
        }
    }
}

__global__ void __launch_bounds__(256) k_copy_hot(const uint2* tmp, const uint32_t* pbase, uint32_t hot_partition,
                                                  uint32_t* sorted) {
    const uint32_t e0 = pbase[hot_partition], e1 = pbase[hot_partition + 1];
    for (uint32_t e = e0 + blockIdx.x * blockDim.x + threadIdx.x; e < e1; e += gridDim.x * blockDim.x) sorted[e] = tmp[e].x;
}

// ---------------------------------------------------------------- k_acc_slice (hot loop)
template <int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_acc_slice(const Affine* bases, const Affine* block_sums, const uint32_t* sorted,
                                                   const uint32_t* starts, uint32_t nbt, uint32_t log_s, XYZZ* partials) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;  // slice index
    const uint32_t total = starts[nbt];                        // number of (non-zero digit) entries
    uint32_t e = s << log_s;
    if (e >= total) return;
    uint32_t end = e + (1u << log_s);
    if (end > total) end = total;
    // bucket of the first entry: largest b with starts[b] <= e (ties -> the last one, which is non-empty)
    uint32_t lo = 0, hi = nbt;  // invariant: starts[lo] <= e < starts[hi]
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (starts[mid] <= e)
            lo = mid;
        else
            hi = mid;
    }
    uint32_t b = lo;
    uint32_t bnext = starts[b + 1];  // first entry of the next non-empty bucket
    XYZZ acc = xyzz_identity();
    for (; e < end; e++) {
        if (e == bnext) {  // bucket boundary inside the slice (rare: once per ~n/2^(c-1) entries)
            xyzz_store(partials + (b + s), acc);
            acc = xyzz_identity();
            b++;
            bnext = starts[b + 1];
            if (bnext == e) {
                // a run of empty buckets (sparse columns: a few distinct values spread over 2^(c-1) buckets): locate the
                // bucket that owns entry e by bisection instead of walking the run one dependent load at a time
                uint32_t l2 = b, h2 = nbt;  // starts[l2] <= e < starts[h2]
                while (h2 - l2 > 1) {
                    uint32_t mid = (l2 + h2) >> 1;
                    if (starts[mid] <= e)
                        l2 = mid;
                    else
                        h2 = mid;
                }
                b = l2;
                bnext = starts[b + 1];
            }
        }
        uint32_t ref = sorted[e];
        const uint32_t idx = ref & ~SIGN_BIT;
        Affine p = affine_load(idx >= BLOCK_INDEX0 ? block_sums + (idx - BLOCK_INDEX0) : bases + idx);
        acc = xyzz_madd(acc, p, (ref & SIGN_BIT) != 0);
    }
    xyzz_store(partials + (b + s), acc);
}

// ---------------------------------------------------------------- k_finish / k_finish_heavy
// These kernels and k_reduce are chains and trees of dependent point additions with few of them in flight: they run on
// QUADS (ec_quad.hpp) -- four lanes share one chain and each addition costs 4 dependent field products instead of 14.
// `q` = lane inside the quad, `qd` = quad inside the workgroup; loads are replicated over the quad, lane k stores
// coordinate k.
__device__ __forceinline__ void xyzz_store_q(XYZZ* p, const XYZZ& v, uint32_t q) {
    fp_store(&p->x + q, quad_pick(q, v.x, v.y, v.zz, v.zzz));
}

// sum over the QUADS quads of a workgroup, result in quad 0 (sh: one XYZZ per quad)
template <uint32_t QUADS>
__device__ __forceinline__ XYZZ quad_tree_sum(XYZZ acc, XYZZ* sh, uint32_t qd, uint32_t q) {
    for (uint32_t off = QUADS / 2; off >= 1; off >>= 1) {
        if (qd >= off && qd < 2 * off) xyzz_store_q(sh + qd, acc, q);
        __syncthreads();
        if (qd < off) acc = xyzz_add_q(acc, xyzz_load(sh + qd + off), q);
        __syncthreads();
    }
    return acc;
}

// bucket b's partials are slots b + first .. b + last, first/last = slices of its first/last entry
__global__ void __launch_bounds__(256) k_finish(const XYZZ* partials, const uint32_t* starts, uint32_t nbt,
                                                uint32_t log_s, XYZZ* buckets, uint32_t* heavy_list,
                                                uint32_t* heavy_count) {
    const uint32_t q = threadIdx.x & 3;
    uint32_t b = (blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    if (b >= nbt) return;
    uint32_t e0 = starts[b], e1 = starts[b + 1];
    XYZZ acc = xyzz_identity();
    if (e1 > e0) {
        uint32_t first = e0 >> log_s, last = (e1 - 1) >> log_s;
        if (last - first + 1 > FINISH_SERIAL) {
            if (q == 0) heavy_list[atomicAdd(heavy_count, 1u)] = b;
            return;
        }
        acc = xyzz_load(partials + (b + first));
        XYZZ nxt = first < last ? xyzz_load(partials + (b + first + 1)) : xyzz_identity();
        for (uint32_t sl = first + 1; sl <= last; sl++) {
            const XYZZ cur = nxt;
            if (sl < last) nxt = xyzz_load(partials + (b + sl + 1));  // in flight during the addition below
            acc = xyzz_add_q(acc, cur, q);
        }
    }
    xyzz_store_q(buckets + b, acc, q);
}

// The same fold with ONE LANE per bucket, for bucket counts that fill the chip many times over (2^21 buckets of a
// shifted-base table at 2^24): there the fold is bound by throughput, and a quad spends four lanes on the latency of
// one chain (2^24 over a table, batch of 8: 21.3 -> 20.9 ms per MSM; from 2^19 buckets the lanes win by 1-2 %, at 2^17 the two forms tie,
// below the quads win).
__global__ void __launch_bounds__(256) k_finish_lane(const XYZZ* partials, const uint32_t* starts, uint32_t nbt,
                                                     uint32_t log_s, XYZZ* buckets, uint32_t* heavy_list,
                                                     uint32_t* heavy_count) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nbt) return;
    const uint32_t e0 = starts[b], e1 = starts[b + 1];
    XYZZ acc = xyzz_identity();
    if (e1 > e0) {
        const uint32_t first = e0 >> log_s, last = (e1 - 1) >> log_s;
        if (last - first + 1 > FINISH_SERIAL) {
            heavy_list[atomicAdd(heavy_count, 1u)] = b;
            return;
        }
        acc = xyzz_load(partials + (b + first));
        for (uint32_t sl = first + 1; sl <= last; sl++) acc = xyzz_add(acc, xyzz_load(partials + (b + sl)));
    }
    xyzz_store(buckets + b, acc);
}

// Buckets on the heavy list with <= MID_MAX partials (a narrow column: thousands of buckets with a few hundred entries
// each) take ONE WAVE each: its 16 quads fold the partials strided by 16, then a 4-level tree over the wave by lane
// shuffles; a workgroup per such bucket (k_finish_heavy) would spend a 6-level tree on one partial per quad and walk the
// list 64 buckets at a time (2^20 4-bit values: 2048 such buckets, 1.1 ms -> 0.05 ms).
__device__ __forceinline__ XYZZ xyzz_shfl_down(const XYZZ& v, uint32_t delta) {
    XYZZ r;
    const uint32_t* src = (const uint32_t*)&v;
    uint32_t* dst = (uint32_t*)&r;
#pragma unroll
    for (int i = 0; i < 32; i++) dst[i] = (uint32_t)__shfl_down((int)src[i], delta, 64);
    return r;
}

__global__ void __launch_bounds__(256) k_finish_mid(const XYZZ* partials, const uint32_t* starts, uint32_t log_s,
                                                    const uint32_t* heavy_list, const uint32_t* heavy_count,
                                                    XYZZ* buckets) {
    const uint32_t lane = threadIdx.x & 63, q = lane & 3, qd = lane >> 2;  // 16 quads per wave
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t nheavy = *heavy_count;
    for (uint32_t h = wave; h < nheavy; h += nwaves) {
        const uint32_t b = heavy_list[h];
        const uint32_t first = starts[b] >> log_s, last = (starts[b + 1] - 1) >> log_s;
        if (last - first + 1 > MID_MAX) continue;  // uniform over the wave: k_finish_heavy's
        XYZZ acc = xyzz_identity();
        for (uint32_t sl = first + qd; sl <= last; sl += 16) acc = xyzz_add_q(acc, xyzz_load(partials + (b + sl)), q);
        for (uint32_t off = 8; off >= 1; off >>= 1) {
            const XYZZ other = xyzz_shfl_down(acc, 4 * off);  // every lane takes part in the exchange
            if (qd < off) acc = xyzz_add_q(acc, other, q);
        }
        if (qd == 0) xyzz_store_q(buckets + b, acc, q);
    }
}

// Heavy bucket h with P > MID_MAX partials is shared by NP = ceil(P / 256) (at most HEAVY_SPLIT) workgroups: part y folds the
// slots first + y, first + y + NP, ... and leaves the result in slot first + y -- a slot of its own set, written after
// its last read, so the parts of one bucket never race.  k_finish_heavy2 then folds the NP leading slots.
__device__ __forceinline__ uint32_t heavy_parts(uint32_t first, uint32_t last) {
    const uint32_t np = (last - first + 256) / 256;  // ceil((last - first + 1) / 256)
    return np < HEAVY_SPLIT ? np : HEAVY_SPLIT;
}
__global__ void __launch_bounds__(256) k_finish_heavy(XYZZ* partials, const uint32_t* starts, uint32_t log_s,
                                                      const uint32_t* heavy_list, const uint32_t* heavy_count) {
    __shared__ XYZZ sh[64];
    const uint32_t q = threadIdx.x & 3, qd = threadIdx.x >> 2, part = blockIdx.y;
    uint32_t nheavy = *heavy_count;
    for (uint32_t h = blockIdx.x; h < nheavy; h += gridDim.x) {
        uint32_t b = heavy_list[h];
        uint32_t first = starts[b] >> log_s, last = (starts[b + 1] - 1) >> log_s;
        if (last - first + 1 <= MID_MAX) continue;  // k_finish_mid's
        const uint32_t np = heavy_parts(first, last);
        if (part >= np) continue;  // uniform over the workgroup
        XYZZ acc = xyzz_identity();
        for (uint32_t sl = first + part + qd * np; sl <= last; sl += 64 * np)
            acc = xyzz_add_q(acc, xyzz_load(partials + (b + sl)), q);
        acc = quad_tree_sum<64>(acc, sh, qd, q);
        if (qd == 0) xyzz_store_q(partials + (b + first + part), acc, q);
    }
}

__global__ void __launch_bounds__(4 * HEAVY_SPLIT) k_finish_heavy2(const XYZZ* partials, const uint32_t* starts,
                                                                    uint32_t log_s, const uint32_t* heavy_list,
                                                                    const uint32_t* heavy_count, XYZZ* buckets) {
    __shared__ XYZZ sh[HEAVY_SPLIT];
    const uint32_t q = threadIdx.x & 3, qd = threadIdx.x >> 2;
    uint32_t nheavy = *heavy_count;
    for (uint32_t h = blockIdx.x; h < nheavy; h += gridDim.x) {
        uint32_t b = heavy_list[h];
        uint32_t first = starts[b] >> log_s, last = (starts[b + 1] - 1) >> log_s;
        if (last - first + 1 <= MID_MAX) continue;  // k_finish_mid's
        XYZZ acc = qd < heavy_parts(first, last) ? xyzz_load(partials + (b + first + qd)) : xyzz_identity();
        acc = quad_tree_sum<HEAVY_SPLIT>(acc, sh, qd, q);
        if (qd == 0) xyzz_store_q(buckets + b, acc, q);
    }
}

// ---------------------------------------------------------------- k_reduce
// window w, group g: sum over this group's buckets of (b + 1) * B_b   (b = index inside the window).  A quad walks
// REDUCE_QM consecutive buckets by summation by parts (arithmetic.rs:98-106), lifts its sum by the offset of its first
// bucket, and the quads of the workgroup are folded by a tree.  The last workgroup of a window to finish (a counter per
// window) folds the RG group results, so the host reads one point per window.
__device__ __forceinline__ XYZZ xyzz_load_coherent(const XYZZ* p) {  // written by other workgroups of this launch
    XYZZ r;
    const volatile uint4* src = (const volatile uint4*)p;
    uint4* dst = (uint4*)&r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        dst[i].x = src[i].x;
        dst[i].y = src[i].y;
        dst[i].z = src[i].z;
        dst[i].w = src[i].w;
    }
    return r;
}

__global__ void __launch_bounds__(REDUCE_T) k_reduce(const XYZZ* buckets, uint32_t nb, uint32_t RG, uint32_t qm,
                                                     XYZZ* groups, XYZZ* winsum, uint32_t* rcount) {
    __shared__ XYZZ sh[REDUCE_T / 4];
    __shared__ uint32_t last_flag;
    const uint32_t q = threadIdx.x & 3, qd = threadIdx.x >> 2, g = blockIdx.x, w = blockIdx.y;
    const XYZZ* B = buckets + (size_t)w * nb;
    uint32_t k0 = (g * (REDUCE_T / 4) + qd) * qm;
    XYZZ res = xyzz_identity();
    if (k0 < nb) {
        uint32_t k1 = k0 + qm;
        if (k1 > nb) k1 = nb;
        XYZZ running = xyzz_identity(), acc = xyzz_identity();
        XYZZ nxt = xyzz_load(B + (k1 - 1));
        for (uint32_t b = k1; b-- > k0;) {
            const XYZZ cur = nxt;
            if (b > k0) nxt = xyzz_load(B + (b - 1));  // in flight during the two additions below
            running = xyzz_add_q(running, cur, q);
            acc = xyzz_add_q(acc, running, q);
        }
        // acc = sum (b - k0 + 1) * B_b ;  running = sum B_b
        res = acc;
        if (k0 != 0 && !xyzz_is_identity(running)) res = xyzz_add_q(res, xyzz_mul_u32_q(running, k0, q), q);
    }
    res = quad_tree_sum<REDUCE_T / 4>(res, sh, qd, q);
    if (RG == 1) {
        if (qd == 0) xyzz_store_q(winsum + w, res, q);
        return;
    }
    if (qd == 0) xyzz_store_q(groups + (size_t)w * RG + g, res, q);
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last_flag = atomicAdd(rcount + w, 1u) == RG - 1 ? 1u : 0u;
    __syncthreads();
    if (!last_flag) return;
    __threadfence();
    res = xyzz_identity();
    for (uint32_t g2 = qd; g2 < RG; g2 += REDUCE_T / 4) res = xyzz_add_q(res, xyzz_load_coherent(groups + (size_t)w * RG + g2), q);
    res = quad_tree_sum<REDUCE_T / 4>(res, sh, qd, q);
    if (qd == 0) xyzz_store_q(winsum + w, res, q);
}

// ---------------------------------------------------------------- reduce by bit planes (shifted-base tables)
// k_reduce's quads each lift their chunk by the index of its first bucket -- a double-and-add over up to 16-21 bits, ~80
// dependent product levels, the longest part of a chain that is latency and nothing else (273 us for the 2^16 buckets of a
// 2^20 MSM however few points they hold).  The weights factor instead:
//     sum_b (b + 1) B_b  =  sum_c acc_c  +  qm * sum_c c * R_c          (chunk c = buckets c qm .. c qm + qm - 1,
//                                                                         acc_c = sum (b - c qm + 1) B_b,  R_c = sum B_b)
//     sum_c c * R_c      =  sum_p 2^p P_p,   P_p = sum of the R_c whose index has bit p set
// k_reduce_chunks computes acc_c / R_c by summation by parts (2 qm additions per quad) and tree-sums the acc_c of a
// workgroup; k_reduce_planes forms the P_p -- UNWEIGHTED sums, a few elements per quad and two trees deep -- and the sum
// of the workgroups' acc sums; the host adds the planes up by Horner (log2(chunks) doublings and additions on one core,
// ~20 us, next to the window Horner it already does).  Same group element, a different (equally valid) Jacobian
// representative of it; ~4 extra additions per chunk of work, a third of the dependent depth.
__global__ void __launch_bounds__(REDUCE_T) k_reduce_chunks(const XYZZ* buckets, uint32_t nb, uint32_t qm, uint32_t chunks,
                                                            XYZZ* R, XYZZ* group_acc) {
    __shared__ XYZZ sh[REDUCE_T / 4];
    const uint32_t q = threadIdx.x & 3, qd = threadIdx.x >> 2;
    const uint32_t c = blockIdx.x * (REDUCE_T / 4) + qd;
    XYZZ acc = xyzz_identity();
    if (c < chunks) {
        const uint32_t k0 = c * qm;
        uint32_t k1 = k0 + qm;
        if (k1 > nb) k1 = nb;
        XYZZ running = xyzz_identity();
        XYZZ nxt = xyzz_load(buckets + (k1 - 1));
        for (uint32_t b = k1; b-- > k0;) {
            const XYZZ cur = nxt;
            if (b > k0) nxt = xyzz_load(buckets + (b - 1));  // in flight during the two additions below
            running = xyzz_add_q(running, cur, q);
            acc = xyzz_add_q(acc, running, q);
        }
        xyzz_store_q(R + c, running, q);
    }
    acc = quad_tree_sum<REDUCE_T / 4>(acc, sh, qd, q);
    if (qd == 0) xyzz_store_q(group_acc + blockIdx.x, acc, q);
}

// grid (plane_seg, planes): plane p < nbits sums the chunk sums R[c] with bit p of c set -- quad t of the plane takes the
// L consecutive members t L .. t L + L - 1 of that half of the index space; the last plane sums the group_acc of
// k_reduce_chunks' workgroups (segment 0 alone).  Each workgroup's tree result goes to part[p][segment]; the last
// workgroup of a plane to finish (a counter per plane) folds the segments and writes out[p].
__global__ void __launch_bounds__(REDUCE_T) k_reduce_planes(const XYZZ* R, uint32_t chunks, uint32_t nbits, const XYZZ* group_acc,
                                                            uint32_t groups, uint32_t L, XYZZ* part, uint32_t* counters, XYZZ* out) {
    __shared__ XYZZ sh[REDUCE_T / 4];
    __shared__ uint32_t last_flag;
    const uint32_t q = threadIdx.x & 3, qd = threadIdx.x >> 2, seg = blockIdx.x, p = blockIdx.y, nseg = gridDim.x;
    XYZZ acc = xyzz_identity();
    uint32_t expected = nseg;
    if (p == nbits) {
        expected = 1;
        if (seg != 0) return;
        for (uint32_t g = qd; g < groups; g += REDUCE_T / 4) acc = xyzz_add_q(acc, xyzz_load(group_acc + g), q);
    } else {
        const uint32_t half = 1u << (nbits - 1), low_mask = (1u << p) - 1;
        const uint32_t t = seg * (REDUCE_T / 4) + qd;
        uint32_t j = t * L;
        const uint32_t j1 = j + L < half ? j + L : half;
        for (; j < j1; j++) {
            const uint32_t c = ((j >> p) << (p + 1)) | (1u << p) | (j & low_mask);
            if (c < chunks) acc = xyzz_add_q(acc, xyzz_load(R + c), q);
        }
    }
    acc = quad_tree_sum<REDUCE_T / 4>(acc, sh, qd, q);
    if (expected == 1) {
        if (qd == 0) xyzz_store_q(out + p, acc, q);
        return;
    }
    if (qd == 0) xyzz_store_q(part + (size_t)p * nseg + seg, acc, q);
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last_flag = atomicAdd(counters + p, 1u) == expected - 1 ? 1u : 0u;
    __syncthreads();
    if (!last_flag) return;
    __threadfence();
    acc = xyzz_identity();
    for (uint32_t s2 = qd; s2 < nseg; s2 += REDUCE_T / 4) acc = xyzz_add_q(acc, xyzz_load_coherent(part + (size_t)p * nseg + s2), q);
    acc = quad_tree_sum<REDUCE_T / 4>(acc, sh, qd, q);
    if (qd == 0) xyzz_store_q(out + p, acc, q);
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

// ---------------------------------------------------------------- compressed points (Params::{read, write})
// poly/commitment.rs:241-294 stores g and g_lagrange as `to_bytes()` = 32 bytes per point.  Convention (the layout of
// pairing_bn256@30b052f cannot be checked without its sources -- "parity unpinned"): x little-endian, bit 7 of byte 31 =
// parity of canonical y, identity = 32 zero bytes.  Decompression is one square root (y = rhs^((q+1)/4), q = 3 mod 4)
// per point: the reference does it with a rayon `parallelize` over `from_bytes`, here it is one lane per point.
__device__ __forceinline__ Fq fq_sqrt_candidate(const Fq& rhs) {
    // (q + 1) / 4, little-endian u32 limbs
    const uint32_t E[8] = {0xb61f3f52u, 0x4f082305u, 0x5a1c72a3u, 0x65e05aa4u,
                           0xa0605617u, 0x6e14116du, 0xb84c680au, 0x0c19139cu};
    Fq y = fp_one<FqParams>();
    for (int bit = 253; bit >= 0; bit--) {
        y = fp_sqr(y);
        if ((E[bit >> 5] >> (bit & 31)) & 1) y = fp_mul(y, rhs);
    }
    return y;
}

__global__ void __launch_bounds__(256) k_points_decompress(const uint32_t* bytes, size_t n, Affine* out, uint32_t* bad) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fq x;
    uint32_t any = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        x.l[k] = bytes[8 * i + k];
        any |= x.l[k];
    }
    Fq zero = fp_zero<FqParams>();
    if (any == 0) {  // identity
        fp_store(&out[i].x, zero);
        fp_store(&out[i].y, zero);
        return;
    }
    const uint32_t sign = x.l[7] >> 31;
    x.l[7] &= 0x7fffffffu;
    // x must be a canonical residue (< q)
    bool lt = false, decided = false;
#pragma unroll
    for (int k = 7; k >= 0; k--) {
        if (!decided && x.l[k] != FqParams::MOD[k]) {
            lt = x.l[k] < FqParams::MOD[k];
            decided = true;
        }
    }
    if (!lt) {
        atomicAdd(bad, 1u);
        fp_store(&out[i].x, zero);
        fp_store(&out[i].y, zero);
        return;
    }
    Fq xm = fp_to_mont(x);
    Fq three = fp_add(fp_add(fp_one<FqParams>(), fp_one<FqParams>()), fp_one<FqParams>());
    Fq rhs = fp_add(fp_mul(fp_sqr(xm), xm), three);
    Fq y = fq_sqrt_candidate(rhs);
    if (!fp_eq(fp_sqr(y), rhs)) {  // x is not the abscissa of a curve point
        atomicAdd(bad, 1u);
        fp_store(&out[i].x, zero);
        fp_store(&out[i].y, zero);
        return;
    }
    if ((fp_from_mont(y).l[0] & 1u) != sign) y = fp_neg(y);
    fp_store(&out[i].x, xm);
    fp_store(&out[i].y, y);
}

__global__ void __launch_bounds__(256) k_points_compress(const Affine* pts, size_t n, uint32_t* bytes) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine p = affine_load(pts + i);
    Fq x = fp_from_mont(p.x), y = fp_from_mont(p.y);
    if (fp_is_zero(x) && fp_is_zero(y)) {
#pragma unroll
        for (int k = 0; k < 8; k++) bytes[8 * i + k] = 0;
        return;
    }
    x.l[7] |= (y.l[0] & 1u) << 31;
#pragma unroll
    for (int k = 0; k < 8; k++) bytes[8 * i + k] = x.l[k];
}

int points_decompress_launch(const void* d_bytes, size_t n, uint64_t* d_out, uint32_t* d_bad, hipStream_t stream) {
    if (n == 0) return H2_OK;
    H2_HIP(hipMemsetAsync(d_bad, 0, 4, stream));
    hipLaunchKernelGGL(k_points_decompress, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                       (const uint32_t*)d_bytes, n, (Affine*)d_out, d_bad);
    H2_HIP(hipGetLastError());
    uint32_t bad = 0;
    H2_HIP(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, stream));
    H2_HIP(hipStreamSynchronize(stream));
    if (bad) {
        set_last_error("points_decompress: " + std::to_string(bad) + " encoding(s) are not curve points");
        return H2_ERR_INVALID;
    }
    return H2_OK;
}

int points_compress_launch(const uint64_t* d_points, size_t n, void* d_bytes, hipStream_t stream) {
    if (n == 0) return H2_OK;
    hipLaunchKernelGGL(k_points_compress, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                       (const Affine*)d_points, n, (uint32_t*)d_bytes);
    H2_HIP(hipGetLastError());
    return H2_OK;
}

// ---------------------------------------------------------------- fixed-base multiplication (Params::unsafe_setup)
// out[i] = [scalars[i]] B for one base B given as the table T[j] = [2^j] B, j < 254 (affine): the setup's
// g[i] = [s^i] G and g_lagrange[i] = [l_i(s)] G (poly/commitment.rs:67-112, a rayon `parallelize` with one variable-
// base multiplication per point there).  One lane per point: ~127 mixed additions against the shared table (every
// lane reads the same entry: a broadcast), then one Fq inversion (a^(q-2)) to normalise.
__device__ __forceinline__ Fq fq_inv_device(const Fq& a) {
    // q - 2, little-endian u32 limbs
    const uint32_t E[8] = {0xd87cfd45u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                           0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    Fq acc = fp_one<FqParams>();
#pragma unroll 1
    for (int bit = 253; bit >= 0; bit--) {
        acc = fp_sqr(acc);
        if ((E[bit >> 5] >> (bit & 31)) & 1) acc = fp_mul(acc, a);
    }
    return acc;
}

__global__ void __launch_bounds__(256) k_fixed_base_mul(const Fr* scalars, const Affine* table, size_t n, Affine* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fr s = fp_from_mont(fp_load(scalars + i));
    XYZZ acc = xyzz_identity();
#pragma unroll 1
    for (int bit = 0; bit < 254; bit++)
        if ((s.l[bit >> 5] >> (bit & 31)) & 1) acc = xyzz_madd(acc, affine_load(table + bit), false);
    Fq zero = fp_zero<FqParams>();
    if (fp_is_zero(acc.zz)) {  // scalar 0: the identity
        fp_store(&out[i].x, zero);
        fp_store(&out[i].y, zero);
        return;
    }
    // x = X / ZZ, y = Y / ZZZ with one inversion: t = 1 / ZZZ, 1 / ZZ = (ZZ * t)^2  (ZZ^3 = ZZZ^2)
    const Fq t = fq_inv_device(acc.zzz);
    const Fq u = fp_mul(acc.zz, t);
    fp_store(&out[i].x, fp_mul(acc.x, fp_sqr(u)));
    fp_store(&out[i].y, fp_mul(acc.y, t));
}

int fixed_base_mul_launch(const Fr* d_scalars, const uint64_t* d_table, size_t n, uint64_t* d_out, hipStream_t stream) {
    if (n == 0) return H2_OK;
    hipLaunchKernelGGL(k_fixed_base_mul, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_scalars,
                       (const Affine*)d_table, n, (Affine*)d_out);
    H2_HIP(hipGetLastError());
    return H2_OK;
}

// ---------------------------------------------------------------- shifted-base table build (h2_dev_bases_precompute)
// One lane per base point: level j = [2^(width of digit j - 1)] level j - 1 by Jacobian doublings (a = 0, 2M + 5S
// [dbl-2009-l]; BN254 G1 has prime order, so a doubling never meets the identity), X and Y parked in the table slot and
// Z kept per level; then ONE inversion for all levels of the point (Montgomery's trick) and x = X / Z^2, y = Y / Z^3.
// ~2300 field products per point: 0.3 s for 2^24 points, paid once per SRS.
static constexpr uint32_t TABLE_MAX_D = 32;
static constexpr uint32_t TABLE_MIN_D = 11;  // digits of <= 24 bits: 2^23 buckets

__global__ void __launch_bounds__(256) k_table_build(const Affine* bases, size_t n, size_t stride, uint32_t D, uint32_t c,
                                                     uint32_t wfull, Affine* table) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Affine p = affine_load(bases + i);
    fp_store(&table[i].x, p.x);
    fp_store(&table[i].y, p.y);
    if (affine_is_identity(p)) {
        const Fq zero = fp_zero<FqParams>();
        for (uint32_t j = 1; j < D; j++) {
            fp_store(&table[j * stride + i].x, zero);
            fp_store(&table[j * stride + i].y, zero);
        }
        return;
    }
    Fq z[TABLE_MAX_D], pre[TABLE_MAX_D];
    Fq X = p.x, Y = p.y, Z = fp_one<FqParams>();
#pragma unroll 1
    for (uint32_t j = 1; j < D; j++) {
        const uint32_t cw = c - ((j - 1) >= wfull ? 1u : 0u);
#pragma unroll 1
        for (uint32_t k = 0; k < cw; k++) {
            const Fq A = fp_sqr(X), B = fp_sqr(Y), C = fp_sqr(B);
            Fq Dd = fp_sub(fp_sub(fp_sqr(fp_add(X, B)), A), C);
            Dd = fp_dbl(Dd);
            const Fq E = fp_add(fp_dbl(A), A);
            const Fq F = fp_sqr(E);
            const Fq Z3 = fp_dbl(fp_mul(Y, Z));
            X = fp_sub(F, fp_dbl(Dd));
            const Fq C8 = fp_dbl(fp_dbl(fp_dbl(C)));
            Y = fp_sub(fp_mul(E, fp_sub(Dd, X)), C8);
            Z = Z3;
        }
        fp_store(&table[j * stride + i].x, X);
        fp_store(&table[j * stride + i].y, Y);
        z[j] = Z;
        pre[j] = j == 1 ? Z : fp_mul(pre[j - 1], Z);
    }
    if (D < 2) return;
    Fq inv = fq_inv_device(pre[D - 1]);
#pragma unroll 1
    for (uint32_t j = D - 1; j >= 1; j--) {
        const Fq zinv = j == 1 ? inv : fp_mul(inv, pre[j - 1]);
        inv = fp_mul(inv, z[j]);
        const Fq zi2 = fp_sqr(zinv);
        Affine* slot = table + (j * stride + i);
        fp_store(&slot->x, fp_mul(fp_load(&slot->x), zi2));
        fp_store(&slot->y, fp_mul(fp_load(&slot->y), fp_mul(zi2, zinv)));
    }
}

// sums of BLOCK_ROWS consecutive bases (blocks of a registered base set; see k_digits "dominant value over whole blocks")
__global__ void __launch_bounds__(64) k_block_sums(const Affine* bases, size_t nblocks, Affine* out) {
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    XYZZ acc = xyzz_identity();
#pragma unroll 1
    for (uint32_t k = 0; k < BLOCK_ROWS; k++) acc = xyzz_madd(acc, affine_load(bases + b * BLOCK_ROWS + k), false);
    const Fq zero = fp_zero<FqParams>();
    if (fp_is_zero(acc.zz)) {  // the identity (the bases of a block cancel): (0, 0)
        fp_store(&out[b].x, zero);
        fp_store(&out[b].y, zero);
        return;
    }
    const Fq t = fq_inv_device(acc.zzz);
    const Fq u = fp_mul(acc.zz, t);
    fp_store(&out[b].x, fp_mul(acc.x, fp_sqr(u)));
    fp_store(&out[b].y, fp_mul(acc.y, t));
}

// digits for a table over n points: fewest additions D * n plus the bucket-proportional tail (k_finish, k_reduce: ~5
// additions' worth per bucket), buckets 2^(c - 1) with c = ceil(255 / D) <= 23
static uint32_t table_default_digits(size_t n) {
    if (const char* env = getenv("H2_MSM_TABLE_DIGITS")) {
        int v = atoi(env);
        if (v >= (int)TABLE_MIN_D && v <= (int)TABLE_MAX_D) return (uint32_t)v;
    }
    double best = 1e300;
    uint32_t best_d = 16;
    for (uint32_t D = 12; D <= TABLE_MAX_D; D++) {
        const uint32_t c = (255 + D - 1) / D;
        // (~4.7 additions' worth per bucket since the bit-plane reduce -- 8 with the running-sum chains before it; measured
        // round 5, single / batch of 8: 2^20 14 digits 1.68 / 1.58 ms, 15 digits 1.78 / 1.64, 13 digits 1.79 / 1.61; 2^18 15
        // digits 0.72, 14 digits 0.80; 2^22 13 digits 5.62, 12 digits 6.09, 14 digits 6.07.  Below 2^18 rows, where the chains
        // of k_finish / k_reduce are latency whatever the bucket count, 4 ranks the measured optimum first: 2^16 16 digits
        // 0.40 ms, 15 digits 0.42)
        const double cost = (double)D * (double)n + (n < ((size_t)1 << 18) ? 4.0 : 4.7) * (double)(1u << (c - 1));
        if (cost < best) {
            best = cost;
            best_d = D;
        }
    }
    return best_d;
}

size_t bases_precompute_bytes(size_t n, uint32_t digits) {
    if (digits == 0) digits = table_default_digits(n);
    return (size_t)digits * n * sizeof(Affine);
}

// device memory this translation unit keeps for `ctx`'s device: shifted-base tables (all: they are keyed by device
// address) and the device copies of registered host SRS ranges
size_t msm_library_bytes(DeviceCtx* ctx) {
    size_t bytes = 0;
    {
        std::lock_guard<std::mutex> g(g_tab_mu);
        for (const auto& kv : g_tables)
            bytes += (size_t)kv.second.D * kv.second.n * sizeof(Affine) + (kv.second.blocks ? kv.second.n / BLOCK_ROWS * sizeof(Affine) : 0);
    }
    std::lock_guard<std::mutex> g2(ctx->shared->mu);
    for (const auto& kv : ctx->resident) bytes += kv.second.len * sizeof(Affine);
    return bytes;
}

int bases_forget(const uint64_t* d_bases) {
    Affine *old = nullptr, *old_blocks = nullptr;
    {
        std::lock_guard<std::mutex> g(g_tab_mu);
        auto it = g_tables.find(d_bases);
        if (it == g_tables.end()) return H2_OK;
        old = const_cast<Affine*>(it->second.table);
        old_blocks = const_cast<Affine*>(it->second.blocks);
        g_tables.erase(it);
    }
    H2_HIP(hipDeviceSynchronize());  // nothing in flight reads the table
    H2_HIP(hipFree(old));
    if (old_blocks) H2_HIP(hipFree(old_blocks));
    return H2_OK;
}

int bases_precompute(const uint64_t* d_bases, size_t n, uint32_t digits, hipStream_t stream) {
    if (n == 0) return H2_OK;
    if (digits == 0) digits = table_default_digits(n);
    if (digits < TABLE_MIN_D || digits > TABLE_MAX_D || (size_t)digits * n >= ((size_t)1 << 31)) {
        set_last_error("h2 bases_precompute: digits must be 11..32 and digits * n < 2^31 (table rows are indexed with 31 bits)");
        return H2_ERR_INVALID;
    }
    bases_forget(d_bases);
    ShiftTable t{};
    t.n = n;
    t.D = digits;
    const uint32_t T = 255, base = T / digits, rem = T % digits;  // the balanced cut of msm_shape
    t.c = rem ? base + 1 : base;
    t.wfull = rem ? rem : digits;
    Affine *table = nullptr, *blocks = nullptr;
    const size_t nblocks = n / BLOCK_ROWS;
    H2_HIP(hipMalloc(&table, (size_t)digits * n * sizeof(Affine)));
    if (nblocks && hipMalloc(&blocks, nblocks * sizeof(Affine)) != hipSuccess) blocks = nullptr;  // optional
    hipLaunchKernelGGL(k_table_build, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const Affine*)d_bases, n, n,
                       t.D, t.c, t.wfull, table);
    if (blocks)
        hipLaunchKernelGGL(k_block_sums, dim3((unsigned)((nblocks + 63) / 64)), dim3(64), 0, stream, (const Affine*)d_bases,
                           nblocks, blocks);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) {
        (void)hipFree(table);
        if (blocks) (void)hipFree(blocks);
        H2_HIP(e);
    }
    t.table = table;
    t.blocks = blocks;
    std::lock_guard<std::mutex> g(g_tab_mu);
    g_tables[d_bases] = t;
    return H2_OK;
}

// The window sums (one point per window, a few KB) go back to the host through a store kernel into mapped pinned
// memory rather than hipMemcpyAsync: a DMA-engine copy queues behind whatever bulk transfer is in flight (the prover
// uploads the next witness column while it commits the current one) and would stall the MSM for the whole transfer.
__global__ void __launch_bounds__(256) k_export(const uint4* src, uint4* dst, size_t count16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count16) dst[i] = src[i];
}

static void export_to_host(const XYZZ* d_src, XYZZ* h_dst, size_t count, hipStream_t stream) {
    const size_t count16 = count * sizeof(XYZZ) / 16;
    hipLaunchKernelGGL(k_export, dim3((unsigned)((count16 + 255) / 256)), dim3(256), 0, stream, (const uint4*)d_src,
                       (uint4*)h_dst, count16);
    H2_HIP(hipGetLastError());
}

// ---------------------------------------------------------------- dominant-scalar detection
static constexpr uint32_t HOT_SAMPLES = 64;
static constexpr uint32_t HOT_MIN = 16;  // a value on >= 16 of 64 sampled rows gets its own window

__global__ void __launch_bounds__(HOT_SAMPLES) k_sample(const Fr* scalars, size_t n, Fr* out) {
    size_t idx = ((size_t)threadIdx.x * 0x9E3779B1ull + 0x7F4A7C15ull) % n;
    fp_store(out + threadIdx.x, fp_load(scalars + idx));
}

// A table is keyed by the ADDRESS of its bases: freed and re-allocated device memory can come back at the same address
// with other points in it.  Every MSM that is about to use a table therefore compares 64 sampled base rows with level 0
// of the table (a copy of the bases as they were) next to the scalar sampling -- same stream, same synchronisation -- and
// a table that no longer matches is dropped instead of producing a wrong commitment.
__global__ void __launch_bounds__(HOT_SAMPLES) k_table_check(const Affine* bases, const Affine* level0, size_t n, uint32_t* stale) {
    const size_t idx = ((size_t)threadIdx.x * 0x9E3779B1ull + 0x7F4A7C15ull) % n;
    const uint4* a = (const uint4*)(bases + idx);
    const uint4* b = (const uint4*)(level0 + idx);
    bool differ = false;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint4 x = a[k], y = b[k];
        differ |= x.x != y.x || x.y != y.y || x.z != y.z || x.w != y.w;
    }
    if (differ) *stale = 1u;
}

struct Hot {
    bool on = false;
    Fr value{};         // Montgomery form, as stored in the column
    uint32_t live = HOT_SAMPLES;  // sampled rows that will cost digits: neither zero nor (when `on`) the dominant value
    const Affine* block_sums = nullptr;  // sums of the BLOCK_ROWS-row blocks of this MSM's bases, when they are tabulated
};

static Hot detect_hot(const Fr* samples) {
    Hot h;
    uint32_t best = 0;
    for (uint32_t i = 0; i < HOT_SAMPLES; i++) {
        uint32_t cnt = 0;
        for (uint32_t j = 0; j < HOT_SAMPLES; j++) cnt += fp_eq(samples[i], samples[j]) ? 1u : 0u;
        if (cnt > best && !fp_is_zero(samples[i])) {  // zero scalars are free already
            best = cnt;
            h.value = samples[i];
        }
    }
    h.on = best >= HOT_MIN;
    if (const char* env = getenv("H2_MSM_NO_HOT"))
        if (env[0] == '1') h.on = false;
    h.live = 0;
    for (uint32_t i = 0; i < HOT_SAMPLES; i++)
        if (!fp_is_zero(samples[i]) && !(h.on && fp_eq(samples[i], h.value))) h.live++;
    return h;
}

// Does the shifted-base table beat the windowed shape for a column of which about live / 64 of the n rows carry digits?
// additions: digits x rows, + ~10 additions' worth of k_finish / k_reduce per bucket (measured: 0.8 - 0.9 ns per bucket
// against 0.08 ns per addition in both shapes).  A table built for n rows has 2^21 buckets at 2^24: a column that is 7/8
// one value (a grand product over padding rows) leaves 12 entries per bucket and is better off in 15 windows of 2^16.
static bool table_pays(const ShiftTable& t, size_t n, uint32_t max_bits, const Hot& hot) {
    const MsmShape p = msm_shape(n, max_bits, false);
    const double rows = (double)n * (double)(hot.live ? hot.live : 1) / (double)HOT_SAMPLES;
    const double plain = (double)p.W * rows + 10.0 * (double)p.Wt * (double)p.nb;
    const double table = (double)table_digits_used(t, max_bits) * rows + 10.0 * (double)(1u << (t.c - 1));
    static const bool force = getenv("H2_MSM_TABLE_FORCE") != nullptr;
    return force || table < plain;
}

// ---------------------------------------------------------------- drivers
void msm_identity(uint64_t out_xyz[12]) {
    Jacobian j = xyzz_to_jacobian(xyzz_identity());
    memcpy(out_xyz, &j, 96);
}

// `fused`: non-null for the multi-column shape (s.cols columns): device table of scalar pointers, then of dominant
// values, and the bit mask of the columns that have one
struct FusedCols {
    const Fr* const* scalars;
    const Fr* hot_values;
    uint64_t hot_mask;
};

static void msm_launch(const MsmShape& s, const Hot& hot, const Fr* d_scalars, const Affine* d_bases, uint32_t max_bits,
                       char* scratch, hipStream_t stream, const FusedCols* fused = nullptr) {
    uint32_t* keys = (uint32_t*)(scratch + s.off_keys);
    uint32_t* sorted = (uint32_t*)(scratch + s.off_sorted);
    uint2* tmp = (uint2*)(scratch + s.off_tmp);
    uint32_t* pcount = (uint32_t*)(scratch + s.off_pcount);
    uint32_t* pbase = (uint32_t*)(scratch + s.off_pbase);
    uint32_t* pcursor = (uint32_t*)(scratch + s.off_pcursor);
    uint32_t* starts = (uint32_t*)(scratch + s.off_starts);
    uint32_t* heavy = (uint32_t*)(scratch + s.off_heavy);  // [0] = count, [1..] = list
    XYZZ* partials = (XYZZ*)(scratch + s.off_partials);
    XYZZ* buckets = (XYZZ*)(scratch + s.off_buckets);
    XYZZ* winpart = (XYZZ*)(scratch + s.off_winpart);

    H2_HIP(hipMemsetAsync(pcount, 0, ((size_t)s.np + 2) * 4, stream));
    H2_HIP(hipMemsetAsync(heavy, 0, 4, stream));
    uint8_t* bflags = nullptr;  // per 256-row block: entered the dominant-value bucket as one point (block sums tabulated)
    if (!fused && hot.on && hot.block_sums) {
        bflags = (uint8_t*)(scratch + s.off_bflags);
        H2_HIP(hipMemsetAsync(bflags, 0, s.n / 256 + 4, stream));
    }
    unsigned nblk = (unsigned)((s.n + 255) / 256);
    unsigned dblk = nblk < 1024 ? nblk : 1024;  // grid-stride: one LDS histogram flush per workgroup
    if (fused) {
        const uint32_t np_col = s.Wc << s.hi_bits;
        hipLaunchKernelGGL(k_digits, dim3(dblk, s.cols), dim3(256), (size_t)np_col * 4, stream, (const Fr*)nullptr, s.n, s.c,
                           s.W, s.nb, max_bits > 254 ? 254u : max_bits, s.lo_bits, s.hi_bits, np_col, keys, pcount, 0,
                           hot.value, fused->scalars, fused->hot_values, fused->hot_mask, 31u, 1u, s.wfull, 0u, (size_t)0,
                           (uint8_t*)nullptr);
    } else {
        hipLaunchKernelGGL(k_digits, dim3(dblk), dim3(256), (size_t)s.np * 4, stream, d_scalars, s.n, s.c, s.W, s.nb,
                           max_bits > 254 ? 254u : max_bits, s.lo_bits, s.hi_bits, s.np, keys, pcount, hot.on ? 1 : 0,
                           hot.value, (const Fr* const*)nullptr, (const Fr*)nullptr, (uint64_t)0, s.range_shift, s.R,
                           s.wfull, s.tab, bflags ? s.n / BLOCK_ROWS * BLOCK_ROWS : (size_t)0, bflags);
    }
    hipLaunchKernelGGL(k_scan_parts, dim3(1), dim3(256), 0, stream, pcount, s.np, pbase, pcursor, starts, s.nbt);
    if (s.tab && s.two_level) {
        uint2* tmpa = (uint2*)(scratch + s.off_tmpa);
        uint32_t* cursor_a = (uint32_t*)(scratch + s.off_tmpa + s.entries * 8);
        uint32_t* tile_start = cursor_a + PARTA_GROUPS_MAX;
        const uint32_t a_bits = s.a_bits, b_bits = s.hi_bits - s.a_bits;
        static bool lds_raised[64] = {};   // per device: 64 KiB of staging + the static tables exceed the default dynamic limit
        int dev = 0;
        H2_HIP(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64 || !lds_raised[dev]) {
            H2_HIP(hipFuncSetAttribute((const void*)k_part_a, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
            H2_HIP(hipFuncSetAttribute((const void*)k_part_b, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
            if (dev >= 0 && dev < 64) lds_raised[dev] = true;
        }
        hipLaunchKernelGGL(k_part_init, dim3(1), dim3(PARTA_GROUPS_MAX), 0, stream, pbase, a_bits, b_bits, cursor_a, tile_start);
        hipLaunchKernelGGL(k_part_a, dim3((unsigned)((s.n + PART_AB_T - 1) / PART_AB_T), s.W), dim3(256), (size_t)PART_AB_T * 8,
                           stream, keys, s.n, s.lo_bits, b_bits, a_bits, cursor_a, tmpa, (uint32_t)s.tab_stride,
                           (const uint8_t*)bflags);
        if (s.Wk > s.W)  // the dominant-scalar array: one partition behind the 2^hi_bits of window 0, appended as before
            hipLaunchKernelGGL(k_partition<PART_T_TABLE>, dim3((unsigned)((s.n + PART_T_TABLE - 1) / PART_T_TABLE), s.Wk - s.W),
                               dim3(256), ((size_t)4 << s.hi_bits) + 4 * (PART_T_TABLE / 256), stream, keys, s.n, s.lo_bits,
                               s.hi_bits, pcursor, tmp, s.range_shift, s.W, s.R, (uint32_t)s.tab_stride, (const uint8_t*)bflags,
                               s.W);
        hipLaunchKernelGGL(k_part_b, dim3((unsigned)(s.entries / PART_AB_T + (1u << a_bits) + 1)), dim3(256), (size_t)PART_AB_T * 8,
                           stream, (const uint2*)tmpa, pbase, s.lo_bits, b_bits, a_bits, tile_start, pcursor, tmp);
    } else if (s.tab && s.hi_bits > 10)
        hipLaunchKernelGGL(k_partition<PART_T_TABLE>, dim3((unsigned)((s.n + PART_T_TABLE - 1) / PART_T_TABLE), s.Wk), dim3(256),
                           ((size_t)4 << s.hi_bits) + 4 * (PART_T_TABLE / 256), stream, keys, s.n, s.lo_bits, s.hi_bits, pcursor, tmp, s.range_shift, s.W,
                           s.R, (uint32_t)s.tab_stride, (const uint8_t*)bflags, 0u);
    else
        hipLaunchKernelGGL(k_partition<PART_T>, dim3((unsigned)((s.n + PART_T - 1) / PART_T), s.Wk), dim3(256),
                           ((size_t)4 << s.hi_bits) + 4 * (PART_T / 256), stream, keys, s.n, s.lo_bits, s.hi_bits, pcursor, tmp, s.range_shift, s.W,
                           s.R, (uint32_t)s.tab_stride, (const uint8_t*)bflags, 0u);
    // a partition holding more than 4x its fair share (and at least a few thousand entries) takes the skew path
    uint32_t skew_threshold = (uint32_t)std::max<size_t>(4 * (s.entries / s.np), 4096);
    const uint32_t hot_partition = (!fused && hot.on) ? ((s.tab ? 1u : s.R * s.W) << s.hi_bits) : 0xffffffffu;
    // partitions of a large MSM over a table are all about the same size (24-29 000 entries at 2^21 .. 2^24): with room
    // for one of them in LDS the sorted list is written as whole lines (H2_MSM_SORT_STAGE=0: placed directly)
    static const bool sort_stage = !(getenv("H2_MSM_SORT_STAGE") && atoi(getenv("H2_MSM_SORT_STAGE")) == 0);
    const size_t avg_part = s.entries / s.np;
    const uint32_t stage_cap = (sort_stage && s.tab && avg_part >= 8192 && avg_part + avg_part / 10 <= 32768) ? 32768u : 0u;
    if (stage_cap) {
        static bool raised[64] = {};
        int dev = 0;
        H2_HIP(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64 || !raised[dev]) {
            H2_HIP(hipFuncSetAttribute((const void*)k_bucket_sort, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
            if (dev >= 0 && dev < 64) raised[dev] = true;
        }
    }
    hipLaunchKernelGGL(k_bucket_sort, dim3(s.np), dim3(SORT_T), ((size_t)4 << s.lo_bits) + (size_t)stage_cap * 4, stream, tmp, pbase,
                       s.lo_bits, s.hi_bits, s.nb, skew_threshold, hot_partition, fused ? s.Wc : 0u,
                       fused ? fused->hot_mask : 0ull, starts, sorted, stage_cap);
    if (!fused && hot.on) hipLaunchKernelGGL(k_copy_hot, dim3(2048), dim3(256), 0, stream, tmp, pbase, hot_partition, sorted);
    if (fused)
        for (uint32_t col = 0; col < s.cols; col++)
            if ((fused->hot_mask >> col) & 1)
                hipLaunchKernelGGL(k_copy_hot, dim3(1024), dim3(256), 0, stream, tmp, pbase,
                                   (col * s.Wc + s.W) << s.hi_bits, sorted);
    unsigned nslices = (unsigned)(((s.entries + (1u << s.log_s) - 1) >> s.log_s));
    // waves per SIMD the accumulation is compiled for: 4 (110 VGPRs, no spills) or 5 (96 VGPRs, 14 spilled): H2_MSM_ACC_WAVES
    static const int acc_waves = getenv("H2_MSM_ACC_WAVES") ? atoi(getenv("H2_MSM_ACC_WAVES")) : 4;
    if (acc_waves == 5)
        hipLaunchKernelGGL(k_acc_slice<5>, dim3((nslices + 255) / 256), dim3(256), 0, stream, d_bases,
                           (!fused && hot.on) ? hot.block_sums : (const Affine*)nullptr, sorted, starts, s.nbt, s.log_s, partials);
    else
        hipLaunchKernelGGL(k_acc_slice<4>, dim3((nslices + 255) / 256), dim3(256), 0, stream, d_bases,
                           (!fused && hot.on) ? hot.block_sums : (const Affine*)nullptr, sorted, starts, s.nbt, s.log_s, partials);
    // one lane per bucket from 2^19 buckets on (round 3, same box: 2^22 windowed 6.75 -> 6.61 ms, over a table 5.91 -> 5.76-5.90,
    // 2^24 windowed 23.9 -> 23.7; at 2^17 the forms tie, at 2^16 buckets the quads win by 4-7 %)
    static const uint32_t lane_from = getenv("H2_MSM_FINISH_LANE_LOG") ? 1u << atoi(getenv("H2_MSM_FINISH_LANE_LOG")) : 1u << 19;
    if (s.nbt >= lane_from)
        hipLaunchKernelGGL(k_finish_lane, dim3((s.nbt + 255) / 256), dim3(256), 0, stream, partials, starts, s.nbt, s.log_s,
                           buckets, heavy + 1, heavy);
    else
        hipLaunchKernelGGL(k_finish, dim3((s.nbt + 63) / 64), dim3(256), 0, stream, partials, starts, s.nbt, s.log_s,
                           buckets, heavy + 1, heavy);
    hipLaunchKernelGGL(k_finish_mid, dim3(512), dim3(256), 0, stream, partials, starts, s.log_s, heavy + 1, heavy, buckets);
    hipLaunchKernelGGL(k_finish_heavy, dim3(64, HEAVY_SPLIT), dim3(256), 0, stream, partials, starts, s.log_s, heavy + 1,
                       heavy);
    hipLaunchKernelGGL(k_finish_heavy2, dim3(256), dim3(4 * HEAVY_SPLIT), 0, stream, partials, starts, s.log_s, heavy + 1,
                       heavy, buckets);
    if (s.RG > 1) H2_HIP(hipMemsetAsync(scratch + s.off_rcount, 0, (size_t)s.Wt * 4, stream));
    if (s.tab && s.planes) {
        XYZZ* R = (XYZZ*)(scratch + s.off_planes);
        XYZZ* part = R + s.chunks;
        uint32_t* counters = (uint32_t*)(part + (size_t)s.planes * s.plane_seg);
        XYZZ* group_acc = winpart + s.Wt + s.planes;
        const uint32_t groups = (s.chunks + REDUCE_T / 4 - 1) / (REDUCE_T / 4);
        H2_HIP(hipMemsetAsync(counters, 0, (size_t)s.planes * 4, stream));
        hipLaunchKernelGGL(k_reduce_chunks, dim3(groups), dim3(REDUCE_T), 0, stream, buckets, s.nb, s.qm, s.chunks, R, group_acc);
        hipLaunchKernelGGL(k_reduce_planes, dim3(s.plane_seg, s.planes), dim3(REDUCE_T), 0, stream, (const XYZZ*)R, s.chunks,
                           s.planes - 1, (const XYZZ*)group_acc, groups, s.plane_l, part, counters, winpart + s.Wt);
        if (s.Wt > 1)
            hipLaunchKernelGGL(k_reduce, dim3(1, 1), dim3(REDUCE_T), 0, stream, buckets + s.nb, 1u, 1u, s.qm,
                               winpart + s.Wt, winpart + 1, (uint32_t*)(scratch + s.off_rcount));
    } else if (s.tab) {
        // one window of nb buckets; the dominant-scalar window only ever fills its bucket 0
        hipLaunchKernelGGL(k_reduce, dim3(s.RG, 1), dim3(REDUCE_T), 0, stream, buckets, s.nb, s.RG, s.qm, winpart + s.Wt,
                           winpart, (uint32_t*)(scratch + s.off_rcount));
        if (s.Wt > 1)
            hipLaunchKernelGGL(k_reduce, dim3(1, 1), dim3(REDUCE_T), 0, stream, buckets + s.nb, 1u, 1u, s.qm,
                               winpart + s.Wt, winpart + 1, (uint32_t*)(scratch + s.off_rcount));
    } else {
        hipLaunchKernelGGL(k_reduce, dim3(s.RG, s.Wt), dim3(REDUCE_T), 0, stream, buckets, s.nb, s.RG, s.qm, winpart + s.Wt,
                           winpart, (uint32_t*)(scratch + s.off_rcount));
    }
    H2_HIP(hipGetLastError());
}

// host tail: add the G partials of each window, then Horner over the windows
static void msm_host_tail(const MsmShape& s, const Hot& hot, const std::vector<XYZZ>& winpart, uint64_t out_xyz[12],
                          size_t w0 = 0) {  // w0: first window of the column inside a fused shape
    XYZZ acc = xyzz_identity();
    if (s.tab && s.planes) {
        // the window's sum from its bit planes: qm * sum_p 2^p P_p + (the chunks' own weighted sums)
        const XYZZ* P = winpart.data() + w0 + s.Wt;
        for (int p = (int)s.planes - 2; p >= 0; p--) {
            acc = xyzz_double(acc);
            acc = xyzz_add(acc, P[p]);
        }
        for (uint32_t m = s.qm; m > 1; m >>= 1) acc = xyzz_double(acc);
        acc = xyzz_add(acc, P[s.planes - 1]);
    } else if (s.tab) {
        acc = winpart[w0];  // the digits' weights are in the table's points: one window, no Horner
    }
    for (int w = (int)(s.tab ? 0 : s.W) - 1; w >= 0; w--) {
        const uint32_t cw = s.c - ((uint32_t)w >= s.wfull ? 1u : 0u);  // the windows above w sit 2^cw higher
        for (uint32_t k = 0; k < cw; k++) acc = xyzz_double(acc);
        XYZZ ws = xyzz_identity();
        for (uint32_t r = 0; r < s.R; r++) ws = xyzz_add(ws, winpart[w0 + (size_t)r * s.W + w]);  // the row ranges of window w
        acc = xyzz_add(acc, ws);
    }
    if (hot.on) {  // + v * E, E = the extra window's sum (bucket 0 carries weight 1)
        XYZZ e = xyzz_identity();
        e = winpart[w0 + (s.tab ? 1u : (size_t)s.R * s.W)];
        const Fr v = fp_from_mont(hot.value);
        XYZZ r = xyzz_identity();
        for (int bit = 253; bit >= 0; bit--) {
            r = xyzz_double(r);
            if ((v.l[bit >> 5] >> (bit & 31)) & 1) r = xyzz_add(r, e);
        }
        acc = xyzz_add(acc, r);
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
    if (n > 0x7fffffffu || msm_shape(n, max_bits, true).entries >= ((size_t)1 << 32)) {
        set_last_error("h2 msm: n * windows must be < 2^32 (sorted entries are indexed with 32 bits): split the MSM");
        return H2_ERR_INVALID;
    }
    ShiftTable tab{}, found{};
    bool use_tab = table_lookup(d_bases, n, max_bits, &tab);
    bool have_tab = use_tab;  // a table of these bases exists (its block sums serve the windowed form too)
    if (use_tab)
        found = tab;
    else
        have_tab = table_find(d_bases, n, &found);
    if (!d_scratch || scratch_bytes < msm_scratch_bytes(n, max_bits)) {
        set_last_error("h2 msm: scratch too small (see h2_msm_scratch_bytes)");
        return H2_ERR_INVALID;
    }
    static thread_local PinnedBuf staging;  // per calling thread: this entry point takes no context lock
    // 64 sampled scalars decide whether a dominant value gets its own window
    Fr* h_samples = (Fr*)staging.get((HOT_SAMPLES + 1) * sizeof(Fr));
    uint32_t* h_stale = (uint32_t*)(h_samples + HOT_SAMPLES);
    *h_stale = 0;
    hipLaunchKernelGGL(k_sample, dim3(1), dim3(HOT_SAMPLES), 0, stream, d_scalars, n, (Fr*)h_samples);
    if (have_tab)
        hipLaunchKernelGGL(k_table_check, dim3(1), dim3(HOT_SAMPLES), 0, stream, (const Affine*)d_bases, found.table, n, h_stale);
    H2_HIP(hipStreamSynchronize(stream));
    if (have_tab && *h_stale) {
        table_drop_containing(d_bases);
        use_tab = have_tab = false;
    }
    Hot hot = detect_hot(h_samples);
    // the extra window trades W additions per dominant row for one: with one or two windows there is nothing to gain,
    // only a giant bucket to fold and a 254-bit multiplication on the host
    if (hot.on && msm_shape(n, max_bits, false).W <= 2) hot.on = false;
    use_tab = use_tab && table_pays(tab, n, max_bits, hot);
    if (hot.on && have_tab && found.blocks && (use_tab ? (size_t)tab.D * tab.n : n) < BLOCK_INDEX0) hot.block_sums = found.blocks;
    MsmShape s = use_tab ? msm_shape_table(n, max_bits, hot.on, tab) : msm_shape(n, max_bits, hot.on);
    if (s.total > scratch_bytes) {   // (cannot happen while h2_msm_scratch_bytes covers every shape; never run past the buffer)
        set_last_error("h2 msm: internal: the chosen shape needs more scratch than h2_msm_scratch_bytes reported");
        return H2_ERR_INVALID;
    }
    msm_launch(s, hot, d_scalars, use_tab ? tab.table : (const Affine*)d_bases, max_bits, (char*)d_scratch, stream);
    const size_t wp = (size_t)s.Wt * s.G + s.planes;
    XYZZ* h_win = (XYZZ*)staging.get(wp * sizeof(XYZZ));
    export_to_host((const XYZZ*)((char*)d_scratch + s.off_winpart), h_win, wp, stream);
    H2_HIP(hipStreamSynchronize(stream));
    std::vector<XYZZ> winpart(h_win, h_win + wp);
    msm_host_tail(s, hot, winpart, out_xyz);
    return H2_OK;
}

// `cols` MSMs over the SAME bases and with the same scalar bound as ONE pass of the pipeline: column j's windows are
// windows j * (W + 1) ... of a single wide MSM (k_digits reads each column's scalars, everything after it only sees
// more windows), so the fixed costs -- sampling, sort launches, the latency-bound finish / reduce, the synchronisation
// and the read-back -- are paid once for the group instead of once per column.  This is what a witness with many
// narrow columns needs: at 2^18 a single MSM is ~1 ms of mostly fixed latency.
static int msm_device_fused(DeviceCtx* ctx, const Fr* const* d_scalars, uint32_t cols, const uint64_t* d_bases, size_t n,
                            uint32_t bits, char* scratch, uint64_t* const* outs, hipStream_t stream) {
    const MsmShape s = msm_shape(n, bits, false, cols);
    const size_t wp = (size_t)s.Wt * s.G + s.planes;
    // the column table (cols pointers, padded to 16 bytes, then cols dominant values) is staged in pinned memory too
    const size_t tab_ptr_bytes = ((size_t)cols * 8 + 15) & ~(size_t)15, tab_bytes = tab_ptr_bytes + (size_t)cols * sizeof(Fr);
    char* pinned = (char*)ctx->pinned.get((size_t)cols * HOT_SAMPLES * sizeof(Fr) + wp * sizeof(XYZZ) + tab_bytes);
    Fr* h_samples = (Fr*)pinned;
    XYZZ* h_win = (XYZZ*)(pinned + (size_t)cols * HOT_SAMPLES * sizeof(Fr));
    char* h_tab = pinned + (size_t)cols * HOT_SAMPLES * sizeof(Fr) + wp * sizeof(XYZZ);
    for (uint32_t j = 0; j < cols; j++)
        hipLaunchKernelGGL(k_sample, dim3(1), dim3(HOT_SAMPLES), 0, stream, d_scalars[j], n, h_samples + (size_t)j * HOT_SAMPLES);
    H2_HIP(hipStreamSynchronize(stream));
    std::vector<Hot> hots(cols);
    FusedCols fc{};
    memset(h_tab, 0, tab_bytes);
    for (uint32_t j = 0; j < cols; j++) {
        hots[j] = detect_hot(h_samples + (size_t)j * HOT_SAMPLES);
        if (s.Wc == s.W) hots[j].on = false;  // one or two windows: no dominant-scalar slot
        if (hots[j].on) fc.hot_mask |= 1ull << j;
        ((uint64_t*)h_tab)[j] = (uint64_t)(uintptr_t)d_scalars[j];
        ((Fr*)(h_tab + tab_ptr_bytes))[j] = hots[j].value;
    }
    // ... and goes up through the store kernel, not hipMemcpyAsync: a DMA-engine copy queues behind the bulk transfers in
    // flight (see k_export) -- the first group of a wide witness waited here until the LAST column had crossed PCIe
    // (k = 22, 64 compact columns: 34 ms of a 124 ms advice phase; tools/experiments/busy.sh showed the GPU idle)
    char* d_tab = scratch + s.off_coltab;
    hipLaunchKernelGGL(k_export, dim3((unsigned)((tab_bytes / 16 + 255) / 256)), dim3(256), 0, stream, (const uint4*)h_tab,
                       (uint4*)d_tab, tab_bytes / 16);
    fc.scalars = (const Fr* const*)d_tab;
    fc.hot_values = (const Fr*)(d_tab + tab_ptr_bytes);
    msm_launch(s, Hot{}, nullptr, (const Affine*)d_bases, bits, scratch, stream, &fc);
    export_to_host((const XYZZ*)(scratch + s.off_winpart), h_win, wp, stream);
    H2_HIP(hipStreamSynchronize(stream));  // (the pinned block is this context's until here)
    std::vector<XYZZ> winpart(h_win, h_win + wp);
    for (uint32_t j = 0; j < cols; j++) msm_host_tail(s, hots[j], winpart, outs[j], (size_t)j * s.Wc);
    return H2_OK;
}

// largest fused group (<= 64 columns) whose sort still fits the LDS histograms and whose keys stay below 2^28 entries
static uint32_t fused_group_limit(size_t n, uint32_t bits) {
    const MsmShape one = msm_shape(n, bits, true);
    // Fusing pays while a column is dominated by fixed latencies (measured, 8 uniform columns: 2^14 0.26 vs 0.51 ms per
    // MSM, 2^16 0.40 vs 0.69, 2^18 0.90 vs 0.86, 2^20 2.35 vs 2.0); past ~4M (scalar, window) entries per column the
    // two-stream pipeline, which hides every column's host tail under the next column's kernels, is the better shape.
    // Columns with a short scalar bound have one or two windows: the slot every fused column keeps for its
    // dominant-scalar window would double their finish / reduce work -- they stay in the pipeline too.
    uint32_t fuse_log = 22;
    if (const char* env = getenv("H2_MSM_FUSE_LOG")) {
        int v = atoi(env);
        if (v >= 10 && v <= 28) fuse_log = (uint32_t)v;
    }
    // Short bounds (a few windows): fused only when a column still spreads over >= 16 sort partitions; a column of a few
    // hundred buckets is one partition and nothing but heavy buckets -- the row ranges of the single-column shape are
    // what it needs
    const uint32_t wc = one.W + (one.W <= 2 ? 0u : 1u);
    const uint32_t lo0 = (one.c - 1 < 8) ? (one.c - 1) : 8;
    // (columns of one or two windows -- 16-bit witness cells -- are dominated by the fixed costs of their sort, finish and
    // reduce four times further up: measured at 2^22 rows, 64 such columns, 1.45 -> 1.25 ms per column fused; wide k = 22
    // from a compact witness 384 -> 372 ms)
    const uint32_t fuse_log_w = (one.W <= 2 && !getenv("H2_MSM_FUSE_LOG")) ? fuse_log + 2 : fuse_log;
    if ((size_t)(one.W + 1) * n > ((size_t)1 << fuse_log_w) || (one.W < 8 && (one.W << (one.c - 1 - lo0)) < 16)) return 1;
    uint32_t best = 1;
    for (uint32_t g = 2; g <= 64; g++) {
        const size_t wt = (size_t)g * wc;
        if (wt * n > ((size_t)1 << 28)) break;
        if ((wt << (one.c - 1 - 9)) > 16384 && one.c - 1 > 9) break;  // lo_bits would exceed 9 (512 bins per partition)
        best = g;
    }
    return best;
}

// Batch of MSMs over ONE set of bases (the prover commits every advice / fixed / z column against the
// same g_lagrange: plonk/prover.rs:293-299, :477-487, keygen.rs:288-291).  Consecutive MSMs alternate
// between two streams with their own scratch halves, so the latency-bound tail of one MSM (k_finish,
// k_reduce: about one wave per SIMD) overlaps the ALU-bound hot loop of the next, and the host-side
// window combine of MSM i runs while the GPU works on i+1, i+2.
int msm_device_batch(DeviceCtx* ctx, const Fr* const* d_scalars, size_t count, const uint64_t* d_bases, size_t n,
                     uint32_t max_bits, void* d_scratch, size_t scratch_bytes, uint64_t* out_xyz, hipStream_t stream) {
    return msm_device_batch_ex(ctx, d_scalars, nullptr, nullptr, count, d_bases, n, max_bits, d_scratch, scratch_bytes,
                               out_xyz, stream);
}

// bases_each / bits_each (either may be null): per-column base table and scalar bound -- the prover's batches mix
// g_lagrange with g (the vanishing argument's random polynomial) and 16-bit witness columns with full-width ones
int msm_device_batch_ex(DeviceCtx* ctx, const Fr* const* d_scalars, const uint64_t* const* bases_each,
                        const uint32_t* bits_each, size_t count, const uint64_t* d_bases, size_t n, uint32_t max_bits,
                        void* d_scratch, size_t scratch_bytes, uint64_t* out_xyz, hipStream_t stream) {
    if (count == 0) return H2_OK;
    uint32_t top_bits = 0;
    for (size_t i = 0; i < count; i++) top_bits = std::max(top_bits, bits_each ? bits_each[i] : max_bits);
    if (n > 0x7fffffffu || (top_bits && msm_shape(n, top_bits, true).entries >= ((size_t)1 << 32))) {
        set_last_error("h2 msm: n * windows must be < 2^32 (sorted entries are indexed with 32 bits): split the MSM");
        return H2_ERR_INVALID;
    }
    if (n == 0 || top_bits == 0) {
        for (size_t i = 0; i < count; i++) msm_identity(out_xyz + 12 * i);
        return H2_OK;
    }
    // columns that share their base table and their bound are committed as fused groups when the caller's scratch
    // allows it (h2_msm_batch_scratch_bytes); the rest goes through the two-stream pipeline below
    std::vector<char> done_fused(count, 0);
    // columns whose bases have a shifted-base table go through the pipeline in table form
    std::vector<ShiftTable> tabs(count), founds(count);
    std::vector<char> use_tab(count, 0), have_tab(count, 0);
    for (size_t i = 0; i < count; i++) {
        const uint32_t bits = bits_each ? bits_each[i] : max_bits;
        const uint64_t* bases = bases_each && bases_each[i] ? bases_each[i] : d_bases;
        if (!bits || !bases) continue;
        use_tab[i] = table_lookup(bases, n, bits, &tabs[i]) ? 1 : 0;
        if (use_tab[i]) {
            founds[i] = tabs[i];
            have_tab[i] = 1;
        } else {
            have_tab[i] = table_find(bases, n, &founds[i]) ? 1 : 0;  // its block sums serve the windowed form too
        }
    }
    if (getenv("H2_MSM_NO_FUSE") == nullptr) {
        H2_HIP(hipStreamSynchronize(stream));
        for (size_t i = 0; i < count; i++) {
            if (done_fused[i] || use_tab[i]) continue;
            const uint32_t bits = bits_each ? bits_each[i] : max_bits;
            if (bits == 0) continue;
            const uint64_t* bases = bases_each && bases_each[i] ? bases_each[i] : d_bases;
            std::vector<size_t> members{i};
            for (size_t j = i + 1; j < count; j++) {
                const uint32_t bj = bits_each ? bits_each[j] : max_bits;
                const uint64_t* basej = bases_each && bases_each[j] ? bases_each[j] : d_bases;
                if (!done_fused[j] && !use_tab[j] && bj == bits && basej == bases) members.push_back(j);
            }
            const uint32_t limit = fused_group_limit(n, bits);
            for (size_t m0 = 0; m0 < members.size(); m0 += limit) {
                const uint32_t g = (uint32_t)std::min<size_t>(limit, members.size() - m0);
                if (g < 2) break;
                if (msm_shape(n, bits, false, g).total > scratch_bytes) break;  // caller sized the scratch for the pipeline only
                std::vector<const Fr*> sc(g);
                std::vector<uint64_t*> outs(g);
                for (uint32_t t = 0; t < g; t++) {
                    sc[t] = d_scalars[members[m0 + t]];
                    outs[t] = out_xyz + 12 * members[m0 + t];
                    done_fused[members[m0 + t]] = 1;
                }
                int rc = msm_device_fused(ctx, sc.data(), g, bases, n, bits, (char*)d_scratch, outs.data(), stream);
                if (rc != H2_OK) return rc;
            }
        }
    }
    size_t per = 0, wp_max = 0;
    std::vector<MsmShape> shapes(2 * count), tshapes(2 * count);  // windowed / table form, without / with a dominant scalar
    for (size_t i = 0; i < count; i++) {
        const uint32_t bits = bits_each ? bits_each[i] : max_bits;
        if (done_fused[i]) continue;
        shapes[2 * i] = msm_shape(n, bits ? bits : 1, false);
        shapes[2 * i + 1] = msm_shape(n, bits ? bits : 1, true);
        per = std::max(per, align_up(std::max(shapes[2 * i].total, shapes[2 * i + 1].total), 256));
        wp_max = std::max(wp_max, (size_t)shapes[2 * i + 1].Wt * shapes[2 * i + 1].G + shapes[2 * i + 1].planes);
        if (use_tab[i]) {
            tshapes[2 * i] = msm_shape_table(n, bits, false, tabs[i]);
            tshapes[2 * i + 1] = msm_shape_table(n, bits, true, tabs[i]);
            per = std::max(per, align_up(std::max(tshapes[2 * i].total, tshapes[2 * i + 1].total), 256));
            // (the table form exports its window sum as bit planes: up to 24 points where the windowed form has its W + 1)
            wp_max = std::max(wp_max, (size_t)tshapes[2 * i + 1].Wt * tshapes[2 * i + 1].G + tshapes[2 * i + 1].planes);
        }
    }
    if (!d_scratch || scratch_bytes < 2 * per) {
        set_last_error("h2 msm batch: scratch too small (need 2 x h2_msm_scratch_bytes of the widest column, 256-byte aligned)");
        return H2_ERR_INVALID;
    }
    // pinned staging: the sampled scalars of every column, then the per-MSM window partials (async read-back)
    char* pinned = (char*)ctx->pinned.get(count * (HOT_SAMPLES * sizeof(Fr) + wp_max * sizeof(XYZZ) + 16));
    Fr* h_samples = (Fr*)pinned;
    XYZZ* h_win = (XYZZ*)(pinned + count * HOT_SAMPLES * sizeof(Fr));
    uint32_t* h_stale = (uint32_t*)(pinned + count * (HOT_SAMPLES * sizeof(Fr) + wp_max * sizeof(XYZZ)));  // per column
    for (size_t i = 0; i < count; i++) h_stale[i] = 0;
    // pipeline lanes: one stream + one scratch slice each.  Two: a third and fourth lane (H2_MSM_LANES, when the scratch
    // holds them) were measured and do not pay -- 2^18: 0.78 (2) / 0.76 (3) / 0.84 (4) ms per MSM, 2^20: 2.06 / 2.08 / 2.17
    hipStream_t st[4] = {ctx->stream, ctx->copy_stream, ctx->aux_stream[0], ctx->aux_stream[1]};
    size_t lanes = 2;
    if (const char* env = getenv("H2_MSM_LANES")) {
        int v = atoi(env);
        if (v >= 2 && v <= 4 && per && (size_t)v * per <= scratch_bytes) lanes = (size_t)v;
    }
    for (size_t i = 0; i < count; i++) {
        hipLaunchKernelGGL(k_sample, dim3(1), dim3(HOT_SAMPLES), 0, stream, d_scalars[i], n, h_samples + i * HOT_SAMPLES);
        if (have_tab[i] && !done_fused[i])
            hipLaunchKernelGGL(k_table_check, dim3(1), dim3(HOT_SAMPLES), 0, stream,
                               (const Affine*)(bases_each && bases_each[i] ? bases_each[i] : d_bases), founds[i].table, n, h_stale + i);
    }
    H2_HIP(hipStreamSynchronize(stream));  // inputs produced on the caller's stream are complete; samples are in
    for (size_t i = 0; i < count; i++)
        if (have_tab[i] && h_stale[i]) {  // the bases changed under their table (see k_table_check): windowed form, table dropped
            table_drop_containing(bases_each && bases_each[i] ? bases_each[i] : d_bases);
            use_tab[i] = have_tab[i] = 0;
        }
    std::vector<Hot> hots(count);
    for (size_t i = 0; i < count; i++) {
        hots[i] = detect_hot(h_samples + i * HOT_SAMPLES);
        if (hots[i].on && !done_fused[i] && shapes[2 * i].W <= 2) hots[i].on = false;  // nothing to gain (see msm_device)
        if (use_tab[i] && !done_fused[i]) {
            const uint32_t bits = bits_each ? bits_each[i] : max_bits;
            if (table_pays(tabs[i], n, bits, hots[i])) {
                shapes[2 * i] = tshapes[2 * i];
                shapes[2 * i + 1] = tshapes[2 * i + 1];
            } else {
                use_tab[i] = 0;
            }
        }
        if (hots[i].on && have_tab[i] && !done_fused[i] && founds[i].blocks &&
            (use_tab[i] ? (size_t)tabs[i].D * tabs[i].n : n) < BLOCK_INDEX0)
            hots[i].block_sums = founds[i].blocks;
    }
    std::vector<hipEvent_t> done(count, nullptr);
    size_t lane_of = 0;
    for (size_t i = 0; i < count; i++) {
        const uint32_t bits = bits_each ? bits_each[i] : max_bits;
        if (bits == 0 || done_fused[i]) continue;  // identity (arithmetic.rs:346) / already committed in a fused group
        const MsmShape& s = shapes[2 * i + (hots[i].on ? 1 : 0)];
        const uint64_t* bases = bases_each && bases_each[i] ? bases_each[i] : d_bases;
        char* scratch = (char*)d_scratch + (lane_of % lanes) * per;
        hipStream_t q = st[lane_of % lanes];
        lane_of++;
        msm_launch(s, hots[i], d_scalars[i], use_tab[i] ? tabs[i].table : (const Affine*)bases, bits, scratch, q);
        export_to_host((const XYZZ*)(scratch + s.off_winpart), h_win + i * wp_max, (size_t)s.Wt * s.G + s.planes, q);
        H2_HIP(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
        H2_HIP(hipEventRecord(done[i], q));
    }
    for (size_t i = 0; i < count; i++) {
        if (done_fused[i]) continue;
        if (!done[i]) {
            msm_identity(out_xyz + 12 * i);
            continue;
        }
        const MsmShape& s = shapes[2 * i + (hots[i].on ? 1 : 0)];
        H2_HIP(hipEventSynchronize(done[i]));
        H2_HIP(hipEventDestroy(done[i]));
        std::vector<XYZZ> winpart(h_win + i * wp_max, h_win + i * wp_max + (size_t)s.Wt * s.G + s.planes);
        msm_host_tail(s, hots[i], winpart, out_xyz + 12 * i);
    }
    return H2_OK;
}

size_t msm_batch_scratch_bytes(size_t n, uint32_t max_bits, size_t count) {
    const size_t pipeline = 2 * align_up(msm_scratch_bytes(n, max_bits), 256);
    if (count < 2 || n == 0 || max_bits == 0) return pipeline;
    const uint32_t g = (uint32_t)std::min<size_t>(count, fused_group_limit(n, max_bits));
    return g >= 2 ? std::max(pipeline, msm_shape(n, max_bits, false, g).total) : pipeline;
}

// ---- resident SRS: host ranges the caller promised not to modify (h2_bases_register).  The reference
// re-uploads the bases on every MSM (arithmetic.rs:354-360); at 2^20 that is 64 MiB of PCIe per call,
// about as long as the MSM itself.
namespace {
struct Registration {
    size_t len;    // in units of `unit` bytes: points (64) for an SRS range, field elements (32) for a polynomial
    uint64_t gen;  // bumped by every register / unregister: a device copy made for another generation is stale
    uint32_t unit; // 64: G1Affine points (h2_bases_register); 32: Fr coefficients / values (h2_poly_register: no table)
};
std::mutex g_reg_mu;
std::map<const uint64_t*, Registration> g_registered;  // host pointer -> length, generation, element size
uint64_t g_reg_gen = 0;
}  // namespace

int bases_register(const uint64_t* bases, size_t n) {
    std::lock_guard<std::mutex> g(g_reg_mu);
    g_registered[bases] = Registration{n, ++g_reg_gen, 64u};
    return H2_OK;
}

// h2_poly_register: a host vector of n field elements the caller promises not to modify -- the proving key's fixed / sigma / l_0 /
// l_last coefficient forms (plonk.rs:226-240), read by every proof (plonk/evaluation.rs:1229-1241, prover.rs:731-737) -- so that
// the host-slice entry points that READ vectors find a device copy uploaded once per device instead of crossing PCIe per call.
// Same registry, generations and unregister path as the SRS ranges; no shifted-base table, of course.
int poly_register(const uint64_t* values, size_t n) {
    std::lock_guard<std::mutex> g(g_reg_mu);
    g_registered[values] = Registration{n, ++g_reg_gen, 32u};
    return H2_OK;
}

int bases_unregister(const uint64_t* bases) {
    {
        std::lock_guard<std::mutex> g(g_reg_mu);
        g_registered.erase(bases);
        ++g_reg_gen;
    }
    // free the device copies now, on every device (under the locks of ALL its host-API slots: a host-buffer MSM holds its
    // slot's lock until its result is back, so nothing in flight reads the copy; then the shared map's own lock); a device
    // that never runs another MSM would keep them otherwise
    std::map<DeviceShared*, std::vector<DeviceCtx*>> by_device;
    for (DeviceCtx* ctx : existing_contexts()) by_device[ctx->shared].push_back(ctx);
    for (auto& dv : by_device) {
        std::sort(dv.second.begin(), dv.second.end(), [](DeviceCtx* a, DeviceCtx* b) { return a->slot < b->slot; });
        for (DeviceCtx* ctx : dv.second) ctx->mu.lock();
        {
            std::lock_guard<std::mutex> g(dv.first->mu);
            auto it = dv.first->resident.find((const void*)bases);
            if (it != dv.first->resident.end()) {
                bases_forget((const uint64_t*)it->second.ptr);
                (void)hipFree(it->second.ptr);
                dv.first->resident.erase(it);
            }
            for (ResidentCopy& c : dv.first->retired) {
                bases_forget((const uint64_t*)c.ptr);
                (void)hipFree(c.ptr);
            }
            dv.first->retired.clear();
        }
        for (DeviceCtx* ctx : dv.second) ctx->mu.unlock();
    }
    return H2_OK;
}

// device copy of a registered range that contains [bases, bases + n) -- or nullptr.  A copy is reused only for the
// registration (generation, length) it was uploaded for: unregister + refill + register of the same address uploads
// the new points.
static const void* resident_lookup_unit(DeviceCtx* ctx, const uint64_t* bases, size_t n, uint32_t unit);
static const Affine* resident_lookup(DeviceCtx* ctx, const uint64_t* bases, size_t n) {
    return (const Affine*)resident_lookup_unit(ctx, bases, n, 64u);
}
// the device copy of host Fr values [values, values + n) when they lie inside a range registered with h2_poly_register (uploaded
// on this device's first use, complete when this returns) -- or nullptr: the caller uploads as before.  Call with the slot's lock.
const Fr* poly_resident(DeviceCtx* ctx, const uint64_t* values, size_t n) {
    {
        std::lock_guard<std::mutex> g(g_reg_mu);
        if (g_registered.empty()) return nullptr;     // (the common case of a caller that registers nothing: no device lock taken)
    }
    return (const Fr*)resident_lookup_unit(ctx, values, n, 32u);
}
static const void* resident_lookup_unit(DeviceCtx* ctx, const uint64_t* bases, size_t n, uint32_t unit) {
    const uint64_t* key = nullptr;
    Registration reg{0, 0, 0};
    const size_t words = unit / 8;
    // the device copies are shared by the host-API slots of the device: one slot uploads (and tabulates) an SRS, the other
    // waits here and finds it complete
    std::lock_guard<std::mutex> shared_lock(ctx->shared->mu);
    {
        std::lock_guard<std::mutex> g(g_reg_mu);
        // Device copies whose registration is gone or was replaced are never FREED here: this caller holds its own slot's
        // lock only, and the other host-API slot of the device may be in the middle of an MSM over such a copy.
        // A copy whose registration is gone is about to be freed by the h2_bases_unregister that removed it (under every
        // slot's lock); one whose registration was REPLACED (register again without unregister) steps aside into `retired`,
        // which the next unregister empties the same way.
        for (auto it = ctx->resident.begin(); it != ctx->resident.end();) {
            auto r = g_registered.find((const uint64_t*)it->first);
            if (r != g_registered.end() && (r->second.gen != it->second.gen || r->second.len != it->second.len)) {
                ctx->shared->retired.push_back(it->second);
                it = ctx->resident.erase(it);
            } else {
                ++it;
            }
        }
        auto it = g_registered.upper_bound(bases);
        if (it == g_registered.begin()) return nullptr;
        --it;
        if (it->second.unit != unit) return nullptr;      // points asked of a polynomial's range or the reverse
        if (bases + words * n > it->first + words * it->second.len) return nullptr;
        if ((size_t)(bases - it->first) % words) return nullptr;   // (not on an element boundary of the registered range)
        key = it->first;
        reg = it->second;
    }
    auto rit = ctx->resident.find((const void*)key);
    if (rit != ctx->resident.end() && (rit->second.gen != reg.gen || rit->second.len != reg.len)) return nullptr;  // (cannot happen: swept above)
    if (rit == ctx->resident.end()) {
        ResidentCopy c;
        c.len = reg.len;
        c.gen = reg.gen;
        H2_HIP(hipMalloc(&c.ptr, reg.len * (size_t)unit));
        host_upload(c.ptr, key, reg.len * (size_t)unit, ctx->stream);
        rit = ctx->resident.emplace((const void*)key, c).first;
        // a registered SRS is committed against for the life of the process: give its device copy a shifted-base table
        // when that takes less than half of the free memory (H2_MSM_TABLES=0: never)
        // -- an OPTIONAL optimisation: a failed build (allocation, launch) must not fail the MSM that triggered it; the
        // windowed pipeline answers with the same point
        const char* env = getenv("H2_MSM_TABLES");
        size_t free_b = 0, total_b = 0;
        if (unit == 64u && !(env && env[0] == '0') && reg.len >= ((size_t)1 << 15) && hipMemGetInfo(&free_b, &total_b) == hipSuccess &&
            bases_precompute_bytes(reg.len, 0) < free_b / 2) {
            try {
                if (bases_precompute((const uint64_t*)c.ptr, reg.len, 0, ctx->stream) != H2_OK) (void)hipGetLastError();
            } catch (const HipError&) {
                (void)hipGetLastError();  // clear the sticky error; no table for this copy
            }
        }
        H2_HIP(hipStreamSynchronize(ctx->stream));  // complete before another slot (another stream) can find it
    }
    return (const char*)rit->second.ptr + (size_t)(bases - key) * 8;
}

int msm_host_resident_scalars(DeviceCtx* ctx, const Fr* d_scalars, const uint64_t* bases, size_t n, uint32_t max_bits,
                              uint64_t out_xyz[12]) {
    const Affine* d_bases = resident_lookup(ctx, bases, n);
    if (!d_bases) {
        Affine* up = (Affine*)ctx->buf_c.get(n * sizeof(Affine));
        host_upload(up, bases, n * sizeof(Affine), ctx->stream);
        d_bases = up;
    }
    size_t sb = msm_scratch_bytes(n, max_bits);
    void* scratch = ctx->msm_scratch.get(sb);
    return msm_device(ctx, d_scalars, (const uint64_t*)d_bases, n, max_bits, scratch, sb, out_xyz, ctx->stream);
}

int msm_host(DeviceCtx* ctx, const uint64_t* scalars, const uint64_t* bases, size_t n, uint32_t max_bits,
             uint64_t out_xyz[12]) {
    Fr* d_s = (Fr*)ctx->buf_d.get(n * sizeof(Fr));
    host_upload(d_s, scalars, n * sizeof(Fr), ctx->stream);
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
// Device-side fold of gathered partial points (one proof over several ranks): points[r * count + j] = rank r's partial of
// MSM j (Jacobian, 96 B), out[j] = sum over r in rank order -- the `.reduce(|acc, x| acc + x)` of arithmetic.rs:433-435
// after an all-gather, without bringing world x count points back to the host.  One lane per MSM (count is ~10).
__global__ void __launch_bounds__(64) k_g1_fold(const Jacobian* points, uint32_t world, uint32_t count, Jacobian* out) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    XYZZ acc = xyzz_identity();
    for (uint32_t r = 0; r < world; r++) {
        const Jacobian p = points[(size_t)r * count + j];
        XYZZ q;
        q.x = p.x;
        q.y = p.y;
        q.zz = fp_sqr(p.z);
        q.zzz = fp_mul(q.zz, p.z);
        acc = xyzz_add(acc, q);
    }
    out[j] = xyzz_to_jacobian(acc);
}

int g1_fold_launch(const uint64_t* d_points, uint32_t world, uint32_t count, uint64_t* d_out, hipStream_t stream) {
    if (count == 0) return H2_OK;
    hipLaunchKernelGGL(k_g1_fold, dim3((count + 63) / 64), dim3(64), 0, stream, (const Jacobian*)d_points, world, count,
                       (Jacobian*)d_out);
    H2_HIP(hipGetLastError());
    return H2_OK;
}

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
