// logup.hip -- the multiplicity column m(X) of a logup lookup (plonk/logup/prover.rs:104-180).
// The reference sorts the compressed table on the CPU, binary-searches every compressed input value (with a
// per-thread BTreeMap cache) and merges per-thread count maps.  Here the first `usable` table rows are inserted
// into an open-addressing hash table of row indices keyed by the 256-bit value (a duplicated table value keeps
// its LOWEST row: atomicMin), every input value probes the table and bumps a 32-bit counter of the row it hits
// (wave-aggregated: padding rows make most lanes hit the same row), and the counters become field elements.
// An input value that is not in the table is an error (the reference panics: "logup binary_search_by_key should hit").
#include "common.hpp"
#include "logup.hpp"

namespace h2 {

static constexpr uint32_t SLOT_EMPTY = 0xffffffffu;

__device__ __forceinline__ uint32_t key_hash(const Fr& k) {
    uint32_t h = 0x9e3779b9u;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        h ^= k.l[i];
        h *= 0x85ebca6bu;
        h ^= h >> 13;
    }
    h *= 0xc2b2ae35u;
    return h ^ (h >> 16);
}

__global__ void __launch_bounds__(256) k_logup_build(const Fr* table, uint32_t usable, uint32_t mask, uint32_t* slots) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= usable) return;
    Fr key = fp_load(table + i);
    // a row that repeats its predecessor is never the lowest row of its value (the padding rows of a table: one load
    // instead of a probe)
    if (i > 0 && fp_eq(fp_load(table + i - 1), key)) return;
    uint32_t h = key_hash(key) & mask;
    for (;;) {
        // A table is padded with one value over most of its rows (a range table of 2^16 entries in 2^20 rows: a million
        // rows of zero), and workgroups start in row order: by the time a later duplicate arrives its value is almost
        // always present under a LOWER row already.  A plain read settles that without an atomic -- a million
        // compare-and-swaps + atomic minima on one address took 23 ms per lookup at 2^20.
        uint32_t cur = __atomic_load_n(&slots[h], __ATOMIC_RELAXED);
        if (cur == SLOT_EMPTY) cur = atomicCAS(&slots[h], SLOT_EMPTY, i);
        if (cur == SLOT_EMPTY) return;                      // claimed an empty slot
        if (fp_eq(fp_load(table + cur), key)) {             // same value already present: keep the lowest row
            if (i < cur) atomicMin(&slots[h], i);
            return;
        }
        h = (h + 1) & mask;
    }
}

// One lane per (input column, row).  count[row hit] += 1, aggregated over the lanes of a wave that hit the row of
// the first active lane (two rounds), the rest individually.
__global__ void __launch_bounds__(256) k_logup_count(const Fr* table, const Fr* input, uint32_t first, uint32_t usable,
                                                     uint32_t mask, const uint32_t* slots, uint32_t* count, uint32_t* miss) {
    uint32_t i = first + blockIdx.x * blockDim.x + threadIdx.x;   // rows [first, usable) of the input column
    bool valid = i < usable;
    uint32_t hit = SLOT_EMPTY;
    if (valid) {
        Fr key = fp_load(input + i);
        uint32_t h = key_hash(key) & mask;
        for (;;) {
            uint32_t cur = slots[h];
            if (cur == SLOT_EMPTY) break;
            if (fp_eq(fp_load(table + cur), key)) {
                hit = cur;
                break;
            }
            h = (h + 1) & mask;
        }
        if (hit == SLOT_EMPTY) {
            atomicAdd(miss, 1u);
            valid = false;
        }
    }
    const uint32_t lane = threadIdx.x & 63;
    uint64_t active = __ballot(valid);
#pragma unroll
    for (int round = 0; round < 2; round++) {
        if (active == 0) break;
        const int leader = __ffsll((unsigned long long)active) - 1;
        const uint32_t k = __shfl(hit, leader, 64);
        const uint64_t same = __ballot(valid && hit == k) & active;
        if ((int)lane == leader) atomicAdd(&count[k], (uint32_t)__popcll(same));
        active &= ~same;
    }
    if ((active >> lane) & 1) atomicAdd(&count[hit], 1u);
}

// `max_out` (nullable): the largest multiplicity, for the caller's scalar bound of m's commitment -- one atomic per wave
__global__ void __launch_bounds__(256) k_logup_emit(const uint32_t* count, uint32_t usable, size_t n, Fr* m, uint32_t* max_out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t c = 0;
    if (i < n) {
        Fr v = fp_zero<FrParams>();
        if (i < usable) {
            c = count[i];
            v.l[0] = c;
            v = fp_to_mont(v);
        }
        fp_store(m + i, v);
    }
    if (max_out) {
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)c, off, 64);
            c = o > c ? o : c;
        }
        if ((threadIdx.x & 63) == 0 && c) atomicMax(max_out, c);
    }
}

static uint32_t table_capacity(size_t usable) {
    uint32_t cap = 64;
    while ((size_t)cap < 2 * usable) cap <<= 1;
    return cap;
}

size_t logup_scratch_bytes(size_t n) { return ((size_t)table_capacity(n) + n + 64) * sizeof(uint32_t); }

int logup_multiplicity_launch(const Fr* d_table, const Fr* const* d_inputs, size_t n_inputs, size_t usable, size_t n,
                              Fr* d_m, void* d_scratch, size_t scratch_bytes, hipStream_t stream, uint32_t* max_count_out) {
    if (usable > n || n >= 0x7fffffffu) {
        set_last_error("h2_dev_logup_multiplicity: bad sizes");
        return H2_ERR_INVALID;
    }
    if (scratch_bytes < logup_scratch_bytes(n)) {
        set_last_error("h2_dev_logup_multiplicity: scratch too small (h2_logup_scratch_bytes)");
        return H2_ERR_INVALID;
    }
    const uint32_t cap = table_capacity(usable), mask = cap - 1;
    uint32_t* slots = (uint32_t*)d_scratch;
    uint32_t* count = slots + cap;
    uint32_t* miss = count + n;
    H2_HIP(hipMemsetAsync(slots, 0xff, (size_t)cap * 4, stream));
    H2_HIP(hipMemsetAsync(count, 0, (n + 2) * 4, stream));   // n counters, the misses, the largest counter
    const unsigned blocks = (unsigned)((usable + 255) / 256);
    if (usable) {
        hipLaunchKernelGGL(k_logup_build, dim3(blocks), dim3(256), 0, stream, d_table, (uint32_t)usable, mask, slots);
        for (size_t j = 0; j < n_inputs; j++)
            hipLaunchKernelGGL(k_logup_count, dim3(blocks), dim3(256), 0, stream, d_table, d_inputs[j], 0u, (uint32_t)usable,
                               mask, slots, count, miss);
    }
    hipLaunchKernelGGL(k_logup_emit, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, count, (uint32_t)usable, n,
                       d_m, max_count_out ? miss + 1 : (uint32_t*)nullptr);
    H2_HIP(hipGetLastError());
    uint32_t h_tail[2] = {0, 0};
    H2_HIP(hipMemcpyAsync(h_tail, miss, 8, hipMemcpyDeviceToHost, stream));
    H2_HIP(hipStreamSynchronize(stream));
    const uint32_t h_miss = h_tail[0];
    if (max_count_out) *max_count_out = h_tail[1];
    if (h_miss) {
        set_last_error("logup: " + std::to_string(h_miss) + " input value(s) are missing from the table");
        return H2_ERR_INVALID;
    }
    return H2_OK;
}

// The same counting restricted to the input rows [row_begin, row_end) -- one rank's share when the rows of a proof are dealt
// over several devices -- with the RAW counters out (n u32, then the number of input values missing from the table): integer
// counts are an RCCL reduction (sum), field elements are not; logup_emit_launch turns the summed counters into m(X).
int logup_counts_launch(const Fr* d_table, const Fr* const* d_inputs, size_t n_inputs, size_t usable, size_t n, size_t row_begin,
                        size_t row_end, uint32_t* d_counts, void* d_scratch, size_t scratch_bytes, hipStream_t stream) {
    if (usable > n || n >= 0x7fffffffu || row_begin > row_end || row_end > n) {
        set_last_error("h2_dev_logup_counts: bad sizes");
        return H2_ERR_INVALID;
    }
    if (scratch_bytes < logup_scratch_bytes(n)) {
        set_last_error("h2_dev_logup_counts: scratch too small (h2_logup_scratch_bytes)");
        return H2_ERR_INVALID;
    }
    const uint32_t cap = table_capacity(usable), mask = cap - 1;
    uint32_t* slots = (uint32_t*)d_scratch;
    H2_HIP(hipMemsetAsync(slots, 0xff, (size_t)cap * 4, stream));
    H2_HIP(hipMemsetAsync(d_counts, 0, (n + 1) * 4, stream));
    const size_t last = row_end < usable ? row_end : usable;
    if (usable) {
        hipLaunchKernelGGL(k_logup_build, dim3((unsigned)((usable + 255) / 256)), dim3(256), 0, stream, d_table, (uint32_t)usable,
                           mask, slots);
        if (row_begin < last)
            for (size_t j = 0; j < n_inputs; j++)
                hipLaunchKernelGGL(k_logup_count, dim3((unsigned)((last - row_begin + 255) / 256)), dim3(256), 0, stream, d_table,
                                   d_inputs[j], (uint32_t)row_begin, (uint32_t)last, mask, slots, d_counts, d_counts + n);
    }
    H2_HIP(hipGetLastError());
    return H2_OK;
}

int logup_emit_launch(const uint32_t* d_counts, size_t usable, size_t n, Fr* d_m, hipStream_t stream) {
    if (usable > n || n >= 0x7fffffffu) {
        set_last_error("h2_dev_logup_emit: bad sizes");
        return H2_ERR_INVALID;
    }
    hipLaunchKernelGGL(k_logup_emit, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_counts, (uint32_t)usable, n, d_m,
                       (uint32_t*)nullptr);
    H2_HIP(hipGetLastError());
    return H2_OK;
}

}  // namespace h2
