// logup.hpp -- multiplicity column of a logup lookup (logup.hip)
#pragma once
#include "common.hpp"

namespace h2 {
size_t logup_scratch_bytes(size_t n);
int logup_multiplicity_launch(const Fr* d_table, const Fr* const* d_inputs, size_t n_inputs, size_t usable, size_t n,
                              Fr* d_m, void* d_scratch, size_t scratch_bytes, hipStream_t stream, uint32_t* max_count_out = nullptr);
int logup_counts_launch(const Fr* d_table, const Fr* const* d_inputs, size_t n_inputs, size_t usable, size_t n, size_t row_begin,
                        size_t row_end, uint32_t* d_counts, void* d_scratch, size_t scratch_bytes, hipStream_t stream);
int logup_emit_launch(const uint32_t* d_counts, size_t usable, size_t n, Fr* d_m, hipStream_t stream);
}  // namespace h2
