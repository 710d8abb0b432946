// logup.hpp -- multiplicity column of a logup lookup (logup.hip)
#pragma once
#include "common.hpp"

namespace h2 {
size_t logup_scratch_bytes(size_t n);
int logup_multiplicity_launch(const Fr* d_table, const Fr* const* d_inputs, size_t n_inputs, size_t usable, size_t n,
                              Fr* d_m, void* d_scratch, size_t scratch_bytes, hipStream_t stream);
}  // namespace h2
