// evalh.hpp -- evaluate_h drivers (evalh.hip)
#pragma once
#include "common.hpp"

namespace h2 {
// column pointers inside `d` are device pointers; the descriptor and its program arrays are host memory
int evalh_device(DeviceCtx* ctx, const h2_evalh_desc* d, Fr* d_values, hipStream_t stream, bool have_lock);
// everything in host memory
int evalh_host(DeviceCtx* ctx, const h2_evalh_desc* d, uint64_t* values);
// host memory, columns as COEFFICIENT vectors of 2^k elements (l_active_row extended): the cuda evaluate_h's shape
int evalh_host_coeffs(DeviceCtx* ctx, const h2_evalh_desc* d, uint64_t* values);
}  // namespace h2
