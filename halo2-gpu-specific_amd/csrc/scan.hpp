// scan.hpp -- prefix scans over Fr (scan.hip)
#pragma once
#include "common.hpp"

namespace h2 {
size_t scan_tmp_elems(size_t n);
int kate_division_launch(const Fr* d_a, size_t n, const uint64_t b[4], Fr* d_q, Fr* d_tmp, hipStream_t stream);
int prefix_product_launch(const Fr* d_f, size_t n, const uint64_t init[4], Fr* d_z, Fr* d_tmp, hipStream_t stream);
int prefix_sum_launch(const Fr* d_f, size_t n, const uint64_t init[4], Fr* d_z, Fr* d_tmp, hipStream_t stream);
}  // namespace h2
