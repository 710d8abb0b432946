// poly.hpp -- launchers of the elementwise kernels (poly.hip)
#pragma once
#include "common.hpp"

namespace h2 {
int eval_op_launch(int op, Fr* res, const Fr* l, const Fr* r, int32_t l_rot, int32_t r_rot, size_t size,
                   const uint64_t c[4], hipStream_t stream);
int divide_by_vanishing_launch(Fr* a, size_t size, const Fr* t, size_t t_len, hipStream_t stream);
int batch_mont_launch(Fr* a, size_t n, bool to_mont, hipStream_t stream);
}  // namespace h2
