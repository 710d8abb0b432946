// poly.hpp -- launchers of the elementwise kernels (poly.hip)
#pragma once
#include "common.hpp"

namespace h2 {
int eval_op_launch(int op, Fr* res, const Fr* l, const Fr* r, int32_t l_rot, int32_t r_rot, size_t size,
                   const uint64_t c[4], hipStream_t stream);
int divide_by_vanishing_launch(Fr* a, size_t size, const Fr* t, size_t t_len, hipStream_t stream);
int batch_mont_launch(Fr* a, size_t n, bool to_mont, hipStream_t stream);
int widen_u64_launch(const uint64_t* src, size_t n, Fr* dst, hipStream_t stream);
int max_scalar_bits_launch(const Fr* const* d_cols, size_t count, size_t n, uint32_t* d_words, uint32_t* out_bits,
                           hipStream_t stream);
int eval_polynomial_launch(const Fr* d_poly, size_t n, const uint64_t point[4], Fr* d_tmp, uint64_t out[4],
                           hipStream_t stream);
size_t eval_polynomial_tmp_elems(size_t n);
int eval_polynomial_batch_launch(const Fr* const* d_polys, size_t count, size_t n, const uint64_t* points, Fr* d_tmp,
                                 uint64_t* out, hipStream_t stream);
size_t eval_polynomial_batch_tmp_bytes(size_t count, size_t n);
int batch_invert_launch(Fr* d_a, Fr* d_tmp, size_t n, hipStream_t stream);
int lincomb_launch(Fr* res, const Fr* const* polys, const uint64_t* coeffs, size_t count, size_t size, hipStream_t stream);
int perm_sigma_launch(Fr* out, const uint32_t* map_col, const uint32_t* map_row, size_t n, const uint64_t delta[4],
                      const uint64_t omega[4], hipStream_t stream);
int perm_terms_launch(Fr* num, Fr* den, const Fr* value, const Fr* sigma, size_t n, const uint64_t beta[4],
                      const uint64_t gamma[4], const uint64_t delta_pow[4], const uint64_t omega[4], int first,
                      hipStream_t stream);
int distribute_powers_launch(Fr* a, size_t n, const uint64_t g[4], hipStream_t stream);
int random_fr_launch(const uint8_t key[32], size_t n, uint64_t* d_out, hipStream_t stream);
}  // namespace h2
