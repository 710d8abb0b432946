// msm.hpp -- MSM drivers (msm.hip)
#pragma once
#include "common.hpp"

namespace h2 {
size_t msm_scratch_bytes(size_t n, uint32_t max_bits);
void msm_shape_query(size_t n, uint32_t max_bits, uint32_t* c, uint32_t* windows, uint32_t* buckets_per_window);
void g1_sum_host(const uint64_t* points, size_t count, uint64_t out_xyz[12]);
void msm_identity(uint64_t out_xyz[12]);
// device-resident scalars + bases; result to host memory (synchronises `stream` for the final
// W*G-point read-back and the host-side window combine)
int msm_device(DeviceCtx* ctx, const Fr* d_scalars, const uint64_t* d_bases, size_t n, uint32_t max_bits,
               void* d_scratch, size_t scratch_bytes, uint64_t* out_xyz, hipStream_t stream);
int msm_device_batch(DeviceCtx* ctx, const Fr* const* d_scalars, size_t count, const uint64_t* d_bases, size_t n,
                     uint32_t max_bits, void* d_scratch, size_t scratch_bytes, uint64_t* out_xyz, hipStream_t stream);
int bases_register(const uint64_t* bases, size_t n);
int bases_unregister(const uint64_t* bases);
// h2_poly_register: host Fr vectors the caller will not modify (the proving key's coefficient forms, a proof's final polynomials):
// the same registry / generations / unregister path as the SRS ranges, no table.  poly_resident: the device copy of
// [values, values + n) when it lies inside such a range (uploaded on the device's first use, complete on return), else nullptr.
int poly_register(const uint64_t* values, size_t n);
const Fr* poly_resident(DeviceCtx* ctx, const uint64_t* values, size_t n);
int msm_host(DeviceCtx* ctx, const uint64_t* scalars, const uint64_t* bases, size_t n, uint32_t max_bits,
             uint64_t out_xyz[12]);
int msm_host_resident_scalars(DeviceCtx* ctx, const Fr* d_scalars, const uint64_t* bases, size_t n,
                              uint32_t max_bits, uint64_t out_xyz[12]);
int msm_host_multi(const uint64_t* scalars, const uint64_t* bases, size_t n, uint32_t max_bits, uint64_t out_xyz[12]);
int random_points_launch(uint64_t seed, size_t n, uint64_t* d_out, hipStream_t stream);
int points_decompress_launch(const void* d_bytes, size_t n, uint64_t* d_out, uint32_t* d_bad, hipStream_t stream);
int points_compress_launch(const uint64_t* d_points, size_t n, void* d_bytes, hipStream_t stream);
int msm_device_batch_ex(DeviceCtx* ctx, const Fr* const* d_scalars, const uint64_t* const* bases_each,
                        const uint32_t* bits_each, size_t count, const uint64_t* d_bases, size_t n, uint32_t max_bits,
                        void* d_scratch, size_t scratch_bytes, uint64_t* out_xyz, hipStream_t stream);
int fixed_base_mul_launch(const Fr* d_scalars, const uint64_t* d_table, size_t n, uint64_t* d_out, hipStream_t stream);
size_t msm_batch_scratch_bytes(size_t n, uint32_t max_bits, size_t count);
// shifted-base table of a device-resident base set (msm.hip "shifted-base tables"): built once, used by every
// msm_device* call whose bases lie inside [d_bases, d_bases + n)
int bases_precompute(const uint64_t* d_bases, size_t n, uint32_t digits, hipStream_t stream);
int bases_forget(const uint64_t* d_bases);
size_t bases_precompute_bytes(size_t n, uint32_t digits);
size_t msm_library_bytes(DeviceCtx* ctx);
int g1_fold_launch(const uint64_t* d_points, uint32_t world, uint32_t count, uint64_t* d_out, hipStream_t stream);

}  // namespace h2
