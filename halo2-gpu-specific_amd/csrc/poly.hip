// poly.hip -- batched polynomial arithmetic as fused elementwise kernels (HBM-bound:
// 32 B per operand per element, one 2 x dwordx4 access per lane).
//
// Replaces the ec-gpu-gen kernels named at the reference's launch sites:
//   eval_mul_c / eval_sum_c / eval_sum / eval_mul / eval_lctheta / eval_lcbeta / eval_addgamma /
//   eval_constant  (plonk/evaluation_gpu.rs:148-163,202-217,246-259,279-305,560,579-585,622-669)
//   batch_mont / batch_unmont  (arithmetic.rs:235-241,280-286)
// and the CPU loops Polynomial +,-,*scalar (poly.rs:191-257) and
// divide_by_vanishing_poly (poly/domain.rs:354-373).
#include "common.hpp"
#include "poly.hpp"

namespace h2 {

__device__ __forceinline__ size_t rot_index(size_t i, int32_t rot, size_t size) {
    // (i + rot) mod size for |rot| < size   (get_rotation_idx, plonk/evaluation.rs:40-42)
    long long v = (long long)i + rot;
    if (v < 0) v += (long long)size;
    if (v >= (long long)size) v -= (long long)size;
    return (size_t)v;
}

template <int OP>
__global__ void __launch_bounds__(256) k_eval_op(Fr* res, const Fr* l, const Fr* r, int32_t l_rot, int32_t r_rot,
                                                 size_t size, Fr c) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < size; i += stride) {
        Fr lv, rv, out;
        if (OP != H2_OP_CONSTANT) lv = fp_load(l + rot_index(i, l_rot, size));
        if (OP == H2_OP_SUM || OP == H2_OP_MUL || OP == H2_OP_SUB || OP == H2_OP_LCTHETA || OP == H2_OP_LCBETA)
            rv = fp_load(r + rot_index(i, r_rot, size));
        if (OP == H2_OP_MUL_C) out = fp_mul(lv, c);
        else if (OP == H2_OP_SUM_C || OP == H2_OP_ADDGAMMA) out = fp_add(lv, c);
        else if (OP == H2_OP_SUM) out = fp_add(lv, rv);
        else if (OP == H2_OP_MUL) out = fp_mul(lv, rv);
        else if (OP == H2_OP_SUB) out = fp_sub(lv, rv);
        else if (OP == H2_OP_LCTHETA) out = fp_add(fp_mul(lv, c), rv);
        else if (OP == H2_OP_LCBETA) out = fp_mul(fp_add(lv, c), rv);
        else out = c;
        fp_store(res + i, out);
    }
}

static unsigned grid_for(size_t n) {
    size_t blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;  // grid-stride beyond 16 blocks per CU
    if (blocks == 0) blocks = 1;
    return (unsigned)blocks;
}

static Fr fr_host(const uint64_t v[4]) {
    Fr r;
    for (int i = 0; i < 4; i++) {
        r.l[2 * i] = (uint32_t)v[i];
        r.l[2 * i + 1] = (uint32_t)(v[i] >> 32);
    }
    return r;
}

int eval_op_launch(int op, Fr* res, const Fr* l, const Fr* r, int32_t l_rot, int32_t r_rot, size_t size,
                   const uint64_t c[4], hipStream_t stream) {
    if (size == 0) return H2_OK;
    bool need_l = op != H2_OP_CONSTANT;
    bool need_r = op == H2_OP_SUM || op == H2_OP_MUL || op == H2_OP_SUB || op == H2_OP_LCTHETA || op == H2_OP_LCBETA;
    bool need_c = !(op == H2_OP_SUM || op == H2_OP_MUL || op == H2_OP_SUB);
    if (!res || (need_l && !l) || (need_r && !r) || (need_c && !c)) {
        set_last_error("h2_eval_op: missing operand");
        return H2_ERR_INVALID;
    }
    // in-place is only defined for an un-rotated operand (evaluation_gpu.rs:631-639)
    if ((res == l && need_l && l_rot != 0) || (res == r && need_r && r_rot != 0)) {
        set_last_error("h2_eval_op: result aliases a rotated operand");
        return H2_ERR_INVALID;
    }
    // normalise rotations into (-size, size)
    long long sz = (long long)size;
    l_rot = (int32_t)(((long long)l_rot % sz));
    r_rot = (int32_t)(((long long)r_rot % sz));
    Fr cv = c ? fr_host(c) : Fr{};
    dim3 g(grid_for(size)), b(256);
#define H2_CASE(OPC) \
    case OPC: hipLaunchKernelGGL(k_eval_op<OPC>, g, b, 0, stream, res, l, r, l_rot, r_rot, size, cv); break;
    switch (op) {
        H2_CASE(H2_OP_MUL_C)
        H2_CASE(H2_OP_SUM_C)
        H2_CASE(H2_OP_SUM)
        H2_CASE(H2_OP_MUL)
        H2_CASE(H2_OP_SUB)
        H2_CASE(H2_OP_LCTHETA)
        H2_CASE(H2_OP_LCBETA)
        H2_CASE(H2_OP_ADDGAMMA)
        H2_CASE(H2_OP_CONSTANT)
        default: set_last_error("h2_eval_op: unknown op"); return H2_ERR_INVALID;
    }
#undef H2_CASE
    H2_HIP(hipGetLastError());
    return H2_OK;
}

// a[i] *= t[i % t_len]; t_len is a power of two (2^(extended_k - k)), table read through L1/L2
__global__ void __launch_bounds__(256) k_divide_by_vanishing(Fr* a, size_t size, const Fr* t, size_t t_mask) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < size; i += stride)
        fp_store(a + i, fp_mul(fp_load(a + i), fp_load(t + (i & t_mask))));
}

int divide_by_vanishing_launch(Fr* a, size_t size, const Fr* t, size_t t_len, hipStream_t stream) {
    if (size == 0) return H2_OK;
    if (!a || !t || t_len == 0 || (t_len & (t_len - 1)) != 0) {
        set_last_error("h2_divide_by_vanishing_poly: t_len must be a non-zero power of two");
        return H2_ERR_INVALID;
    }
    hipLaunchKernelGGL(k_divide_by_vanishing, dim3(grid_for(size)), dim3(256), 0, stream, a, size, t, t_len - 1);
    H2_HIP(hipGetLastError());
    return H2_OK;
}

template <bool TO_MONT>
__global__ void __launch_bounds__(256) k_batch_mont(Fr* a, size_t n) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        Fr v = fp_load(a + i);
        fp_store(a + i, TO_MONT ? fp_to_mont(v) : fp_from_mont(v));
    }
}

int batch_mont_launch(Fr* a, size_t n, bool to_mont, hipStream_t stream) {
    if (n == 0) return H2_OK;
    if (to_mont)
        hipLaunchKernelGGL(k_batch_mont<true>, dim3(grid_for(n)), dim3(256), 0, stream, a, n);
    else
        hipLaunchKernelGGL(k_batch_mont<false>, dim3(grid_for(n)), dim3(256), 0, stream, a, n);
    H2_HIP(hipGetLastError());
    return H2_OK;
}

}  // namespace h2
