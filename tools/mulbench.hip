// tools/mulbench.hip -- throughput of the Montgomery multiplier variants at several occupancies.
// Each thread runs CHAINS independent dependent-chains of ITERS multiplications.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../halo2-gpu-specific_amd/csrc/field.hpp"
using namespace h2;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 256;
extern __shared__ uint4 dyn[];

template <int MODE, int CHAINS>
__global__ void __launch_bounds__(256) k(Fr* out, const Fr* in) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    Fr x[CHAINS], w = fp_load(in + (i & 1023));
#pragma unroll
    for (int c = 0; c < CHAINS; c++) x[c] = fp_load(in + ((i + 7 * c) & 1023));
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) x[c] = fp_mul(x[c], w);
    }
    Fr acc = x[0];
#pragma unroll
    for (int c = 1; c < CHAINS; c++) acc = fp_add(acc, x[c]);
    fp_store(out + i, acc);
    if (dyn[0].x == 0x12345) out[0].l[0] = 1;  // keep the dynamic LDS allocation alive
}

template <int MODE, int CHAINS>
int run(const char* name, int lds_bytes, Fr* d_out, Fr* d_in) {
    int blocks = 256 * 32;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<MODE, CHAINS>), dim3(blocks), dim3(256), lds_bytes, 0, d_out, d_in);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<MODE, CHAINS>), dim3(blocks), dim3(256), lds_bytes, 0, d_out, d_in);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double mults = (double)blocks * 256 * ITERS * CHAINS;
    printf("%-28s chains=%d lds=%6d B/blk  %8.3f ms  %.3e mul/s\n", name, CHAINS, lds_bytes, ms, mults / (ms * 1e-3));
    return 0;
}

int main() {
    Fr *d_in, *d_out;
    CK(hipMalloc(&d_in, 1024 * 32)); CK(hipMalloc(&d_out, (size_t)256 * 32 * 256 * 32));
    CK(hipMemset(d_in, 0x5a, 1024 * 32));
    // LDS per block controls occupancy: 160 KiB/CU -> 20 KiB = 8 blocks (32 waves/CU), 40 KiB = 4 blocks (16 waves), 80 KiB = 2 blocks (8 waves), 160 KiB = 1 (4 waves)
    int ldss[4] = {16 * 1024, 40 * 1024, 80 * 1024, 160 * 1024};
    for (int l = 0; l < 4; l++) {
        run<0, 1>("fp_mul", ldss[l], d_out, d_in);
        run<0, 2>("fp_mul", ldss[l], d_out, d_in);
        run<0, 4>("fp_mul", ldss[l], d_out, d_in);
    }
    return 0;
}
