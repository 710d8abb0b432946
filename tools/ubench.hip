// tools/ubench.hip -- instruction-rate microbenchmarks for gfx950 integer/FP64 multiply paths.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/ubench
// Used to size the modular-multiplication roofline ("the builder must
// microbenchmark v_mad_u64_u32 issue rate on gfx950").
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 8;   // independent chains per thread

template <int OP> __global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x;
    uint64_t acc[UNROLL];
    double d[UNROLL];
    for (int i = 0; i < UNROLL; i++) { acc[i] = a + i; d[i] = 1.0 + i; }
    double da = 1.0000001 + a * 1e-9, db = 0.5 + b * 1e-12;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
            if (OP == 0) {  // v_mad_u64_u32
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
            } else if (OP == 1) {  // v_mul_lo_u32
                uint32_t x = (uint32_t)acc[i];
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(b));
                acc[i] = x;
            } else if (OP == 2) {  // v_mul_hi_u32
                uint32_t x = (uint32_t)acc[i];
                asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(b));
                acc[i] = x;
            } else if (OP == 3) {  // v_mad_u32_u24
                uint32_t x = (uint32_t)acc[i];
                asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(a));
                acc[i] = x;
            } else if (OP == 4) {  // v_fma_f64
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(da), "v"(db));
            } else if (OP == 5) {  // v_add_co_u32 + v_addc_co_u32 pair (64-bit add)
                uint32_t lo = (uint32_t)acc[i], hi = (uint32_t)(acc[i] >> 32);
                asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc");
                acc[i] = ((uint64_t)hi << 32) | lo;
            } else if (OP == 6) {  // v_add_u32 (plain full-rate reference)
                uint32_t x = (uint32_t)acc[i];
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b));
                acc[i] = x;
            } else if (OP == 7) {  // v_mul_u32_u24
                uint32_t x = (uint32_t)acc[i];
                asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(b));
                acc[i] = x;
            } else if (OP == 8) {  // v_mad_u64_u32 with SGPR carry + v_addc (Comba step)
                uint32_t hi2 = (uint32_t)d[i];
                uint64_t c;
                asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc[i]), "=s"(c) : "v"(a), "v"(b));
                asm volatile("v_addc_co_u32 %0, %1, %0, 0, %1" : "+v"(hi2), "+s"(c));
                d[i] = hi2;
            } else if (OP == 9) {  // v_mul_f64
                asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(da));
            } else if (OP == 10) { // v_pk_mul_lo_u16 (packed 16-bit)
                uint32_t x = (uint32_t)acc[i];
                asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(x) : "v"(b));
                acc[i] = x;
            } else if (OP == 11) { // v_mad_i32_i24
                uint32_t x = (uint32_t)acc[i];
                asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x) : "v"(b));
                acc[i] = x;
            }
        }
    }
    uint64_t s = 0; double ds = 0;
    for (int i = 0; i < UNROLL; i++) { s += acc[i]; ds += d[i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s + (uint32_t)(s >> 32) + (uint32_t)ds;
}

template <int OP> int run(const char* name, int ops_per_iter, uint32_t* d_out, int blocks) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u + r);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double lane_ops = (double)reps * blocks * 256 * ITERS * UNROLL * ops_per_iter;
    double rate = lane_ops / (ms * 1e-3);
    // lanes/clk/SIMD assuming 256 CUs x 4 SIMDs at 2.4 GHz
    printf("%-44s %8.3f ms  %10.3f Glane-op/s  %6.2f lanes/clk/SIMD@2.4GHz\n", name, ms / reps, rate * 1e-9, rate / (256.0 * 4 * 2.4e9));
    return 0;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs=%d clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    int blocks = p.multiProcessorCount * 8;   // 8 blocks x 4 waves = 32 waves/CU
    uint32_t* d_out; CK(hipMalloc(&d_out, (size_t)blocks * 256 * 4));
    run<6>("v_add_u32", 1, d_out, blocks);
    run<0>("v_mad_u64_u32", 1, d_out, blocks);
    run<8>("v_mad_u64_u32(sgpr carry)+v_addc_co_u32", 1, d_out, blocks);
    run<1>("v_mul_lo_u32", 1, d_out, blocks);
    run<2>("v_mul_hi_u32", 1, d_out, blocks);
    run<3>("v_mad_u32_u24", 1, d_out, blocks);
    run<7>("v_mul_u32_u24", 1, d_out, blocks);
    run<11>("v_mul_hi_u32_u24", 1, d_out, blocks);
    run<10>("v_pk_mul_lo_u16", 1, d_out, blocks);
    run<4>("v_fma_f64", 1, d_out, blocks);
    run<9>("v_mul_f64", 1, d_out, blocks);
    run<5>("v_add_co_u32+v_addc_co_u32 (pair)", 1, d_out, blocks);
    return 0;
}
