// tools/membench.hip -- what HBM streaming reaches on this device with the plainest kernels, as the yardstick for the
// elementwise family (csrc/poly.hip k_eval_op): copy / read-only / write-only / two reads + one write, 16 bytes per lane
// per access, grid-stride, 1 GiB per operand.   Build: make -C tools membench     Run: ./tools/membench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_copy(const uint4* a, uint4* out, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) out[i] = a[i];
}
__global__ void __launch_bounds__(256) k_read(const uint4* a, uint4* out, size_t n16) {
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = a[i];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = acc;  // never true in practice: keeps the loads alive
}
__global__ void __launch_bounds__(256) k_write(uint4* out, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        out[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
__global__ void __launch_bounds__(256) k_add(const uint4* a, const uint4* b, uint4* out, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        uint4 x = a[i], y = b[i];
        out[i] = make_uint4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    }
}
// the elementwise kernels' own access shape: one 32-byte element per lane as two 16-byte accesses at a 32-byte stride
__global__ void __launch_bounds__(256) k_add32(const uint4* a, const uint4* b, uint4* out, size_t n32) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n32; i += (size_t)gridDim.x * blockDim.x) {
        uint4 x0 = a[2 * i], x1 = a[2 * i + 1], y0 = b[2 * i], y1 = b[2 * i + 1];
        out[2 * i] = make_uint4(x0.x + y0.x, x0.y + y0.y, x0.z + y0.z, x0.w + y0.w);
        out[2 * i + 1] = make_uint4(x1.x + y1.x, x1.y + y1.y, x1.z + y1.z, x1.w + y1.w);
    }
}

int main(int argc, char** argv) {
    const size_t bytes = (size_t)1 << 30, n16 = bytes / 16;
    uint4 *a, *b, *c;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMalloc(&c, bytes));
    CK(hipMemset(a, 1, bytes));
    CK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int reps = 20;
    for (int per_cu : {4, 8, 16, 32, 0}) {
        const unsigned grid = per_cu ? 256u * per_cu : (unsigned)(n16 / 256);
        auto run = [&](const char* name, double moved, auto launch) {
            launch();
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < reps; r++) launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            printf("grid %8u  %-34s %7.3f ms  %7.1f GB/s\n", grid, name, ms / reps, moved / (ms / reps * 1e-3) / 1e9);
        };
        run("copy  (1 read + 1 write)", 2.0 * bytes, [&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, c, n16); });
        run("read  (1 read)", 1.0 * bytes, [&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, c, n16); });
        run("write (1 write)", 1.0 * bytes, [&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, c, n16); });
        run("add   (2 reads + 1 write)", 3.0 * bytes, [&] { hipLaunchKernelGGL(k_add, dim3(grid), dim3(256), 0, 0, a, b, c, n16); });
        run("add32 (2 reads + 1 write, 32 B lanes)", 3.0 * bytes,
            [&] { hipLaunchKernelGGL(k_add32, dim3(grid > n16 / 512 ? (unsigned)(n16 / 512) : grid), dim3(256), 0, 0, a, b, c, n16 / 2); });
    }
    {
        hipEventRecord(e0);
        for (int r = 0; r < reps; r++) hipMemcpyAsync(c, a, bytes, hipMemcpyDeviceToDevice, 0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("hipMemcpyAsync device-to-device 1 GiB     %7.3f ms  %7.1f GB/s (read + write)\n", ms / reps, 2.0 * bytes / (ms / reps * 1e-3) / 1e9);
    }
    return 0;
}
