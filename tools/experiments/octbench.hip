// octbench.hip -- EXPERIMENT: the octet Montgomery product of oct.hpp against the single-lane fp_mul: bit-exactness on
// random and edge inputs, then the LATENCY of a dependent chain (one wave), which is what the MSM tail kernels are bound by.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/experiments/octbench.hip -o tools/experiments/octbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "oct.hpp"
using namespace h2;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <class P>
__global__ void __launch_bounds__(256) k_oct_check(Fp<P>* out, const Fp<P>* a, const Fp<P>* b, int n) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, o = t >> 3, k = t & 7;
    if (o >= n) return;
    const OctCtx c = oct_ctx<P>();
    out[o].l[k] = oct_mul<P>(a[o].l[k], b[o].l[k], c);
}

template <class P>
__global__ void __launch_bounds__(64) k_chain_oct(Fp<P>* out, const Fp<P>* in, int iters) {
    const int o = threadIdx.x >> 3, k = threadIdx.x & 7;
    const OctCtx c = oct_ctx<P>();
    uint32_t x = in[o].l[k];
    const uint32_t w = in[o + 8].l[k];
    for (int i = 0; i < iters; i++) x = oct_mul<P>(x, w, c);
    out[o].l[k] = x;
}
template <class P>
__global__ void __launch_bounds__(64) k_chain_lane(Fp<P>* out, const Fp<P>* in, int iters) {
    Fp<P> x = fp_load(in + threadIdx.x);
    const Fp<P> w = fp_load(in + 64 + threadIdx.x);
    for (int i = 0; i < iters; i++) x = fp_mul(x, w);
    fp_store(out + threadIdx.x, x);
}

static uint64_t rng_state = 0x48414c4f33ull;
static uint64_t next64() {
    uint64_t z = (rng_state += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
template <class P>
static bool lt_mod(const Fp<P>& a) {
    for (int i = 7; i >= 0; i--)
        if (a.l[i] != P::MOD[i]) return a.l[i] < P::MOD[i];
    return false;
}

template <class P>
int check(const char* name) {
    typedef Fp<P> F;
    const int n = 1 << 15;
    std::vector<F> a(n), b(n), got(n);
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < 8; j++) { a[i].l[j] = (uint32_t)next64(); b[i].l[j] = (uint32_t)next64(); }
        a[i].l[7] &= 0x3fffffffu;
        b[i].l[7] &= 0x3fffffffu;
        if (!lt_mod(a[i])) a[i] = fp_reduce_once(a[i]);
        if (!lt_mod(b[i])) b[i] = fp_reduce_once(b[i]);
    }
    F zero = fp_zero<P>(), one = fp_zero<P>(), pm1, ones;
    one.l[0] = 1;
    for (int j = 0; j < 8; j++) { pm1.l[j] = P::MOD[j]; ones.l[j] = 0xffffffffu; }
    pm1.l[0] -= 1;
    ones.l[7] = 0x1fffffffu;
    F edges[4] = {zero, one, pm1, ones};
    int e = 0;
    for (int x = 0; x < 4; x++)
        for (int y = 0; y < 4; y++) { a[e] = edges[x]; b[e] = edges[y]; e++; }
    F *da, *db, *dg;
    CK(hipMalloc(&da, n * sizeof(F))); CK(hipMalloc(&db, n * sizeof(F))); CK(hipMalloc(&dg, n * sizeof(F)));
    CK(hipMemcpy(da, a.data(), n * sizeof(F), hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b.data(), n * sizeof(F), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_oct_check<P>, dim3(n * 8 / 256), dim3(256), 0, 0, dg, da, db, n);
    CK(hipMemcpy(got.data(), dg, n * sizeof(F), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < n; i++) {
        F want = fp_mul(a[i], b[i]);
        if (!fp_eq(want, got[i])) {
            if (bad < 3) {
                printf("  %s mismatch at %d: want", name, i);
                for (int j = 7; j >= 0; j--) printf(" %08x", want.l[j]);
                printf("\n                     got ");
                for (int j = 7; j >= 0; j--) printf(" %08x", got[i].l[j]);
                printf("\n");
            }
            bad++;
        }
    }
    printf("%-3s octet product vs host fp_mul on %d pairs incl. edge values: %d mismatches\n", name, n, bad);
    // latency of a dependent chain, one wave
    const int iters = 4096;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms_o = 0, ms_l = 0;
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_chain_oct<P>, dim3(1), dim3(64), 0, 0, dg, da, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms_o, e0, e1));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_chain_lane<P>, dim3(1), dim3(64), 0, 0, dg, da, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms_l, e0, e1));
    }
    printf("%-3s latency per dependent product, one wave: single lane %.0f ns, octet %.0f ns (x%.2f)\n", name, ms_l * 1e6 / iters,
           ms_o * 1e6 / iters, ms_l / ms_o);
    hipFree(da); hipFree(db); hipFree(dg);
    return bad;
}

int main() {
    int bad = check<FrParams>("Fr") + check<FqParams>("Fq");
    return bad ? 1 : 0;
}
