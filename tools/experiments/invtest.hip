// invtest.hip -- fp_inv (Kaliski) against a^(p-2) on the host and on the device, random and edge values, both fields.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../halo2-gpu-specific_amd/csrc/field.hpp"
using namespace h2;
static uint64_t st = 0x1234567ull;
static uint64_t next64() { uint64_t z = (st += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
template <class P> H2_DEV Fp<P> fermat(const Fp<P>& a) {
    uint32_t e[8];
    for (int i = 0; i < 8; i++) e[i] = P::MOD[i];
    e[0] -= 2;  // both moduli end in ...01 / ...47: no borrow
    Fp<P> acc = fp_one<P>();
    for (int bit = 253; bit >= 0; bit--) { acc = fp_sqr(acc); if ((e[bit >> 5] >> (bit & 31)) & 1) acc = fp_mul(acc, a); }
    return acc;
}
template <class P> __global__ void k_inv(Fp<P>* out, const Fp<P>* in, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) fp_store(out + i, fp_inv(fp_load(in + i)));
}
template <class P> static bool lt_mod(const Fp<P>& a) { for (int i = 7; i >= 0; i--) if (a.l[i] != P::MOD[i]) return a.l[i] < P::MOD[i]; return false; }
template <class P> int run(const char* name, bool device) {
    typedef Fp<P> F;
    const int n = 4096;
    std::vector<F> a(n), got(n);
    for (int i = 0; i < n; i++) { for (int j = 0; j < 8; j++) a[i].l[j] = (uint32_t)next64(); a[i].l[7] &= 0x3fffffffu; if (!lt_mod(a[i])) a[i] = fp_reduce_once(a[i]); }
    F one = fp_zero<P>(); one.l[0] = 1;
    F pm1; for (int j = 0; j < 8; j++) pm1.l[j] = P::MOD[j]; pm1.l[0] -= 1;
    a[0] = fp_zero<P>(); a[1] = one; a[2] = pm1; a[3] = fp_one<P>(); a[4] = fp_dbl(one); a[5] = fp_neg(fp_one<P>());
    for (int i = 6; i < 40; i++) { a[i] = fp_zero<P>(); a[i].l[(i - 6) / 5] = 1u << ((i * 7) & 31); }   // powers of two
    int bad = 0;
    for (int i = 0; i < n; i++) {
        F want = fp_is_zero(a[i]) ? a[i] : fermat(a[i]);
        F h = fp_inv(a[i]);
        if (!fp_eq(h, want)) bad++;
    }
    printf("%s host   fp_inv vs a^(p-2) on %d values: %d mismatches\n", name, n, bad);
    if (device) {
        F *d_in, *d_out;
        hipMalloc(&d_in, n * sizeof(F)); hipMalloc(&d_out, n * sizeof(F));
        hipMemcpy(d_in, a.data(), n * sizeof(F), hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_inv<P>, dim3(n / 64), dim3(64), 0, 0, d_out, d_in, n);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_inv<P>, dim3(n / 64), dim3(64), 0, 0, d_out, d_in, n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(got.data(), d_out, n * sizeof(F), hipMemcpyDeviceToHost);
        int dbad = 0;
        for (int i = 0; i < n; i++) { F want = fp_is_zero(a[i]) ? a[i] : fermat(a[i]); if (!fp_eq(got[i], want)) dbad++; }
        printf("%s device fp_inv vs a^(p-2) on %d values: %d mismatches; one wave per workgroup: %.1f us per launch\n", name, n, dbad, ms * 1e3);
        bad += dbad;
        // the shape k_batch_invert has: every lane of the wave holds the same value
        for (int i = 0; i < n; i++) a[i] = a[100 + (i / 64)];
        hipMemcpy(d_in, a.data(), n * sizeof(F), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_inv<P>, dim3(n / 64), dim3(64), 0, 0, d_out, d_in, n);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_inv<P>, dim3(n / 64), dim3(64), 0, 0, d_out, d_in, n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("%s device, one value per wave: %.1f us per launch\n", name, ms * 1e3);
    }
    return bad;
}
int main(int argc, char** argv) {
    bool device = argc > 1;
    int bad = run<FrParams>("Fr", device) + run<FqParams>("Fq", device);
    return bad ? 1 : 0;
}
