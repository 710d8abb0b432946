// tests/fp_mul_const_check.hip -- the constant-operand product of the NTT's twiddles (csrc/field.hpp fp_mul_const: quotient from the
// anti-diagonals >= 6, exact or ONE SHORT) as the compiled gfx950 code runs it, against host arithmetic, on operands that include
// the case the transforms meet about once in 2^29 products: x built so that x w' mod 2^256 is tiny and the truncated quotient IS
// short.  Compiled and run by tests/test_gpu_numerics.py (TEST INFRASTRUCTURE: it includes the product's header, nothing else).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../halo2-gpu-specific_amd/csrc/field.hpp"
using namespace h2;

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__);      \
            return 1;                                                              \
        }                                                                          \
    } while (0)

template <class F>
__global__ void k_const(const F* x, const F* w, const F* wq, F* raw, F* fixed, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const F a = fp_load(x + i), b = fp_load(w + i), c = fp_load(wq + i);
    fp_store(raw + i, fp_mul_const_dev(a, b, c));   // before the rare subtraction
    fp_store(fixed + i, fp_mul_const(a, b, c));     // what the passes use: below 2p
}

static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() {
    g_state ^= g_state << 13;
    g_state ^= g_state >> 7;
    g_state ^= g_state << 17;
    return (uint32_t)(g_state >> 16);
}
struct U8 {
    uint32_t l[8];
};
static U8 mul_lo(const U8& a, const U8& b) {  // a b mod 2^256
    U8 r{};
    for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
        for (int j = 0; i + j < 8; j++) {
            c += (uint64_t)a.l[i] * b.l[j] + r.l[i + j];
            r.l[i + j] = (uint32_t)c;
            c >>= 32;
        }
    }
    return r;
}
static U8 inv_mod_2_256(const U8& a) {  // a odd: Newton, y <- y (2 - a y)
    U8 y{};
    y.l[0] = 1;
    for (int it = 0; it < 9; it++) {
        U8 ay = mul_lo(a, y), two{};
        two.l[0] = 2;
        uint64_t bw = 0;
        for (int i = 0; i < 8; i++) {  // two - ay
            uint64_t t = (uint64_t)two.l[i] - ay.l[i] - bw;
            two.l[i] = (uint32_t)t;
            bw = (t >> 32) & 1;
        }
        y = mul_lo(y, two);
    }
    return y;
}
template <class P>
static int geq(const Fp<P>& a, const uint32_t* m) {
    for (int i = 7; i >= 0; i--)
        if (a.l[i] != m[i]) return a.l[i] > m[i];
    return 1;
}

template <class P>
static int run(const char* name) {
    using F = Fp<P>;
    const int n_random = 1 << 16, n_short = 4096, n = n_random + n_short;
    std::vector<F> x(n), w(n), wq(n), raw(n), fixed(n);
    for (int i = 0; i < n; i++) {
        F wm;  // a random residue in Montgomery form -> the pair the tables hold
        do {
            for (int l = 0; l < 8; l++) wm.l[l] = rnd();
            wm.l[7] &= 0x3fffffffu;
        } while (geq(wm, P::MOD));
        if (i == 0) wm = fp_zero<P>();
        if (i == 1) wm = fp_one<P>();
        fp_const_pair(wm, w[i], wq[i]);
        for (int l = 0; l < 8; l++) x[i].l[l] = rnd();           // ANY 256-bit value
        if (i < 8) for (int l = 0; l < 8; l++) x[i].l[l] = (i & 1) ? 0xffffffffu : 0u;
        if (i >= n_random && (wq[i].l[0] & 1)) {
            // x = t / w' mod 2^256 with t < 2^200: x w' mod 2^256 = t is far below 7 * 2^224 -- the truncated quotient is short
            U8 t{}, q8;
            for (int l = 0; l < 6; l++) t.l[l] = rnd();
            t.l[6] = rnd() & 0xff;
            for (int l = 0; l < 8; l++) q8.l[l] = wq[i].l[l];
            U8 xi = mul_lo(t, inv_mod_2_256(q8));
            for (int l = 0; l < 8; l++) x[i].l[l] = xi.l[l];
        }
    }
    F *dx, *dw, *dq, *dr, *df;
    const size_t bytes = (size_t)n * sizeof(F);
    CK(hipMalloc(&dx, bytes)); CK(hipMalloc(&dw, bytes)); CK(hipMalloc(&dq, bytes)); CK(hipMalloc(&dr, bytes)); CK(hipMalloc(&df, bytes));
    CK(hipMemcpy(dx, x.data(), bytes, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, w.data(), bytes, hipMemcpyHostToDevice));
    CK(hipMemcpy(dq, wq.data(), bytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_const<F>, dim3((n + 255) / 256), dim3(256), 0, 0, dx, dw, dq, dr, df, n);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(raw.data(), dr, bytes, hipMemcpyDeviceToHost));
    CK(hipMemcpy(fixed.data(), df, bytes, hipMemcpyDeviceToHost));
    int bad = 0, over = 0, short_seen = 0;
    for (int i = 0; i < n; i++) {
        // host: the exact quotient (all 64 limb products): x w - q p, below 2p
        const F exact = fp_mul_const(x[i], w[i], wq[i]);
        const bool raw_is_exact = fp_eq(raw[i], exact);
        if (!raw_is_exact) {
            // then the device quotient was one short: raw = exact + p, at or above 2p only then
            F plus = exact;
            uint64_t c = 0;
            for (int l = 0; l < 8; l++) {
                c += (uint64_t)plus.l[l] + P::MOD[l];
                plus.l[l] = (uint32_t)c;
                c >>= 32;
            }
            if (!fp_eq(raw[i], plus)) bad++;
            short_seen++;
        }
        if (geq(fixed[i], P::MOD2)) over++;                                  // the contract of the lazy domain: below 2p
        if (!fp_eq(fp_lazy_canon(fixed[i]), fp_lazy_canon(exact))) bad++;    // the same residue
    }
    printf("%s fp_mul_const device vs host on %d operand triples (%d built for a short quotient): %d mismatches, %d results at or above 2p, "
           "%d short quotients met\n", name, n, n_short, bad, over, short_seen);
    return (bad || over || short_seen < n_short / 4) ? 1 : 0;
}

int main() {
    int rc = run<FrParams>("Fr");
    rc |= run<FqParams>("Fq");
    return rc;
}
