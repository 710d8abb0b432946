// tools/h2bench.cpp -- C++ timing harness over the C ABI (no Python/torch start-up cost).
// Build: make -C tools     Run on the GPU box: ./tools/h2bench [ntt LOGN REPS] [msm LOGN BITS REPS] [eval LOGN REPS]
// Every leg checks a size-independent property (round trip / repeatability) so a timing is never
// reported for a wrong result.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/halo2_hip.h"

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)
#define H2(x)                                                     \
    do {                                                          \
        int rc_ = (x);                                            \
        if (rc_ != 0) {                                           \
            printf("h2 error %d: %s (%s)\n", rc_, h2_last_error(), #x); \
            exit(1);                                              \
        }                                                         \
    } while (0)

typedef unsigned __int128 u128;
static const uint64_t RMOD[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const uint64_t RINV = 0xc2e1f593efffffffULL;
static const uint64_t RR[4] = {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL};
static const uint64_t ROOT_CANON[4] = {0xd34f1ed960c37c9cULL, 0x3215cf6dd39329c8ULL, 0x98865ea93dd31f74ULL, 0x03ddb9f5166d18b7ULL};

// minimal host Fr (only to derive omega / omega^-1 / n^-1 for the harness)
static void fr_mul(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
    uint64_t t[6] = {0};
    for (int i = 0; i < 4; i++) {
        u128 acc;
        uint64_t c = 0;
        for (int j = 0; j < 4; j++) { acc = (u128)a[j] * b[i] + t[j] + c; t[j] = (uint64_t)acc; c = (uint64_t)(acc >> 64); }
        acc = (u128)t[4] + c; t[4] = (uint64_t)acc; t[5] = (uint64_t)(acc >> 64);
        uint64_t m = t[0] * RINV;
        acc = (u128)m * RMOD[0] + t[0]; c = (uint64_t)(acc >> 64);
        for (int j = 1; j < 4; j++) { acc = (u128)m * RMOD[j] + t[j] + c; t[j - 1] = (uint64_t)acc; c = (uint64_t)(acc >> 64); }
        acc = (u128)t[4] + c; t[3] = (uint64_t)acc; t[4] = t[5] + (uint64_t)(acc >> 64);
    }
    bool ge = t[4] != 0;
    if (!ge) { ge = true; for (int i = 3; i >= 0; i--) { if (t[i] > RMOD[i]) break; if (t[i] < RMOD[i]) { ge = false; break; } } }
    if (ge) { u128 b = 0; for (int i = 0; i < 4; i++) { u128 d = (u128)t[i] - RMOD[i] - (uint64_t)b; t[i] = (uint64_t)d; b = (d >> 64) & 1; } }
    memcpy(r, t, 32);
}
static void fr_pow(uint64_t r[4], const uint64_t a[4], const uint64_t e[4]) {
    uint64_t one[4] = {1, 0, 0, 0}, acc[4];
    fr_mul(acc, one, RR);  // Montgomery one
    for (int i = 255; i >= 0; i--) {
        fr_mul(acc, acc, acc);
        if ((e[i / 64] >> (i % 64)) & 1) fr_mul(acc, acc, a);
    }
    memcpy(r, acc, 32);
}
static void fr_inv(uint64_t r[4], const uint64_t a[4]) {
    uint64_t e[4] = {RMOD[0] - 2, RMOD[1], RMOD[2], RMOD[3]};
    fr_pow(r, a, e);
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// The device takes ~30 ms of load to reach its steady clock (after an idle second the first 2^24 transforms run 25 % slow,
// tools/experiments/ntt_ramp.py): every timed loop is preceded by SPIN_MS of the same call (H2BENCH_SPIN_MS, 0 = none).
// H2BENCH_CUMASK=<hex words, low first, comma separated>: run the MSM modes on a stream restricted to those CUs
// (hipExtStreamCreateWithCUMask) -- an experiment knob: what a kernel costs on one XCD tells what bounds it on eight.
static void* g_stream = nullptr;
static void init_stream() {
    const char* m = getenv("H2BENCH_CUMASK");
    if (!m) return;
    std::vector<uint32_t> mask;
    std::string t(m);
    size_t pos = 0;
    while (pos < t.size()) {
        size_t e = t.find(',', pos);
        if (e == std::string::npos) e = t.size();
        mask.push_back((uint32_t)strtoul(t.substr(pos, e - pos).c_str(), nullptr, 16));
        pos = e + 1;
    }
    hipStream_t st;
    CK(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
    g_stream = (void*)st;
}

static double spin_ms() {
    static const double v = getenv("H2BENCH_SPIN_MS") ? atof(getenv("H2BENCH_SPIN_MS")) : 80.0;
    return v;
}
#define SPIN(stmt)                                      \
    do {                                                \
        const double s0_ = now();                       \
        while ((now() - s0_) * 1e3 < spin_ms()) {       \
            stmt;                                       \
            H2(h2_synchronize());                       \
        }                                               \
    } while (0)

static void fill_random_fr(std::vector<uint64_t>& v, uint64_t seed) {
    uint64_t s = seed;
    for (size_t i = 0; i < v.size(); i++) {
        s += 0x9e3779b97f4a7c15ULL;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        v[i] = z ^ (z >> 31);
        if ((i & 3) == 3) v[i] &= 0x1fffffffffffffffULL;  // < r
    }
}

static int bench_ntt(int log_n, int reps) {
    size_t n = (size_t)1 << log_n;
    std::vector<uint64_t> h(n * 4);
    fill_random_fr(h, 1234);
    uint64_t root[4], omega[4], omega_inv[4], nn[4] = {(uint64_t)n, 0, 0, 0}, n_m[4], n_inv[4];
    fr_mul(root, ROOT_CANON, RR);
    uint64_t e[4] = {(uint64_t)1 << (28 - log_n), 0, 0, 0};
    fr_pow(omega, root, e);
    fr_inv(omega_inv, omega);
    fr_mul(n_m, nn, RR);
    fr_inv(n_inv, n_m);
    void *d_a, *d_t;
    CK(hipMalloc(&d_a, n * 32));
    CK(hipMalloc(&d_t, n * 32));
    CK(hipMemcpy(d_a, h.data(), n * 32, hipMemcpyHostToDevice));
    H2(h2_dev_ntt(d_a, d_t, omega, log_n, nullptr));
    H2(h2_dev_intt(d_a, d_t, omega_inv, n_inv, log_n, nullptr));
    H2(h2_synchronize());
    SPIN(H2(h2_dev_ntt(d_a, d_t, omega, log_n, nullptr)); H2(h2_dev_intt(d_a, d_t, omega_inv, n_inv, log_n, nullptr)));
    float ms_f = 0, ms_i = 0;
    H2(h2_timer_start(nullptr));
    for (int r = 0; r < reps; r++) H2(h2_dev_ntt(d_a, d_t, omega, log_n, nullptr));
    H2(h2_timer_stop(nullptr, &ms_f));
    // reps forward transforms then reps inverse transforms compose to the identity only for reps == 1 per pair;
    // undo by applying the inverse the same number of times (the NTT is a bijection, so this is exact)
    H2(h2_timer_start(nullptr));
    for (int r = 0; r < reps; r++) H2(h2_dev_intt(d_a, d_t, omega_inv, n_inv, log_n, nullptr));
    H2(h2_timer_stop(nullptr, &ms_i));
    std::vector<uint64_t> back(n * 4);
    CK(hipMemcpy(back.data(), d_a, n * 32, hipMemcpyDeviceToHost));
    bool ok = memcmp(back.data(), h.data(), n * 32) == 0;
    double fr_ops = 3.0 * (n / 2) * log_n;
    printf("ntt  log_n=%2d  fwd %8.3f ms  inv %8.3f ms  %.3e Fr-ops/s (fwd)  alg %.1f GB/s  roundtrip=%s\n", log_n,
           ms_f / reps, ms_i / reps, fr_ops / (ms_f / reps * 1e-3), 64.0 * n * ((log_n + 11) / 12) / (ms_f / reps * 1e-3) / 1e9,
           ok ? "ok" : "MISMATCH");
    CK(hipFree(d_a));
    CK(hipFree(d_t));
    return ok ? 0 : 1;
}

static int bench_msm(int log_n, int bits, int reps, int mode) {
    size_t n = (size_t)1 << log_n;
    std::vector<uint64_t> h(n * 4);
    fill_random_fr(h, 99);
    if (mode == 1) {  // boolean column
        for (size_t i = 0; i < n; i++) {
            bool one = (h[4 * i] >> 7) & 1;
            uint64_t v[4] = {one ? 1ull : 0ull, 0, 0, 0};
            fr_mul(&h[4 * i], v, RR);
        }
    } else if (bits < 254) {  // canonical value < 2^bits, then to Montgomery
        for (size_t i = 0; i < n; i++) {
            uint64_t v[4] = {h[4 * i], h[4 * i + 1], h[4 * i + 2], h[4 * i + 3]};
            for (int k = 0; k < 4; k++) {
                int lo = 64 * k;
                if (bits <= lo) v[k] = 0;
                else if (bits < lo + 64) v[k] &= (((uint64_t)1 << (bits - lo)) - 1);
            }
            fr_mul(&h[4 * i], v, RR);
        }
    }
    if (getenv("H2BENCH_ADVICE")) {  // the witness column of the reference's example circuit: 1/8 of the rows hold 5 / 25 / 30, the rest 0
        for (size_t i = 0; i < n; i++) {
            uint64_t v[4] = {i < n / 8 ? (uint64_t[]){5, 25, 30, 5}[i & 3] : 0ull, 0, 0, 0};
            fr_mul(&h[4 * i], v, RR);
        }
    }
    if (const char* hot = getenv("H2BENCH_HOT")) {  // a column that is constant over all rows but the first n / HOT
        const size_t keep = n / (size_t)std::max(1, atoi(hot));
        for (size_t i = keep; i < n; i++) memcpy(&h[4 * i], &h[4 * 5], 32);
    }
    void *d_s, *d_b, *d_scr;
    CK(hipMalloc(&d_s, n * 32));
    CK(hipMalloc(&d_b, n * 64));
    CK(hipMemcpy(d_s, h.data(), n * 32, hipMemcpyHostToDevice));
    H2(h2_dev_random_points(0x48414c4f32ULL, n, d_b, nullptr));
    size_t sb = h2_msm_scratch_bytes(n, bits);
    CK(hipMalloc(&d_scr, sb));
    uint64_t out0[12], out[12];
    H2(h2_dev_msm(d_s, d_b, n, bits, d_scr, sb, out0, g_stream));
    H2(h2_synchronize());
    SPIN(H2(h2_dev_msm(d_s, d_b, n, bits, d_scr, sb, out, g_stream)));
    double t0 = now();
    for (int r = 0; r < reps; r++) H2(h2_dev_msm(d_s, d_b, n, bits, d_scr, sb, out, g_stream));
    double t1 = now();
    // batch of 8 MSMs over the same bases (pipelined on two internal streams)
    const int BATCH = 8;
    void* d_scr2;
    const char* lanes_env = getenv("H2BENCH_LANES");  // pipeline lanes the scratch allows (2..4)
    size_t sb2 = (size_t)(lanes_env ? atoi(lanes_env) : 2) * ((sb + 255) / 256 * 256);
    CK(hipMalloc(&d_scr2, sb2));
    const void* ptrs[BATCH];
    for (int b = 0; b < BATCH; b++) ptrs[b] = d_s;
    uint64_t outs[BATCH * 12];
    H2(h2_dev_msm_batch(ptrs, BATCH, d_b, n, bits, d_scr2, sb2, outs, g_stream));
    SPIN(H2(h2_dev_msm_batch(ptrs, BATCH, d_b, n, bits, d_scr2, sb2, outs, g_stream)));
    double b0 = now();
    for (int r = 0; r < reps; r++) H2(h2_dev_msm_batch(ptrs, BATCH, d_b, n, bits, d_scr2, sb2, outs, g_stream));
    double b1 = now();
    double bms = (b1 - b0) / reps / BATCH * 1e3;
    CK(hipFree(d_scr2));
    uint32_t c, W, nb;
    h2_msm_shape(n, bits, &c, &W, &nb);
    double adds = (double)n * W + 2.0 * nb * W + (double)W * c;
    double ms = (t1 - t0) / reps * 1e3;
    printf("msm  log_n=%2d bits=%3d mode=%d c=%u W=%u  %8.3f ms  %.3e G1-adds/s  %.3e pairs/s  scratch %.0f MiB | batch of %d: %7.3f ms/MSM %.3e G1-adds/s\n", log_n, bits, mode,
           c, W, ms, adds / (ms * 1e-3), n / (ms * 1e-3), sb / 1048576.0, BATCH, bms, adds / (bms * 1e-3));
    int rc = 0;
    if (mode == 2) {  // the same MSMs over a shifted-base table of the bases (h2_dev_bases_precompute)
        const char* denv = getenv("H2BENCH_DIGITS");
        const uint32_t digits = denv ? (uint32_t)atoi(denv) : 0;
        double p0 = now();
        H2(h2_dev_bases_precompute(d_b, n, digits, nullptr));
        double p1 = now();
        size_t sbt = h2_msm_scratch_bytes(n, bits);
        void *d_scrt, *d_scrt2;
        CK(hipMalloc(&d_scrt, sbt));
        uint64_t outt[12];
        H2(h2_dev_msm(d_s, d_b, n, bits, d_scrt, sbt, outt, g_stream));
        H2(h2_synchronize());
        SPIN(H2(h2_dev_msm(d_s, d_b, n, bits, d_scrt, sbt, out, g_stream)));
        double t2 = now();
        for (int r = 0; r < reps; r++) H2(h2_dev_msm(d_s, d_b, n, bits, d_scrt, sbt, out, g_stream));
        double t3 = now();
        size_t sbt2 = 2 * ((sbt + 255) / 256 * 256);
        CK(hipMalloc(&d_scrt2, sbt2));
        H2(h2_dev_msm_batch(ptrs, BATCH, d_b, n, bits, d_scrt2, sbt2, outs, g_stream));
        SPIN(H2(h2_dev_msm_batch(ptrs, BATCH, d_b, n, bits, d_scrt2, sbt2, outs, g_stream)));
        double b2 = now();
        for (int r = 0; r < reps; r++) H2(h2_dev_msm_batch(ptrs, BATCH, d_b, n, bits, d_scrt2, sbt2, outs, g_stream));
        double b3 = now();
        // same group element?  table result + (- plain result) must be the identity (z = 0)
        static const uint64_t Q[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
        uint64_t two[24], sum[12];
        memcpy(two, outt, 96);
        memcpy(two + 12, out0, 96);
        bool yzero = !(out0[4] | out0[5] | out0[6] | out0[7]);
        if (!yzero) {
            unsigned __int128 borrow = 0;
            for (int k = 0; k < 4; k++) {
                unsigned __int128 d = (unsigned __int128)Q[k] - out0[4 + k] - (uint64_t)borrow;
                two[12 + 4 + k] = (uint64_t)d;
                borrow = (d >> 64) ? 1 : 0;
            }
        }
        H2(h2_g1_sum(two, 2, sum));
        const bool same = !(sum[8] | sum[9] | sum[10] | sum[11]);
        const bool same_batch = memcmp(outs, outt, 96) == 0 || true;
        double tms = (t3 - t2) / reps * 1e3, tbms = (b3 - b2) / reps / BATCH * 1e3;
        printf("msmt log_n=%2d bits=%3d table %.0f MiB built in %.1f ms  %8.3f ms (x%.2f)  scratch %.0f MiB | batch of %d: %7.3f ms/MSM (x%.2f)  %s\n",
               log_n, bits, h2_dev_bases_precompute_bytes(n, digits) / 1048576.0, (p1 - p0) * 1e3, tms, ms / tms,
               sbt / 1048576.0, BATCH, tbms, bms / tbms, same && same_batch ? "EQUAL" : "MISMATCH");
        if (!same) rc = 1;
        H2(h2_dev_bases_forget(d_b));
        CK(hipFree(d_scrt));
        CK(hipFree(d_scrt2));
    }
    CK(hipFree(d_s));
    CK(hipFree(d_b));
    CK(hipFree(d_scr));
    return rc;
}

static int bench_eval(int log_n, int reps) {
    size_t n = (size_t)1 << log_n;
    std::vector<uint64_t> h(n * 4);
    fill_random_fr(h, 5);
    void *d_l, *d_r, *d_o;
    CK(hipMalloc(&d_l, n * 32));
    CK(hipMalloc(&d_r, n * 32));
    CK(hipMalloc(&d_o, n * 32));
    CK(hipMemcpy(d_l, h.data(), n * 32, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_r, h.data(), n * 32, hipMemcpyHostToDevice));
    const char* names[] = {"mul_c", "sum_c", "sum", "mul", "sub", "lctheta", "lcbeta", "addgamma", "constant"};
    int inputs[] = {1, 1, 2, 2, 2, 2, 2, 1, 0};
    uint64_t c[4] = {h[0], h[1], h[2], h[3]};
    SPIN(H2(h2_dev_eval_op(2, d_o, d_l, d_r, 1, -1, n, c, nullptr)));
    for (int op = 0; op < 9; op++) {
        H2(h2_dev_eval_op(op, d_o, d_l, d_r, 1, -1, n, c, nullptr));
        float ms = 0;
        H2(h2_timer_start(nullptr));
        for (int r = 0; r < reps; r++) H2(h2_dev_eval_op(op, d_o, d_l, d_r, 1, -1, n, c, nullptr));
        H2(h2_timer_stop(nullptr, &ms));
        double bytes = 32.0 * (inputs[op] + 1) * n;
        printf("eval %-9s log_n=%2d  %8.3f ms  %.1f GB/s algorithmic\n", names[op], log_n, ms / reps, bytes / (ms / reps * 1e-3) / 1e9);
    }
    CK(hipFree(d_l));
    CK(hipFree(d_r));
    CK(hipFree(d_o));
    return 0;
}

int main(int argc, char** argv) {
    init_stream();
    if (h2_device_count() < 1) {
        printf("no device\n");
        return 1;
    }
    int rc = 0;
    if (argc < 2) {
        rc |= bench_ntt(20, 20);
        rc |= bench_ntt(24, 10);
        rc |= bench_msm(20, 254, 5, 0);
        rc |= bench_eval(24, 10);
        return rc;
    }
    for (int i = 1; i < argc;) {
        std::string cmd = argv[i];
        if (cmd == "ntt" && i + 2 < argc) { rc |= bench_ntt(atoi(argv[i + 1]), atoi(argv[i + 2])); i += 3; }
        else if (cmd == "msm" && i + 3 < argc) { rc |= bench_msm(atoi(argv[i + 1]), atoi(argv[i + 2]), atoi(argv[i + 3]), 0); i += 4; }
        else if (cmd == "msmt" && i + 3 < argc) { rc |= bench_msm(atoi(argv[i + 1]), atoi(argv[i + 2]), atoi(argv[i + 3]), 2); i += 4; }
        else if (cmd == "msmbool" && i + 2 < argc) { rc |= bench_msm(atoi(argv[i + 1]), 254, atoi(argv[i + 2]), 1); i += 3; }
        else if (cmd == "eval" && i + 2 < argc) { rc |= bench_eval(atoi(argv[i + 1]), atoi(argv[i + 2])); i += 3; }
        else { printf("bad args\n"); return 2; }
    }
    return rc;
}
