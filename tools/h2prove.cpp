// tools/h2prove.cpp -- keygen + create_proof (SHPLONK) of the mini-PLONK circuit over the C ABI of libhalo2_hip.so ALONE: no
// Python, no torch, no HIP headers -- plain C++17 (g++), linked against the library and nothing else.  Device memory, streams
// and copies come from h2_dev_alloc / h2_stream_create / h2_dev_upload / h2_dev_download; every vector operation is an h2_dev_*
// call on device-resident data; the host keeps what the reference's host keeps: the Blake2b transcript, the challenges, the
// handful of scalars SHPLONK adjusts.  It issues the call sequence `integration/hip_resident.rs` (ResidentProver) describes --
// the one halo2-gpu-specific_amd/prover.py runs through torch's allocator -- so the proof bytes must be the ones prover.py
// makes and tests/golden/proof_hash_kat.json pins (tests/test_gpu_h2prove.py).  A tool and a test: not a new ABI entry.
//
// Reference orchestration restated: plonk/keygen.rs:330-440 (keygen_pk), plonk/permutation/keygen.rs:112-261,
// plonk/prover.rs:206-850 (create_proof), plonk/permutation/prover.rs:47-330, plonk/vanishing/prover.rs:40-160,
// poly/multiopen/shplonk.rs:58-135 + shplonk/prover.rs:89-225, transcript.rs:81-215 (Blake2bWrite, Challenge255),
// poly/commitment.rs:56-124 (Params::unsafe_setup); the circuit is examples/simple-example-2.rs:177-288.
//
//   h2prove <k> <seed> [--out proof.bin] [--reps N] [--no-tables] [--entropy]   prove on device 0, print seconds and the proof's hex
//                                        (--entropy: blinding from the operating system, as the reference's OsRng; the seeded mode is for tests)
//   h2prove --host-check <k> <seed>                                    the host-side pieces only (no GPU): printed for the CPU test
#include <algorithm>
#include <array>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <numeric>
#include <set>
#include <string>
#include <vector>

#include "../include/halo2_hip.h"

typedef unsigned __int128 u128;
typedef uint64_t u64;

static void die(const char* what) {
    fprintf(stderr, "h2prove: %s: %s\n", what, h2_last_error());
    exit(1);
}
#define CK(call)                                \
    do {                                        \
        if ((call) != 0) die(#call);            \
    } while (0)

// ------------------------------------------------------------------------------------------------ 256-bit prime fields
struct U256 {
    u64 l[4];
    bool operator==(const U256& o) const { return !memcmp(l, o.l, 32); }
    bool operator!=(const U256& o) const { return !(*this == o); }
    bool operator<(const U256& o) const {
        for (int i = 3; i >= 0; i--)
            if (l[i] != o.l[i]) return l[i] < o.l[i];
        return false;
    }
    bool is_zero() const { return !(l[0] | l[1] | l[2] | l[3]); }
};
static U256 u256_from_hex(const char* h) {
    U256 r{{0, 0, 0, 0}};
    for (const char* p = h; *p; p++) {
        int d = (*p >= '0' && *p <= '9') ? *p - '0' : (*p | 32) - 'a' + 10;
        for (int i = 3; i > 0; i--) r.l[i] = (r.l[i] << 4) | (r.l[i - 1] >> 60);
        r.l[0] = (r.l[0] << 4) | (u64)d;
    }
    return r;
}
static bool geq(const U256& a, const U256& b) { return !(a < b); }
static U256 sub_raw(const U256& a, const U256& b) {
    U256 r;
    u64 bw = 0;
    for (int i = 0; i < 4; i++) {
        u128 t = (u128)a.l[i] - b.l[i] - bw;
        r.l[i] = (u64)t;
        bw = (u64)(t >> 64) & 1;
    }
    return r;
}

// A field element in Montgomery form (R = 2^256): the in-memory representation of the reference's Fr / Fq and of every
// vector on the device.
struct Field {
    U256 p, one, rr;  // modulus, R mod p, R^2 mod p
    u64 inv;          // -p^-1 mod 2^64
    explicit Field(const char* hex) {
        p = u256_from_hex(hex);
        u64 x = p.l[0];
        for (int i = 0; i < 6; i++) x *= 2 - p.l[0] * x;
        inv = 0 - x;
        U256 t{{1, 0, 0, 0}};  // 2^512 mod p by doubling
        for (int i = 0; i < 512; i++) {
            t = dbl_raw(t);
            if (i == 255) one = t;
        }
        rr = t;
    }
    U256 dbl_raw(const U256& a) const {  // 2a mod p for a < p
        U256 r;
        u64 c = 0;
        for (int i = 0; i < 4; i++) {
            r.l[i] = (a.l[i] << 1) | c;
            c = a.l[i] >> 63;
        }
        return (c || geq(r, p)) ? sub_raw(r, p) : r;
    }
    U256 add(const U256& a, const U256& b) const {
        U256 r;
        u64 c = 0;
        for (int i = 0; i < 4; i++) {
            u128 t = (u128)a.l[i] + b.l[i] + c;
            r.l[i] = (u64)t;
            c = (u64)(t >> 64);
        }
        return (c || geq(r, p)) ? sub_raw(r, p) : r;
    }
    U256 sub(const U256& a, const U256& b) const {
        if (geq(a, b)) return sub_raw(a, b);
        U256 t = sub_raw(p, b);
        return add(a, t);
    }
    U256 neg(const U256& a) const { return a.is_zero() ? a : sub_raw(p, a); }
    // a * b / R mod p (CIOS); a may be ANY 256-bit value when b < p
    U256 mul(const U256& a, const U256& b) const {
        u64 t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; i++) {
            u128 c = 0;
            for (int j = 0; j < 4; j++) {
                c += (u128)t[j] + (u128)a.l[j] * b.l[i];
                t[j] = (u64)c;
                c >>= 64;
            }
            c += t[4];
            t[4] = (u64)c;
            t[5] = (u64)(c >> 64);
            const u64 m = t[0] * inv;
            c = ((u128)t[0] + (u128)m * p.l[0]) >> 64;
            for (int j = 1; j < 4; j++) {
                c += (u128)t[j] + (u128)m * p.l[j];
                t[j - 1] = (u64)c;
                c >>= 64;
            }
            c += t[4];
            t[3] = (u64)c;
            t[4] = t[5] + (u64)(c >> 64);
        }
        U256 r{{t[0], t[1], t[2], t[3]}};
        return (t[4] || geq(r, p)) ? sub_raw(r, p) : r;
    }
    U256 sqr(const U256& a) const { return mul(a, a); }
    U256 to_mont(const U256& canon) const { return mul(canon, rr); }  // any 256-bit value -> its residue, Montgomery form
    U256 from_mont(const U256& m) const { return mul(m, U256{{1, 0, 0, 0}}); }
    U256 from_u64(u64 v) const { return to_mont(U256{{v, 0, 0, 0}}); }
    U256 pow(const U256& a, const U256& e) const {
        U256 acc = one;
        for (int i = 255; i >= 0; i--) {
            acc = sqr(acc);
            if ((e.l[i / 64] >> (i % 64)) & 1) acc = mul(acc, a);
        }
        return acc;
    }
    U256 pow_u64(const U256& a, u64 e) const { return pow(a, U256{{e, 0, 0, 0}}); }
    U256 invert(const U256& a) const { return pow(a, sub_raw(p, U256{{2, 0, 0, 0}})); }
    // 64 little-endian bytes -> the 512-bit integer mod p (`from_bytes_wide`), Montgomery form
    U256 from_wide(const uint8_t b[64]) const {
        U256 lo, hi;
        memcpy(lo.l, b, 32);
        memcpy(hi.l, b + 32, 32);
        return add(to_mont(lo), mul(to_mont(hi), rr));  // lo + hi * 2^256
    }
    void to_bytes(const U256& m, uint8_t out[32]) const {
        U256 c = from_mont(m);
        memcpy(out, c.l, 32);
    }
};
static const Field FR("30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001");
static const Field FQ("30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47");
static U256 fr_hex(const char* h) { return FR.to_mont(u256_from_hex(h)); }

// ------------------------------------------------------------------------------------------------ Blake2b (RFC 7693)
struct Blake2b {
    u64 h[8], t = 0;
    uint8_t buf[128];
    size_t fill = 0;
    static u64 rotr(u64 x, int n) { return (x >> n) | (x << (64 - n)); }
    Blake2b(size_t outlen, const char person[16]) {
        static const u64 IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                  0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
        memcpy(h, IV, sizeof h);
        h[0] ^= 0x01010000ULL ^ (u64)outlen;
        u64 pw[2];
        memcpy(pw, person, 16);
        h[6] ^= pw[0];
        h[7] ^= pw[1];
    }
    void compress(const uint8_t* block, bool last) {
        static const u64 IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                  0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
        static const uint8_t S[12][16] = {
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
            {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
            {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
            {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
            {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
        u64 m[16], v[16];
        memcpy(m, block, 128);
        for (int i = 0; i < 8; i++) {
            v[i] = h[i];
            v[i + 8] = IV[i];
        }
        v[12] ^= t;
        if (last) v[14] = ~v[14];
        auto G = [&](int a, int b, int c, int d, u64 x, u64 y) {
            v[a] = v[a] + v[b] + x; v[d] = rotr(v[d] ^ v[a], 32);
            v[c] = v[c] + v[d];     v[b] = rotr(v[b] ^ v[c], 24);
            v[a] = v[a] + v[b] + y; v[d] = rotr(v[d] ^ v[a], 16);
            v[c] = v[c] + v[d];     v[b] = rotr(v[b] ^ v[c], 63);
        };
        for (int r = 0; r < 12; r++) {
            const uint8_t* s = S[r];
            G(0, 4, 8, 12, m[s[0]], m[s[1]]);   G(1, 5, 9, 13, m[s[2]], m[s[3]]);
            G(2, 6, 10, 14, m[s[4]], m[s[5]]);  G(3, 7, 11, 15, m[s[6]], m[s[7]]);
            G(0, 5, 10, 15, m[s[8]], m[s[9]]);  G(1, 6, 11, 12, m[s[10]], m[s[11]]);
            G(2, 7, 8, 13, m[s[12]], m[s[13]]); G(3, 4, 9, 14, m[s[14]], m[s[15]]);
        }
        for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
    }
    void update(const void* data, size_t len) {
        const uint8_t* p = (const uint8_t*)data;
        while (len) {
            if (fill == 128) {  // a full buffer is compressed only when more input follows (the last block is special)
                t += 128;
                compress(buf, false);
                fill = 0;
            }
            size_t take = std::min(len, (size_t)128 - fill);
            memcpy(buf + fill, p, take);
            fill += take;
            p += take;
            len -= take;
        }
    }
    void digest(uint8_t out[64]) const {  // of a copy: the running state goes on (transcript.rs squeeze)
        Blake2b c = *this;
        c.t += c.fill;
        memset(c.buf + c.fill, 0, 128 - c.fill);
        c.compress(c.buf, true);
        memcpy(out, c.h, 64);
    }
};

// ------------------------------------------------------------------------------------------------ points (host side)
struct Affine {
    U256 x, y;  // Montgomery Fq; identity = (0, 0)
};
// 12 u64 Jacobian (X, Y, Z Montgomery) -> affine Montgomery
static Affine jac_to_affine(const u64 xyz[12]) {
    U256 X, Y, Z;
    memcpy(X.l, xyz, 32);
    memcpy(Y.l, xyz + 4, 32);
    memcpy(Z.l, xyz + 8, 32);
    if (Z.is_zero()) return Affine{{{0, 0, 0, 0}}, {{0, 0, 0, 0}}};
    U256 zi = FQ.invert(Z), zi2 = FQ.sqr(zi);
    return Affine{FQ.mul(X, zi2), FQ.mul(FQ.mul(Y, zi2), zi)};
}
static void point_to_bytes(const Affine& P, uint8_t out[32]) {  // x little-endian, bit 7 of byte 31 = parity of y
    if (P.x.is_zero() && P.y.is_zero()) {
        memset(out, 0, 32);
        return;
    }
    FQ.to_bytes(P.x, out);
    out[31] |= (uint8_t)((FQ.from_mont(P.y).l[0] & 1) << 7);
}

// ------------------------------------------------------------------------------------------------ transcript.rs:152-226
struct Transcript {
    Blake2b state{64, "Halo2-Transcript"};
    std::vector<uint8_t> writer;
    U256 squeeze() {  // Challenge255: prefix 0, digest of a copy, from_bytes_wide
        const uint8_t pre = 0;
        state.update(&pre, 1);
        uint8_t d[64];
        state.digest(d);
        return FR.from_wide(d);
    }
    void common_point(const Affine& P) {
        const uint8_t pre = 1;
        uint8_t b[64];
        FQ.to_bytes(P.x, b);
        FQ.to_bytes(P.y, b + 32);
        state.update(&pre, 1);
        state.update(b, 64);
    }
    void common_scalar(const U256& v) {
        const uint8_t pre = 2;
        uint8_t b[32];
        FR.to_bytes(v, b);
        state.update(&pre, 1);
        state.update(b, 32);
    }
    void write_point(const Affine& P) {
        common_point(P);
        uint8_t b[32];
        point_to_bytes(P, b);
        writer.insert(writer.end(), b, b + 32);
    }
    void write_scalar(const U256& v) {
        common_scalar(v);
        uint8_t b[32];
        FR.to_bytes(v, b);
        writer.insert(writer.end(), b, b + 32);
    }
};

// ------------------------------------------------------------------------------------------------ the test-mode randomness
// rng.ProverRng.deterministic: xoshiro256** seeded through splitmix64; the random polynomial's ChaCha20 key from a second stream
struct ProverRng {
    u64 s[4], seed;
    bool secure = false;  // --entropy: every draw from the operating system (`OsRng`, as the reference): the proof hides its witness
    static u64 rotl(u64 x, int k) { return (x << k) | (x >> (64 - k)); }
    static void os_entropy(void* out, size_t bytes) {
        FILE* f = fopen("/dev/urandom", "rb");
        if (!f || fread(out, 1, bytes, f) != bytes) {
            fprintf(stderr, "h2prove: no entropy source\n");
            exit(1);
        }
        fclose(f);
    }
    static ProverRng from_os() {
        ProverRng r(0);
        r.secure = true;
        return r;
    }
    explicit ProverRng(u64 sd) : seed(sd) {
        u64 z0 = sd;
        for (int i = 0; i < 4; i++) {
            z0 += 0x9E3779B97F4A7C15ULL;
            u64 z = z0;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
            s[i] = z ^ (z >> 31);
        }
    }
    u64 next() {
        if (secure) {
            u64 v;
            os_entropy(&v, 8);
            return v;
        }
        u64 out = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0];
        s[3] ^= s[1];
        s[1] ^= s[2];
        s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return out;
    }
    u64 u16() { return next() & 0xFFFF; }
    U256 fr() {  // Fr::random: 512 random bits mod r
        u64 w[8];
        for (int i = 0; i < 8; i++) w[i] = next();
        return FR.from_wide((const uint8_t*)w);
    }
    void poly_key(uint8_t key[32]) const {
        if (secure) return os_entropy(key, 32);
        ProverRng second(seed ^ 0x706F6C795F6B6579ULL);
        u64 w[4];
        for (int i = 0; i < 4; i++) w[i] = second.next();
        memcpy(key, w, 32);
    }
};

// ------------------------------------------------------------------------------------------------ the circuit (fixed)
// examples/simple-example-2.rs:177-288 as halo2-gpu-specific_amd/circuits.py mini_plonk() states it: advice a, b, c with
// equality, fixed sm, sa, sb, sc, one gate a sa + b sb + a b sm - c sc.  The evaluator program below is what
// circuit.compile_evaluator makes of it and CS_STORE what formats.cs_store serialises (the verifying key's digest covers it):
// `h2prove --host-check` prints both and tests/test_h2prove_host.py compares them with the Python side.
static const uint32_t BLINDING_FACTORS = 5, DEGREE = 3, N_ADVICE = 3, N_FIXED = 4;
static const char* CONSTANTS_HEX[4] = {"0", "1", "2", "30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000000"};
static const int32_t ROTATIONS[1] = {0};
static const h2_calculation CALCS[9] = {
    {H2_CALC_MUL, {H2_VS_ADVICE, 0, 0}, {H2_VS_FIXED, 1, 0}, 0, 0},       {H2_CALC_MUL, {H2_VS_ADVICE, 1, 0}, {H2_VS_FIXED, 2, 0}, 0, 0},
    {H2_CALC_ADD, {H2_VS_INTERMEDIATE, 0, 0}, {H2_VS_INTERMEDIATE, 1, 0}, 0, 0}, {H2_CALC_MUL, {H2_VS_ADVICE, 0, 0}, {H2_VS_ADVICE, 1, 0}, 0, 0},
    {H2_CALC_MUL, {H2_VS_INTERMEDIATE, 3, 0}, {H2_VS_FIXED, 0, 0}, 0, 0}, {H2_CALC_ADD, {H2_VS_INTERMEDIATE, 2, 0}, {H2_VS_INTERMEDIATE, 4, 0}, 0, 0},
    {H2_CALC_MUL, {H2_VS_ADVICE, 2, 0}, {H2_VS_FIXED, 3, 0}, 0, 0},       {H2_CALC_MUL, {H2_VS_INTERMEDIATE, 6, 0}, {H2_VS_CONSTANT, 3, 0}, 0, 0},
    {H2_CALC_ADD, {H2_VS_INTERMEDIATE, 5, 0}, {H2_VS_INTERMEDIATE, 7, 0}, 0, 0}};
static const h2_value_source VALUE_PARTS[1] = {{H2_VS_INTERMEDIATE, 8, 0}};
static const int ADVICE_QUERIES[3][2] = {{0, 0}, {1, 0}, {2, 0}};          // (column, rotation), in query order
static const int FIXED_QUERIES[4][2] = {{1, 0}, {2, 0}, {3, 0}, {0, 0}};
static const char* CS_STORE_HEX =
    "03000000000000000000000004000000030000000100000001000000010000000000000000000000030000000000000000000000010000000000000002"
    "00000000000000000000000400000001000000000000000200000000000000030000000000000000000000000000000300000000000000000000000100"
    "00000000000002000000000000000000000000000000000000000000000001000000010000000500000005000000050000000600000002000000000000"
    "00000000000000000001000000000000000100000000000000060000000200000001000000010000000000000001000000010000000200000000000000"
    "06000000060000000200000000000000000000000000000002000000010000000100000000000000010000000300000000000000000000000700000006"
    "0000000200000002000000020000000000000001000000020000000300000000000000000000f093f5e1439170b97948e833285d588181b64550b829a0"
    "31e1724e643007000000000000000000000000000000010000000100000000000000010000000000000000000000020000000100000000000000000000"
    "000100000000000000020000000000000000000000030000000100000000000000";

struct Domain {  // EvaluationDomain::new, poly/domain.rs:44-149 (the scalars; Montgomery form)
    uint32_t k, ek;
    size_t n, en;
    U256 omega, omega_inv, ext_omega, ext_omega_inv, ifft_div, ext_ifft_div, g_coset, g_coset_inv;
    std::vector<U256> t_evals;
    explicit Domain(uint32_t k_) : k(k_) {
        const U256 root = fr_hex("03ddb9f5166d18b798865ea93dd31f743215cf6dd39329c8d34f1ed960c37c9c");
        const U256 zeta = fr_hex("30644e72e131a029048b6e193fd84104cc37a73fec2bc5e9b8ca0b2d36636f23");
        n = (size_t)1 << k;
        ek = k;
        while (((size_t)1 << ek) < n * (DEGREE - 1)) ek++;
        en = (size_t)1 << ek;
        ext_omega = root;
        for (uint32_t i = ek; i < 28; i++) ext_omega = FR.sqr(ext_omega);
        omega = ext_omega;
        for (uint32_t i = k; i < ek; i++) omega = FR.sqr(omega);
        omega_inv = FR.invert(omega);
        ext_omega_inv = FR.invert(ext_omega);
        ifft_div = FR.invert(FR.from_u64(n));
        ext_ifft_div = FR.invert(FR.from_u64(en));
        g_coset = zeta;
        g_coset_inv = FR.sqr(zeta);
        const U256 zn = FR.pow_u64(zeta, n), wn = FR.pow_u64(ext_omega, n);
        U256 cur = zn;  // zeta^n * w^(n i) - 1, inverted
        for (size_t i = 0; i < (en >> k); i++) {
            t_evals.push_back(FR.invert(FR.sub(cur, FR.one)));
            cur = FR.mul(cur, wn);
        }
    }
    U256 rotate(const U256& x, int rot) const {
        return FR.mul(x, FR.pow_u64(rot >= 0 ? omega : omega_inv, (u64)(rot >= 0 ? rot : -rot)));
    }
};

// synthesize (:224-254) as circuits.mini_plonk_synthesize: canonical (n, 4) u64 columns and the copy constraints
struct Witness {
    std::vector<u64*> advice;                     // [col][4 n], in PAGE-LOCKED host memory when `pinned` (DMA to the device)
    std::vector<std::vector<u64>> advice_store;   // ... or here (the host-check mode, which never touches the device)
    std::vector<std::vector<u64>> fixed;          // [col][4 n]
    std::vector<std::array<u64, 4>> copies;       // (left column, left row, right column, right row)
    bool pinned = false;
    Witness() = default;
    Witness(const Witness&) = delete;
    Witness& operator=(const Witness&) = delete;
    ~Witness() {
        if (pinned)
            for (u64* p : advice) h2_host_free_pinned(p);
    }
};
static void synthesize(Witness& w, uint32_t k, bool pinned, u64 a = 5) {
    const size_t n = (size_t)1 << k, pairs = (size_t)1 << (k - 4);
    w.pinned = pinned;
    for (int c = 0; c < 3; c++) {
        if (pinned) {
            void* p = nullptr;
            CK(h2_host_alloc_pinned(32 * n, &p));
            memset(p, 0, 32 * n);
            w.advice.push_back((u64*)p);
        } else {
            w.advice_store.emplace_back(4 * n, 0);
        }
    }
    if (!pinned)
        for (auto& v : w.advice_store) w.advice.push_back(v.data());
    w.fixed.assign(4, std::vector<u64>(4 * n, 0));
    const u64 a2 = a * a;
    for (size_t i = 0; i < pairs; i++) {
        const size_t r0 = 2 * i, r1 = r0 + 1;
        w.advice[0][4 * r0] = a; w.advice[1][4 * r0] = a; w.advice[2][4 * r0] = a2;
        w.fixed[0][4 * r0] = 1; w.fixed[3][4 * r0] = 1;                                   // sm, sc
        w.advice[0][4 * r1] = a; w.advice[1][4 * r1] = a2; w.advice[2][4 * r1] = a + a2;
        w.fixed[1][4 * r1] = 1; w.fixed[2][4 * r1] = 1; w.fixed[3][4 * r1] = 1;           // sa, sb, sc
    }
    for (size_t i = 0; i < pairs; i++) w.copies.push_back({0, 2 * i, 0, 2 * i + 1});
    for (size_t i = 0; i < pairs; i++) w.copies.push_back({1, 2 * i + 1, 2, 2 * i});
}

// plonk/permutation/keygen.rs:112-143: every cycle sorted by (column, row), each cell pointing at its successor
static void permutation_mapping(size_t ncols, size_t n, const std::vector<std::array<u64, 4>>& copies,
                                std::vector<std::vector<uint32_t>>& map_col, std::vector<std::vector<uint32_t>>& map_row) {
    map_col.assign(ncols, std::vector<uint32_t>(n));
    map_row.assign(ncols, std::vector<uint32_t>(n));
    for (size_t c = 0; c < ncols; c++)
        for (size_t r = 0; r < n; r++) {
            map_col[c][r] = (uint32_t)c;
            map_row[c][r] = (uint32_t)r;
        }
    std::vector<u64> nodes;
    for (auto& cp : copies) {
        nodes.push_back(cp[0] * n + cp[1]);
        nodes.push_back(cp[2] * n + cp[3]);
    }
    std::sort(nodes.begin(), nodes.end());
    nodes.erase(std::unique(nodes.begin(), nodes.end()), nodes.end());
    std::vector<uint32_t> parent(nodes.size());
    std::iota(parent.begin(), parent.end(), 0u);
    auto find = [&](uint32_t x) {
        while (parent[x] != x) x = parent[x] = parent[parent[x]];
        return x;
    };
    auto at = [&](u64 id) { return (uint32_t)(std::lower_bound(nodes.begin(), nodes.end(), id) - nodes.begin()); };
    for (auto& cp : copies) {
        uint32_t a = find(at(cp[0] * n + cp[1])), b = find(at(cp[2] * n + cp[3]));
        if (a != b) parent[a] = b;
    }
    std::map<uint32_t, std::vector<u64>> cycles;  // root -> members, ascending (nodes is sorted)
    for (uint32_t i = 0; i < nodes.size(); i++) cycles[find(i)].push_back(nodes[i]);
    for (auto& kv : cycles) {
        const std::vector<u64>& m = kv.second;
        for (size_t i = 0; i < m.size(); i++) {
            const u64 from = m[i], to = m[(i + 1) % m.size()];
            map_col[from / n][from % n] = (uint32_t)(to / n);
            map_row[from / n][from % n] = (uint32_t)(to % n);
        }
    }
}

static std::vector<uint8_t> unhex(const char* h) {
    std::vector<uint8_t> out;
    for (size_t i = 0; h[i] && h[i + 1]; i += 2) {
        auto d = [](char c) { return (c >= '0' && c <= '9') ? c - '0' : (c | 32) - 'a' + 10; };
        out.push_back((uint8_t)(d(h[i]) << 4 | d(h[i + 1])));
    }
    return out;
}
static std::string hex(const uint8_t* p, size_t n) {
    std::string s;
    char b[3];
    for (size_t i = 0; i < n; i++) {
        snprintf(b, sizeof b, "%02x", p[i]);
        s += b;
    }
    return s;
}
static std::string fr_str(const U256& m) {  // canonical value, big-endian hex (as Python's hex())
    uint8_t b[32];
    FR.to_bytes(m, b);
    std::reverse(b, b + 32);
    return hex(b, 32);
}

// VerifyingKey::hash_into as prover.vk_digest restates it (the Debug text of the reference cannot be reproduced: DESIGN 4)
static U256 vk_digest(const Domain& dom, const std::vector<Affine>& fixed_c, const std::vector<Affine>& perm_c) {
    std::vector<uint8_t> body;
    auto put = [&](const void* p, size_t len) { body.insert(body.end(), (const uint8_t*)p, (const uint8_t*)p + len); };
    auto put32 = [&](uint32_t v) { put(&v, 4); };
    put("halo2-hip-vk-v2", 15);
    put32(dom.k);
    put32(dom.ek);
    uint8_t b[32];
    FR.to_bytes(dom.omega, b);
    put(b, 32);
    put(FR.p.l, 32);
    put(FQ.p.l, 32);
    const std::vector<uint8_t> cs = unhex(CS_STORE_HEX);
    put32((uint32_t)cs.size());
    put(cs.data(), cs.size());
    for (const std::vector<Affine>* group : {&fixed_c, &perm_c}) {
        put32((uint32_t)group->size());
        for (const Affine& P : *group) {
            point_to_bytes(P, b);
            put(b, 32);
        }
    }
    Blake2b h(64, "Halo2-Verify-Key");
    const u64 len = body.size();
    h.update(&len, 8);
    h.update(body.data(), body.size());
    uint8_t d[64];
    h.digest(d);
    return FR.from_wide(d);
}

// ------------------------------------------------------------------------------------------------ device helpers
static void* g_stream = nullptr;
static void* g_copy_stream = nullptr;
// Device blocks are recycled by size (a proof allocates the same few sizes again and again, and hipMalloc / hipFree cost more
// than the kernels between them at small k): everything this tool launches is ordered on ONE stream -- and the library's MSM /
// evaluator calls return with their own streams drained -- so a block handed out again is only ever touched behind its last use.
static std::multimap<size_t, void*> g_pool;
static void* pool_get(size_t bytes) {
    auto it = g_pool.find(bytes);
    if (it != g_pool.end()) {
        void* p = it->second;
        g_pool.erase(it);
        return p;
    }
    void* p = nullptr;
    CK(h2_dev_alloc(bytes, &p));
    return p;
}
static void pool_drain() {
    for (auto& kv : g_pool) h2_dev_free(kv.second);
    g_pool.clear();
}
struct DVec {  // n elements on the device (32 bytes each unless said otherwise)
    void* p = nullptr;
    size_t n = 0, bytes = 0;
    DVec() = default;
    explicit DVec(size_t n_, size_t elt = 32) : p(pool_get(n_ * elt)), n(n_), bytes(n_ * elt) {}
    DVec(DVec&& o) noexcept : p(o.p), n(o.n), bytes(o.bytes) { o.p = nullptr; }
    DVec& operator=(DVec&& o) noexcept {
        if (this != &o) {
            if (p) g_pool.emplace(bytes, p);
            p = o.p;
            n = o.n;
            bytes = o.bytes;
            o.p = nullptr;
        }
        return *this;
    }
    DVec(const DVec&) = delete;
    DVec& operator=(const DVec&) = delete;
    ~DVec() {
        if (p) g_pool.emplace(bytes, p);
    }
    void* at(size_t i) const { return (char*)p + 32 * i; }
};
static void upload(const DVec& d, size_t first, const void* src, size_t count) { CK(h2_dev_upload(d.at(first), src, 32 * count, g_stream)); }
static void set_rows(const DVec& d, size_t first, const std::vector<U256>& vals) {
    if (vals.empty()) return;
    upload(d, first, vals.data(), vals.size());
    CK(h2_stream_synchronize(g_stream));  // (the source is a temporary)
}
static std::vector<U256> get_rows(const DVec& d, size_t first, size_t count) {
    std::vector<U256> out(count);
    CK(h2_dev_download(out.data(), d.at(first), 32 * count, g_stream));
    return out;
}
static DVec clone(const DVec& s) {
    DVec d(s.n);
    CK(h2_dev_eval_op(H2_OP_SUM_C, d.p, s.p, nullptr, 0, 0, s.n, U256{{0, 0, 0, 0}}.l, g_stream));  // d = s + 0
    return d;
}
static DVec g_scratch;
static void* scratch(size_t bytes) {
    if (g_scratch.bytes < bytes) g_scratch = DVec((bytes + 31) / 32);
    return g_scratch.p;
}
static Affine msm(const DVec& scalars, const void* bases, size_t n, uint32_t bits = 254) {
    const size_t sb = h2_msm_scratch_bytes(n, bits);
    u64 out[12];
    CK(h2_dev_msm(scalars.p, bases, n, bits, scratch(sb), sb, out, g_stream));
    return jac_to_affine(out);
}
// one MSM per column over the same bases, pipelined inside the library (plonk/prover.rs:293-299, :477-487: the per-column loops)
static std::vector<Affine> msm_batch(const std::vector<const DVec*>& cols, const void* bases, size_t n, const std::vector<uint32_t>& bits) {
    const size_t count = cols.size();
    if (count == 1) return {msm(*cols[0], bases, n, bits[0])};
    size_t sb = 0;
    for (uint32_t b : bits) sb = std::max(sb, std::max(2 * ((h2_msm_scratch_bytes(n, b) + 255) / 256 * 256), h2_msm_batch_scratch_bytes(n, b, count)));
    std::vector<const void*> sp, bp(count, bases);
    for (const DVec* c : cols) sp.push_back(c->p);
    std::vector<u64> out(12 * count);
    CK(h2_dev_msm_batch_ex(sp.data(), bp.data(), bits.data(), count, n, scratch(sb), sb, out.data(), g_stream));
    std::vector<Affine> pts;
    for (size_t i = 0; i < count; i++) pts.push_back(jac_to_affine(&out[12 * i]));
    return pts;
}
static void intt(const DVec& t, const Domain& dom) {
    DVec tmp(dom.n);
    CK(h2_dev_intt(t.p, tmp.p, dom.omega_inv.l, dom.ifft_div.l, dom.k, g_stream));
}
static DVec to_extended(const DVec& coeffs, const Domain& dom) {
    DVec out(dom.en), tmp(dom.en);
    CK(h2_dev_coeff_to_extended(coeffs.p, out.p, tmp.p, dom.k, dom.ek, dom.g_coset.l, dom.g_coset_inv.l, dom.ext_omega.l, g_stream));
    return out;
}
static void eval_op(int op, const DVec& res, const DVec* l, const DVec* r, const U256* c, size_t size) {
    CK(h2_dev_eval_op(op, res.p, l ? l->p : nullptr, r ? r->p : nullptr, 0, 0, size, c ? c->l : nullptr, g_stream));
}
static DVec lincomb(const std::vector<const DVec*>& polys, const std::vector<U256>& coeffs, size_t n) {
    DVec res(n);
    std::vector<const void*> ptrs;
    for (const DVec* p : polys) ptrs.push_back(p->p);
    CK(h2_dev_lincomb(res.p, ptrs.data(), (const u64*)coeffs.data(), polys.size(), n, g_stream));
    return res;
}
static void sub_low(const DVec& t, const std::vector<U256>& low) {  // t[i] -= low[i] for the first few coefficients
    if (low.empty()) return;
    std::vector<U256> cur = get_rows(t, 0, low.size());
    for (size_t i = 0; i < low.size(); i++) cur[i] = FR.sub(cur[i], low[i]);
    set_rows(t, 0, cur);
}
static void kate_division(const DVec& a, size_t n, const U256& b, const DVec& out) {  // out[n - 1] = 0 (shplonk/prover.rs:112)
    CK(h2_dev_kate_division(a.p, n, b.l, out.p, g_stream));
    set_rows(out, n - 1, {U256{{0, 0, 0, 0}}});
}
static std::vector<U256> eval_batch(const std::vector<const DVec*>& polys, size_t n, const std::vector<U256>& points) {
    std::vector<const void*> ptrs;
    for (const DVec* p : polys) ptrs.push_back(p->p);
    std::vector<U256> out(polys.size());
    CK(h2_dev_eval_polynomial_batch(ptrs.data(), polys.size(), n, (const u64*)points.data(), (u64*)out.data(), g_stream));
    return out;
}

// ------------------------------------------------------------------------------------------------ Params::unsafe_setup
struct Params {
    uint32_t k;
    DVec g, g_lagrange;  // n x 64 B affine Montgomery
    bool tables = false;
    ~Params() {
        if (tables) {
            h2_dev_bases_forget(g.p);
            h2_dev_bases_forget(g_lagrange.p);
        }
    }
};
static void unsafe_setup(Params& P, uint32_t k, const U256& s, const Domain& dom, bool tables) {
    const size_t n = dom.n;
    P.k = k;
    // [2^j] G for j < 254, affine (host: 254 doublings)
    std::vector<Affine> pow2;
    Affine Pt{FQ.from_u64(1), FQ.from_u64(2)};
    const U256 three = FQ.from_u64(3), two = FQ.from_u64(2);
    for (int j = 0; j < 254; j++) {
        pow2.push_back(Pt);
        U256 lam = FQ.mul(FQ.mul(three, FQ.sqr(Pt.x)), FQ.invert(FQ.mul(two, Pt.y)));
        U256 x3 = FQ.sub(FQ.sqr(lam), FQ.mul(two, Pt.x));
        Pt = Affine{x3, FQ.sub(FQ.mul(lam, FQ.sub(Pt.x, x3)), Pt.y)};
    }
    DVec table(254, 64);
    CK(h2_dev_upload(table.p, pow2.data(), 64 * 254, g_stream));
    auto powers = [&](const U256& base) {  // [base^i]: the running product of a constant column
        DVec f(n), out(n);
        eval_op(H2_OP_CONSTANT, f, nullptr, nullptr, &base, n);
        CK(h2_dev_prefix_product(f.p, n, FR.one.l, out.p, g_stream));
        CK(h2_stream_synchronize(g_stream));
        return out;
    };
    auto fixed_base = [&](const DVec& scalars) {
        DVec out(n, 64);
        CK(h2_dev_fixed_base_mul(scalars.p, table.p, n, out.p, g_stream));
        CK(h2_stream_synchronize(g_stream));
        return out;
    };
    {
        DVec sp = powers(s);
        P.g = fixed_base(sp);
    }
    DVec w = powers(dom.omega), t(n), tmp(n);
    const U256 neg_s = FR.neg(s);
    eval_op(H2_OP_SUM_C, t, &w, nullptr, &neg_s, n);  // w^i - s
    CK(h2_dev_batch_invert(t.p, tmp.p, n, g_stream));
    eval_op(H2_OP_MUL, t, &t, &w, nullptr, n);        // w^i / (w^i - s)
    const U256 mult = FR.mul(FR.sub(FR.pow_u64(s, n), FR.one), dom.ifft_div), neg_mult = FR.neg(mult);
    eval_op(H2_OP_MUL_C, t, &t, nullptr, &neg_mult, n);  // (s^n - 1) / n * w^i / (s - w^i)
    P.g_lagrange = fixed_base(t);
    if (tables && n >= ((size_t)1 << 15)) {  // shifted-base tables of both point sets (as prover.Params does)
        CK(h2_dev_bases_precompute(P.g_lagrange.p, n, 0, g_stream));
        CK(h2_dev_bases_precompute(P.g.p, n, 0, g_stream));
        P.tables = true;
    }
    CK(h2_stream_synchronize(g_stream));
}

// ------------------------------------------------------------------------------------------------ keygen
struct ProvingKey {
    std::vector<DVec> fixed_values, fixed_polys, fixed_cosets, sigma_values, sigma_polys, sigma_cosets;
    DVec l0, l_last, l_active_row, t_evals;
    std::vector<Affine> fixed_commitments, perm_commitments;
    U256 transcript_repr;
};
static void keygen(ProvingKey& pk, const Params& P, const Domain& dom, const Witness& w) {
    const size_t n = dom.n, en = dom.en;
    for (const auto& col : w.fixed) {
        DVec t(n);
        upload(t, 0, col.data(), n);
        CK(h2_dev_batch_mont(t.p, n, g_stream));
        pk.fixed_commitments.push_back(msm(t, P.g_lagrange.p, n));
        DVec poly = clone(t);
        intt(poly, dom);
        pk.fixed_cosets.push_back(to_extended(poly, dom));
        pk.fixed_polys.push_back(std::move(poly));
        pk.fixed_values.push_back(std::move(t));
    }
    std::vector<std::vector<uint32_t>> mc, mr;
    permutation_mapping(N_ADVICE, n, w.copies, mc, mr);
    const U256 delta = fr_hex("09226b6e22c6f0ca64ec26aad4c86e715b5f898e5e963f25870e56bbe533e9a2");
    for (size_t i = 0; i < N_ADVICE; i++) {
        DVec out(n), dc(n, 4), dr(n, 4);
        CK(h2_dev_upload(dc.p, mc[i].data(), 4 * n, g_stream));
        CK(h2_dev_upload(dr.p, mr[i].data(), 4 * n, g_stream));
        CK(h2_dev_permutation_sigma(out.p, dc.p, dr.p, n, delta.l, dom.omega.l, g_stream));
        CK(h2_stream_synchronize(g_stream));
        pk.perm_commitments.push_back(msm(out, P.g_lagrange.p, n));
        DVec poly = clone(out);
        intt(poly, dom);
        pk.sigma_cosets.push_back(to_extended(poly, dom));
        pk.sigma_polys.push_back(std::move(poly));
        pk.sigma_values.push_back(std::move(out));
    }
    auto lagrange_poly = [&](size_t first, size_t count) {  // coefficient form of the indicator of rows [first, first + count)
        DVec t(n);
        const U256 zero{{0, 0, 0, 0}};
        eval_op(H2_OP_CONSTANT, t, nullptr, nullptr, &zero, n);
        set_rows(t, first, std::vector<U256>(count, FR.one));
        intt(t, dom);
        return t;
    };
    DVec l0p = lagrange_poly(0, 1), llp = lagrange_poly(n - BLINDING_FACTORS - 1, 1), lbp = lagrange_poly(n - BLINDING_FACTORS, BLINDING_FACTORS);
    pk.l0 = to_extended(l0p, dom);
    pk.l_last = to_extended(llp, dom);
    DVec lb = to_extended(lbp, dom), tmp(en);
    eval_op(H2_OP_SUM, tmp, &pk.l_last, &lb, nullptr, en);
    pk.l_active_row = DVec(en);
    eval_op(H2_OP_CONSTANT, pk.l_active_row, nullptr, nullptr, &FR.one, en);
    eval_op(H2_OP_SUB, pk.l_active_row, &pk.l_active_row, &tmp, nullptr, en);  // 1 - (l_last + l_blind)
    pk.t_evals = DVec(dom.t_evals.size());
    set_rows(pk.t_evals, 0, dom.t_evals);
    pk.transcript_repr = vk_digest(dom, pk.fixed_commitments, pk.perm_commitments);
    CK(h2_stream_synchronize(g_stream));
}

// ------------------------------------------------------------------------------------------------ multiopen (SHPLONK)
struct Query {
    int key;  // index into `polys`
    int rot;
    U256 point, eval;
};
static std::vector<U256> lagrange_interpolate(const std::vector<U256>& pts, const std::vector<U256>& evals) {  // arithmetic.rs:849-903
    const size_t n = pts.size();
    std::vector<U256> out(n, U256{{0, 0, 0, 0}});
    for (size_t j = 0; j < n; j++) {
        std::vector<U256> num{FR.one};
        U256 den = FR.one;
        for (size_t m = 0; m < n; m++) {
            if (m == j) continue;
            std::vector<U256> next(num.size() + 1);
            for (size_t i = 0; i < next.size(); i++) {
                U256 hi = i ? num[i - 1] : U256{{0, 0, 0, 0}}, lo = i < num.size() ? num[i] : U256{{0, 0, 0, 0}};
                next[i] = FR.sub(hi, FR.mul(pts[m], lo));
            }
            num = next;
            den = FR.mul(den, FR.sub(pts[j], pts[m]));
        }
        const U256 c = FR.mul(evals[j], FR.invert(den));
        for (size_t i = 0; i < n; i++) out[i] = FR.add(out[i], FR.mul(c, num[i]));
    }
    return out;
}
static U256 horner(const std::vector<U256>& c, const U256& x) {
    U256 acc{{0, 0, 0, 0}};
    for (size_t i = c.size(); i-- > 0;) acc = FR.add(FR.mul(acc, x), c[i]);
    return acc;
}
static void shplonk(Transcript& tr, const Params& P, const std::vector<Query>& queries, const std::vector<const DVec*>& polys, size_t n) {
    const U256 y = tr.squeeze();
    // construct_intermediate_sets (shplonk.rs:58-135): BTreeMap / BTreeSet orders = sorted
    std::map<int, U256> rot_point;
    for (const Query& q : queries) rot_point.emplace(q.rot, q.point);
    std::vector<U256> super_points;
    for (auto& kv : rot_point) super_points.push_back(kv.second);
    std::vector<int> order;
    std::map<int, std::set<int>> rotsets;
    for (const Query& q : queries) {
        if (!rotsets.count(q.key)) order.push_back(q.key);
        rotsets[q.key].insert(q.rot);
    }
    std::map<std::vector<int>, std::vector<int>> groups;  // sorted rotation set -> polynomials, in first-query order
    for (int key : order) groups[std::vector<int>(rotsets[key].begin(), rotsets[key].end())].push_back(key);
    std::map<std::pair<int, int>, U256> evals;
    for (const Query& q : queries) evals[{q.key, q.rot}] = q.eval;
    struct RotSet {
        std::vector<int> rots, keys;
        std::vector<U256> points;
        std::vector<std::vector<U256>> low;  // r_i(X) per polynomial
    };
    std::vector<RotSet> sets;
    for (auto& kv : groups) {
        RotSet rs;
        rs.rots = kv.first;
        rs.keys = kv.second;
        for (int r : rs.rots) rs.points.push_back(rot_point[r]);
        for (int key : rs.keys) {
            std::vector<U256> e;
            for (int r : rs.rots) e.push_back(evals[{key, r}]);
            rs.low.push_back(lagrange_interpolate(rs.points, e));
        }
        sets.push_back(rs);
    }
    const U256 v = tr.squeeze();
    auto powers_desc = [](const U256& c, size_t m) {  // c^(m-1), ..., c, 1
        std::vector<U256> out(m, FR.one);
        for (size_t i = m - 1; i-- > 0;) out[i] = FR.mul(out[i + 1], c);
        return out;
    };
    std::vector<DVec> quotients;
    DVec ping(n), pong(n);
    for (const RotSet& rs : sets) {
        const size_t m = rs.keys.size(), width = rs.points.size();
        const std::vector<U256> ypow = powers_desc(y, m);
        std::vector<const DVec*> ps;
        for (int key : rs.keys) ps.push_back(polys[key]);
        DVec n_x = lincomb(ps, ypow, n);
        std::vector<U256> low(width, U256{{0, 0, 0, 0}});
        for (size_t j = 0; j < width; j++)
            for (size_t i = 0; i < m; i++) low[j] = FR.add(low[j], FR.mul(ypow[i], rs.low[i][j]));
        sub_low(n_x, low);
        const DVec* cur = &n_x;
        for (const U256& pt : rs.points) {
            const DVec* nxt = cur != &ping ? &ping : &pong;
            kate_division(*cur, n, pt, *nxt);
            cur = nxt;
        }
        quotients.push_back(clone(*cur));
    }
    const size_t R = sets.size();
    const std::vector<U256> vpow = powers_desc(v, R);
    std::vector<const DVec*> qs;
    for (DVec& q : quotients) qs.push_back(&q);
    DVec h_x = lincomb(qs, vpow, n);
    tr.write_point(msm(h_x, P.g.p, n));
    const U256 u = tr.squeeze();
    auto vanishing = [&](const std::vector<U256>& roots) {
        U256 acc = FR.one;
        for (const U256& r : roots) acc = FR.mul(acc, FR.sub(u, r));
        return acc;
    };
    const U256 zt_eval = vanishing(super_points);
    std::vector<U256> z_diffs;
    for (const RotSet& rs : sets) {
        std::vector<U256> others;
        for (const U256& p : super_points)
            if (std::find(rs.points.begin(), rs.points.end(), p) == rs.points.end()) others.push_back(p);
        z_diffs.push_back(vanishing(others));
    }
    const U256 scale = FR.invert(z_diffs[0]);
    std::vector<const DVec*> lin_polys;
    std::vector<U256> lin_coeffs;
    U256 constant{{0, 0, 0, 0}};
    for (size_t r = 0; r < R; r++) {
        const RotSet& rs = sets[r];
        const size_t m = rs.keys.size();
        const std::vector<U256> ypow = powers_desc(y, m);
        for (size_t i = 0; i < m; i++) {
            const U256 c = FR.mul(FR.mul(FR.mul(vpow[r], z_diffs[r]), ypow[i]), scale);
            lin_polys.push_back(polys[rs.keys[i]]);
            lin_coeffs.push_back(c);
            constant = FR.add(constant, FR.mul(c, horner(rs.low[i], u)));
        }
    }
    lin_polys.push_back(&h_x);
    lin_coeffs.push_back(FR.neg(FR.mul(zt_eval, scale)));
    DVec l_x = lincomb(lin_polys, lin_coeffs, n);
    sub_low(l_x, {constant});
    if (!eval_batch({&l_x}, n, {u})[0].is_zero()) {  // the reference's must_be_zero (shplonk/prover.rs:204-207)
        fprintf(stderr, "h2prove: shplonk: l(u) != 0\n");
        exit(1);
    }
    kate_division(l_x, n, u, pong);
    tr.write_point(msm(pong, P.g.p, n));
}

// ------------------------------------------------------------------------------------------------ create_proof
static std::vector<uint8_t> create_proof(const Params& P, const ProvingKey& pk, const Domain& dom, const Witness& w, u64 seed, bool entropy = false) {
    const size_t n = dom.n, en = dom.en, bf = BLINDING_FACTORS, usable = n - (bf + 1);
    const int last_rot = -(int)(bf + 1);
    ProverRng rng = entropy ? ProverRng::from_os() : ProverRng(seed);
    Transcript tr;
    tr.common_scalar(pk.transcript_repr);
    // the witness columns start crossing PCIe now, on a stream of their own (DMA out of page-locked memory) ...
    std::vector<DVec> advice;
    for (size_t c = 0; c < N_ADVICE; c++) {
        advice.emplace_back(n);
        CK(h2_dev_upload(advice[c].p, w.advice[c], 32 * n, g_copy_stream));
    }
    // ... under the vanishing argument's random polynomial and its commitment (vanishing/prover.rs:40-67), which depend on
    // nothing the transcript has hashed
    uint8_t key[32];
    rng.poly_key(key);
    DVec random_poly(n);
    CK(h2_dev_random_fr(key, n, random_poly.p, g_stream));
    const Affine random_commitment = msm(random_poly, P.g.p, n);
    CK(h2_stream_synchronize(g_copy_stream));  // the columns have arrived
    // advice columns: blinding rows (16-bit values, drawn column by column), bounded commitments (prover.rs:255-312)
    std::vector<std::vector<U256>> blind(N_ADVICE);
    for (size_t c = 0; c < N_ADVICE; c++)
        for (size_t r = usable; r < n; r++) blind[c].push_back(U256{{rng.u16(), 0, 0, 0}});  // canonical for now
    for (size_t c = 0; c < N_ADVICE; c++) set_rows(advice[c], usable, blind[c]);
    {
        const void* cols[N_ADVICE];
        uint32_t bits[N_ADVICE];
        for (size_t c = 0; c < N_ADVICE; c++) cols[c] = advice[c].p;
        DVec words(8 * N_ADVICE, 4);
        CK(h2_dev_max_scalar_bits(cols, N_ADVICE, n, words.p, bits, g_stream));  // find_max_scalar_bits (prover.rs:237-254)
        std::vector<const DVec*> ptrs;
        std::vector<uint32_t> vb;
        for (size_t c = 0; c < N_ADVICE; c++) {
            CK(h2_dev_batch_mont(advice[c].p, n, g_stream));
            ptrs.push_back(&advice[c]);
            vb.push_back(std::max(bits[c], 1u));
        }
        for (const Affine& pt : msm_batch(ptrs, P.g_lagrange.p, n, vb)) tr.write_point(pt);
    }
    (void)tr.squeeze();  // theta (no lookups in this circuit)
    const U256 beta = tr.squeeze(), gamma = tr.squeeze();
    // permutation grand products (permutation/prover.rs:47-165): chunk = degree - 2 = 1 column per set, 3 sets
    const U256 delta = fr_hex("09226b6e22c6f0ca64ec26aad4c86e715b5f898e5e963f25870e56bbe533e9a2");
    const size_t nsets = N_ADVICE;
    DVec nums(nsets * n), inv(nsets * n), inv_tmp(nsets * n);
    U256 dpow = FR.one;
    for (size_t s = 0; s < nsets; s++) {
        CK(h2_dev_permutation_terms(nums.at(s * n), inv.at(s * n), advice[s].p, pk.sigma_values[s].p, n, beta.l, gamma.l, dpow.l,
                                    dom.omega.l, 1, g_stream));
        dpow = FR.mul(dpow, delta);
    }
    CK(h2_dev_batch_invert(inv.p, inv_tmp.p, nsets * n, g_stream));
    eval_op(H2_OP_MUL, nums, &nums, &inv, nullptr, nsets * n);
    std::vector<DVec> z;
    U256 last_z = FR.one;
    for (size_t s = 0; s < nsets; s++) {
        DVec zs(n);
        CK(h2_dev_prefix_product(nums.at(s * n), n, last_z.l, zs.p, g_stream));
        last_z = get_rows(zs, usable, 1)[0];
        std::vector<U256> b;
        for (size_t i = 0; i < bf; i++) b.push_back(rng.fr());
        set_rows(zs, n - bf, b);
        z.push_back(std::move(zs));
    }
    {
        std::vector<const DVec*> ptrs;
        for (DVec& t : z) ptrs.push_back(&t);
        for (const Affine& pt : msm_batch(ptrs, P.g_lagrange.p, n, std::vector<uint32_t>(nsets, 254))) tr.write_point(pt);
    }
    tr.write_point(random_commitment);
    const U256 y = tr.squeeze();
    // h(X): coefficient forms, extended cosets, the fused evaluator (plonk/evaluation.rs:1229-1985)
    for (DVec& t : advice) intt(t, dom);
    for (DVec& t : z) intt(t, dom);
    std::vector<DVec> advice_ext, z_ext;
    for (DVec& t : advice) advice_ext.push_back(to_extended(t, dom));
    for (DVec& t : z) z_ext.push_back(to_extended(t, dom));
    DVec h(en);
    {
        std::vector<U256> constants;
        for (const char* c : CONSTANTS_HEX) constants.push_back(fr_hex(c));
        std::vector<const u64*> fixed, adv, zs, sig;
        for (const DVec& t : pk.fixed_cosets) fixed.push_back((const u64*)t.p);
        for (const DVec& t : advice_ext) adv.push_back((const u64*)t.p);
        for (const DVec& t : z_ext) zs.push_back((const u64*)t.p);
        for (const DVec& t : pk.sigma_cosets) sig.push_back((const u64*)t.p);
        const uint32_t col_type[3] = {H2_ANY_ADVICE, H2_ANY_ADVICE, H2_ANY_ADVICE}, col_index[3] = {0, 1, 2};
        h2_evalh_desc d;
        memset(&d, 0, sizeof d);
        d.k = dom.k; d.extended_k = dom.ek; d.blinding_factors = BLINDING_FACTORS; d.chunk_len = DEGREE - 2;
        d.constants = (const u64*)constants.data(); d.n_constants = 4;
        d.rotations = ROTATIONS; d.n_rotations = 1;
        d.calculations = CALCS; d.n_calculations = 9;
        d.value_parts = VALUE_PARTS; d.n_value_parts = 1;
        d.fixed = fixed.data(); d.n_fixed = N_FIXED;
        d.advice = adv.data(); d.n_advice = N_ADVICE;
        d.l0 = (const u64*)pk.l0.p; d.l_last = (const u64*)pk.l_last.p; d.l_active_row = (const u64*)pk.l_active_row.p;
        d.n_perm_sets = (uint32_t)nsets; d.perm_z = zs.data();
        d.n_perm_columns = N_ADVICE; d.perm_col_type = col_type; d.perm_col_index = col_index; d.perm_sigma = sig.data();
        memcpy(d.y, y.l, 32); memcpy(d.beta, beta.l, 32); memcpy(d.gamma, gamma.l, 32);
        memcpy(d.delta, delta.l, 32); memcpy(d.zeta, dom.g_coset.l, 32); memcpy(d.extended_omega, dom.ext_omega.l, 32);
        CK(h2_dev_evaluate_h(&d, h.p, g_stream));
        CK(h2_stream_synchronize(g_stream));  // (the descriptor's host arrays leave scope)
    }
    advice_ext.clear();
    z_ext.clear();
    // vanishing construct (vanishing/prover.rs:69-112): divide, back to coefficients, n-coefficient pieces
    CK(h2_dev_divide_by_vanishing_poly(h.p, en, pk.t_evals.p, dom.t_evals.size(), g_stream));
    {
        DVec tmp(en);
        CK(h2_dev_extended_to_coeff(h.p, tmp.p, dom.ek, dom.g_coset.l, dom.g_coset_inv.l, dom.ext_omega_inv.l, dom.ext_ifft_div.l, g_stream));
    }
    const size_t npieces = DEGREE - 1;
    std::vector<DVec> pieces;
    for (size_t i = 0; i < npieces; i++) {
        DVec pc(n);
        CK(h2_dev_eval_op(H2_OP_SUM_C, pc.p, h.at(i * n), nullptr, 0, 0, n, U256{{0, 0, 0, 0}}.l, g_stream));
        pieces.push_back(std::move(pc));
    }
    {
        std::vector<const DVec*> ptrs;
        for (DVec& t : pieces) ptrs.push_back(&t);
        for (const Affine& pt : msm_batch(ptrs, P.g.p, n, std::vector<uint32_t>(npieces, 254))) tr.write_point(pt);
    }
    const U256 x = tr.squeeze(), xn = FR.pow_u64(x, n);
    // h(X) = sum_i x^(n i) piece_i (vanishing/prover.rs:120-124)
    std::vector<U256> xpow{FR.one};
    for (size_t i = 1; i < npieces; i++) xpow.push_back(FR.mul(xpow.back(), xn));
    std::vector<const DVec*> pcs;
    for (DVec& pc : pieces) pcs.push_back(&pc);
    DVec h_poly = lincomb(pcs, xpow, n);
    // evaluations (prover.rs:700-790), then the multiopen query list in the reference's order (:792-840)
    std::vector<const DVec*> polys;  // key -> polynomial
    auto key_of = [&](const DVec* p) {
        for (size_t i = 0; i < polys.size(); i++)
            if (polys[i] == p) return (int)i;
        polys.push_back(p);
        return (int)polys.size() - 1;
    };
    std::vector<std::pair<int, int>> wanted, written;  // (key, rotation)
    auto want = [&](const DVec* p, int rot, bool write = true) {
        std::pair<int, int> kr{key_of(p), rot};
        if (std::find(wanted.begin(), wanted.end(), kr) == wanted.end()) wanted.push_back(kr);
        if (write) written.push_back(kr);
    };
    for (auto& q : ADVICE_QUERIES) want(&advice[q[0]], q[1]);
    for (auto& q : FIXED_QUERIES) want(&pk.fixed_polys[q[0]], q[1]);
    want(&random_poly, 0);
    for (const DVec& s : pk.sigma_polys) want(&s, 0);
    for (size_t i = 0; i < z.size(); i++) {
        want(&z[i], 0);
        want(&z[i], 1);
        if (i + 1 < z.size()) want(&z[i], last_rot);
    }
    want(&h_poly, 0, false);  // opened, not written (vanishing/prover.rs:140-155)
    std::vector<const DVec*> ev_polys;
    std::vector<U256> ev_points;
    for (auto& kr : wanted) {
        ev_polys.push_back(polys[kr.first]);
        ev_points.push_back(dom.rotate(x, kr.second));
    }
    const std::vector<U256> values = eval_batch(ev_polys, n, ev_points);
    std::map<std::pair<int, int>, U256> evals;
    for (size_t i = 0; i < wanted.size(); i++) evals[wanted[i]] = values[i];
    for (auto& kr : written) tr.write_scalar(evals[kr]);
    std::vector<Query> queries;
    auto query = [&](const DVec* p, int rot) {
        const int key = key_of(p);
        queries.push_back(Query{key, rot, dom.rotate(x, rot), evals[{key, rot}]});
    };
    for (auto& q : ADVICE_QUERIES) query(&advice[q[0]], q[1]);
    for (size_t i = 0; i < z.size(); i++) {
        query(&z[i], 0);
        query(&z[i], 1);
    }
    for (size_t i = z.size() - 1; i-- > 0;) query(&z[i], last_rot);
    for (auto& q : FIXED_QUERIES) query(&pk.fixed_polys[q[0]], q[1]);
    for (const DVec& s : pk.sigma_polys) query(&s, 0);
    query(&h_poly, 0);
    query(&random_poly, 0);
    shplonk(tr, P, queries, polys, n);
    return tr.writer;
}

// ------------------------------------------------------------------------------------------------ main
static const char* TRAPDOOR = "1d0c5f0a3b7e91c2a4d6f8091b2c3d4e5f60718293a4b5c6d7e8f9010203";  // tests' fixed toxic scalar

static int host_check(uint32_t k, u64 seed) {
    // everything the prover computes on the host, printed for tests/test_h2prove_host.py (no GPU needed)
    Blake2b b(64, "Halo2-Transcript");
    b.update("abc", 3);
    uint8_t d[64];
    b.digest(d);
    printf("blake2b_abc %s\n", hex(d, 64).c_str());
    Blake2b big(64, "Halo2-Verify-Key");
    std::vector<uint8_t> blob(1000);
    for (size_t i = 0; i < blob.size(); i++) blob[i] = (uint8_t)(i * 7 + 3);
    big.update(blob.data(), 128);      // a block boundary, then the rest
    big.update(blob.data() + 128, 872);
    big.digest(d);
    printf("blake2b_1000 %s\n", hex(d, 64).c_str());
    ProverRng rng(seed);
    const u64 d0 = rng.u16(), d1 = rng.u16();  // (in this order: the arguments of a call are not sequenced)
    printf("rng_u16 %llu %llu\n", (unsigned long long)d0, (unsigned long long)d1);
    printf("rng_fr %s\n", fr_str(rng.fr()).c_str());
    uint8_t key[32];
    rng.poly_key(key);
    printf("rng_poly_key %s\n", hex(key, 32).c_str());
    Domain dom(k);
    printf("domain %u %u omega %s ext_omega %s t0 %s tlast %s\n", dom.k, dom.ek, fr_str(dom.omega).c_str(), fr_str(dom.ext_omega).c_str(),
           fr_str(dom.t_evals[0]).c_str(), fr_str(dom.t_evals.back()).c_str());
    Transcript tr;
    tr.common_scalar(FR.from_u64(12345));
    Affine G{FQ.from_u64(1), FQ.from_u64(2)};
    tr.write_point(G);
    const U256 ch1 = tr.squeeze();
    printf("challenge %s writer %s\n", fr_str(ch1).c_str(), hex(tr.writer.data(), tr.writer.size()).c_str());
    printf("challenge2 %s\n", fr_str(tr.squeeze()).c_str());
    printf("vk_digest_of_generators %s\n", fr_str(vk_digest(dom, {G, G}, {G})).c_str());
    Witness w;
    synthesize(w, k, false);
    std::vector<std::vector<uint32_t>> mc, mr;
    permutation_mapping(N_ADVICE, dom.n, w.copies, mc, mr);
    u64 acc = 1469598103934665603ULL;  // FNV-1a over the mapping and the witness
    auto mix = [&](u64 v) { acc = (acc ^ v) * 1099511628211ULL; };
    for (size_t c = 0; c < N_ADVICE; c++)
        for (size_t r = 0; r < dom.n; r++) {
            mix(mc[c][r]);
            mix(mr[c][r]);
        }
    printf("mapping_fnv %016llx\n", (unsigned long long)acc);
    acc = 1469598103934665603ULL;
    for (u64* col : w.advice)
        for (size_t i = 0; i < 4 * dom.n; i++) mix(col[i]);
    for (auto& col : w.fixed)
        for (u64 v : col) mix(v);
    printf("witness_fnv %016llx\n", (unsigned long long)acc);
    printf("cs_store %s\n", CS_STORE_HEX);
    printf("program");
    for (const char* c : CONSTANTS_HEX) printf(" c:%s", c);
    for (const h2_calculation& c : CALCS)
        printf(" calc:%u,%u,%u,%u,%u,%u,%u,%u,%u", c.op, c.a.kind, c.a.index, c.a.rot, c.b.kind, c.b.index, c.b.rot, c.challenge, c.power);
    printf(" vp:%u,%u,%u", VALUE_PARTS[0].kind, VALUE_PARTS[0].index, VALUE_PARTS[0].rot);
    for (auto& q : ADVICE_QUERIES) printf(" aq:%d,%d", q[0], q[1]);
    for (auto& q : FIXED_QUERIES) printf(" fq:%d,%d", q[0], q[1]);
    printf("\n");
    // interpolation: the polynomial through three points, evaluated back
    std::vector<U256> pts{FR.from_u64(3), FR.from_u64(10), rng.fr()}, ev{rng.fr(), rng.fr(), rng.fr()};
    std::vector<U256> poly = lagrange_interpolate(pts, ev);
    printf("interpolate %d\n", (int)(horner(poly, pts[0]) == ev[0] && horner(poly, pts[1]) == ev[1] && horner(poly, pts[2]) == ev[2]));
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 4 && !strcmp(argv[1], "--host-check")) return host_check((uint32_t)atoi(argv[2]), strtoull(argv[3], nullptr, 0));
    if (argc < 3) {
        fprintf(stderr, "usage: h2prove <k> <seed> [--out proof.bin] [--reps N] [--no-tables] [--entropy]   |   h2prove --host-check <k> <seed>\n");
        return 2;
    }
    const uint32_t k = (uint32_t)atoi(argv[1]);
    const u64 seed = strtoull(argv[2], nullptr, 0);
    const char* out_path = nullptr;
    int reps = 1;
    bool tables = true, entropy = false;
    for (int i = 3; i < argc; i++) {
        if (!strcmp(argv[i], "--out") && i + 1 < argc) out_path = argv[++i];
        else if (!strcmp(argv[i], "--reps") && i + 1 < argc) reps = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--no-tables")) tables = false;
        else if (!strcmp(argv[i], "--entropy")) entropy = true;   // blinding from the OS (the seed is then ignored): not reproducible
    }
    if (k < 4 || k > 26) {
        fprintf(stderr, "h2prove: k out of range\n");
        return 2;
    }
    if (h2_device_count() < 1) die("no HIP device");
    CK(h2_set_device(0));
    CK(h2_stream_create(&g_stream));
    CK(h2_stream_create(&g_copy_stream));
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };
    Domain dom(k);
    std::vector<uint8_t> proof;
    double t_setup, t_keygen, t_first, t_best = 1e30;
    {
        Params P;
        auto t0 = now();
        unsafe_setup(P, k, fr_hex(TRAPDOOR), dom, tables);
        t_setup = secs(t0, now());
        Witness w;
        synthesize(w, k, true);
        ProvingKey pk;
        t0 = now();
        keygen(pk, P, dom, w);
        t_keygen = secs(t0, now());
        t0 = now();
        proof = create_proof(P, pk, dom, w, seed, entropy);
        t_first = secs(t0, now());
        for (int r = 1; r < reps; r++) {
            t0 = now();
            std::vector<uint8_t> again = create_proof(P, pk, dom, w, seed, entropy);
            t_best = std::min(t_best, secs(t0, now()));
            if (!entropy && again != proof) {
                fprintf(stderr, "h2prove: two proofs from one seed differ\n");
                return 1;
            }
        }
        printf("vk_digest 0x%s\n", fr_str(pk.transcript_repr).c_str());
        g_scratch = DVec();
    }
    pool_drain();
    if (reps < 2) t_best = t_first;
    printf("k %u seed %llu proof_bytes %zu setup_s %.4f keygen_s %.4f first_proof_s %.4f create_proof_s %.4f generated_launches %llu\n", k,
           (unsigned long long)seed, proof.size(), t_setup, t_keygen, t_first, t_best, (unsigned long long)h2_evalh_generated_launches());
    printf("proof %s\n", hex(proof.data(), proof.size()).c_str());
    if (out_path) {
        FILE* f = fopen(out_path, "wb");
        if (!f || fwrite(proof.data(), 1, proof.size(), f) != proof.size()) {
            fprintf(stderr, "h2prove: cannot write %s\n", out_path);
            return 1;
        }
        fclose(f);
    }
    CK(h2_stream_destroy(g_copy_stream));
    CK(h2_stream_destroy(g_stream));
    return 0;
}
