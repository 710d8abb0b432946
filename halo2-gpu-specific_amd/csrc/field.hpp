// field.hpp -- BN254 Fr / Fq arithmetic for gfx950 (CDNA4), 8 x u32 limbs, Montgomery form.
//
// Replaces (device side) the arithmetic the reference gets from the un-vendored crates
// pairing_bn256 (Fr/Fq, /root/reference/halo2_proofs/src/arithmetic.rs:16-17) and
// ec-gpu-gen's generated FIELD_* CUDA source (/root/reference/halo2_proofs/build.rs:1-11).
//
// In-memory layout is the reference's: 32 B = 4 x u64 LE limbs in Montgomery form with
// R = 2^256 (prover.rs:176,183; helpers.rs:185-194) -- identical bytes to 8 x u32 LE.
//
// The multiplier is a product-scanning (Comba) Montgomery multiplication built on
// v_mad_u64_u32 with its carry-out routed through an SGPR pair into one v_addc_co_u32,
// i.e. 1 multiply-add (half the issue rate of a 32-bit add: 12.4 vs 24.6 lanes/clk/SIMD, profiles/r1_ubench.txt) + at most 1 add-with-carry per 32x32 partial product (136 mads per
// modular multiplication; measured chip ceiling 1.31e11 multiplications/s, profiles/r1_mulbench.txt).  No MFMA: there is no dense contraction here.
#pragma once
#ifdef __HIPCC_RTC__
// hipRTC (the evaluate_h generator, evalh_gen.cpp, compiles this header from memory): the runtime header is built in and
// there is no <stdint.h>
typedef unsigned char uint8_t;
typedef unsigned short uint16_t;
typedef unsigned int uint32_t;
typedef unsigned long uint64_t;
typedef int int32_t;
typedef long int64_t;
#else
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

namespace h2 {

struct FrParams {
    static constexpr uint32_t MOD[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                                        0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t MOD2[8] = {0xe0000002u, 0x87c3eb27u, 0xf372e122u, 0x5067d090u,
                                         0x0302b0bau, 0x70a08b6du, 0xc2634053u, 0x60c89ce5u};  // 2r (the lazy domain of the NTT)
    static constexpr uint32_t NMOD[8] = {0x0fffffffu, 0xbc1e0a6cu, 0x86468f6eu, 0xd7cc17b7u,
                                         0x7e7ea7a2u, 0x47afba49u, 0x1ece5fd6u, 0xcf9bb18du};  // 2^256 - r (fp_mul_const)
    static constexpr uint32_t PINV256[8] = {0xefffffffu, 0xc2e1f593u, 0x4c6911b3u, 0x6586864bu,
                                         0x99062391u, 0xe39a9828u, 0x0d8341b2u, 0x73f82f1du};  // -r^-1 mod 2^256 (fp_const_pair)
    static constexpr uint32_t INV = 0xefffffffu;  // -r^-1 mod 2^32
    static constexpr uint32_t ONE[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                                        0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};  // R mod r
    static constexpr uint32_t RR[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u,
                                       0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};  // R^2 mod r
};

struct FqParams {
    static constexpr uint32_t MOD[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                                        0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t MOD2[8] = {0xb0f9fa8eu, 0x7841182du, 0xd0e3951au, 0x2f02d522u,
                                         0x0302b0bbu, 0x70a08b6du, 0xc2634053u, 0x60c89ce5u};  // 2q
    static constexpr uint32_t NMOD[8] = {0x278302b9u, 0xc3df73e9u, 0x978e3572u, 0x687e956eu,
                                         0x7e7ea7a2u, 0x47afba49u, 0x1ece5fd6u, 0xcf9bb18du};  // 2^256 - q
    static constexpr uint32_t PINV256[8] = {0xe4866389u, 0x87d20782u, 0x1eca6ac9u, 0x9ede7d65u,
                                         0x1833da80u, 0xd8afcbd0u, 0x91888c6bu, 0xf57a22b7u};  // -q^-1 mod 2^256
    static constexpr uint32_t INV = 0xe4866389u;  // -q^-1 mod 2^32
    static constexpr uint32_t ONE[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u,
                                        0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};  // R mod q
    static constexpr uint32_t RR[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u,
                                       0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};  // R^2 mod q
};

template <class P>
struct alignas(16) Fp {
    uint32_t l[8];
};
using Fr = Fp<FrParams>;
using Fq = Fp<FqParams>;

#define H2_DEV __host__ __device__ __forceinline__

// ---- 32-byte vector load/store (two global_load_dwordx4) -------------------------------
template <class P>
H2_DEV Fp<P> fp_load(const Fp<P>* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    Fp<P> r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}
template <class P>
H2_DEV void fp_store(Fp<P>* p, const Fp<P>& v) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

template <class P>
H2_DEV Fp<P> fp_zero() {
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = 0;
    return r;
}
template <class P>
H2_DEV Fp<P> fp_one() {
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = P::ONE[i];
    return r;
}
template <class P>
H2_DEV bool fp_is_zero(const Fp<P>& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.l[i];
    return o == 0;
}
template <class P>
H2_DEV bool fp_eq(const Fp<P>& a, const Fp<P>& b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.l[i] ^ b.l[i];
    return o == 0;
}

// ---- carry-chain primitives (device): explicit SGPR-pair carries so two independent chains can be
// interleaved -- on gfx950 a VALU-written carry cannot feed the very next VALU instruction, so a lone
// chain is padded with s_nop by hipcc while two alternating chains issue back to back.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(H2_PORTABLE_MUL)
#define H2_ASM_CHAINS 1
H2_DEV uint32_t add_co(uint32_t a, uint32_t b, uint64_t& c) {
    uint32_t r;
    asm("v_add_co_u32 %0, %1, %2, %3" : "=v"(r), "=s"(c) : "v"(a), "v"(b));
    return r;
}
H2_DEV uint32_t addc_co(uint32_t a, uint32_t b, uint64_t& c) {
    uint32_t r;
    asm("v_addc_co_u32 %0, %1, %2, %3, %1" : "=v"(r), "+s"(c) : "v"(a), "v"(b));
    return r;
}
H2_DEV uint32_t sub_co(uint32_t a, uint32_t b, uint64_t& c) {
    uint32_t r;
    asm("v_sub_co_u32 %0, %1, %2, %3" : "=v"(r), "=s"(c) : "v"(a), "v"(b));
    return r;
}
H2_DEV uint32_t subb_co(uint32_t a, uint32_t b, uint64_t& c) {
    uint32_t r;
    asm("v_subb_co_u32 %0, %1, %2, %3, %1" : "=v"(r), "+s"(c) : "v"(a), "v"(b));
    return r;
}
// mask ? x : y   (mask = per-lane carry/borrow bits)
H2_DEV uint32_t sel_co(uint32_t y, uint32_t x, uint64_t mask) {
    uint32_t r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(y), "v"(x), "s"(mask));
    return r;
}
#endif

// r = a - p if a >= p else a   (a < 2p)
template <class P>
H2_DEV Fp<P> fp_reduce_once(const Fp<P>& a) {
#ifdef H2_ASM_CHAINS
    Fp<P> t, r;
    uint64_t cb;
    t.l[0] = sub_co(a.l[0], P::MOD[0], cb);
#pragma unroll
    for (int i = 1; i < 8; i++) t.l[i] = subb_co(a.l[i], P::MOD[i], cb);
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = sel_co(t.l[i], a.l[i], cb);  // borrow: a < p, keep a
    return r;
#else
    Fp<P> d;
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t t = (uint64_t)a.l[i] - P::MOD[i] - borrow;
        d.l[i] = (uint32_t)t;
        borrow = (t >> 32) & 1;
    }
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = borrow ? a.l[i] : d.l[i];
    return r;
#endif
}

template <class P>
H2_DEV Fp<P> fp_add(const Fp<P>& a, const Fp<P>& b) {
#ifdef H2_ASM_CHAINS
    // s = a + b (chain A) and t = s - p (chain B, one limb behind); the final borrow of B picks s or t
    Fp<P> s, t, r;
    uint64_t ca, cb;
    s.l[0] = add_co(a.l[0], b.l[0], ca);
    s.l[1] = addc_co(a.l[1], b.l[1], ca);
    t.l[0] = sub_co(s.l[0], P::MOD[0], cb);
#pragma unroll
    for (int i = 2; i < 8; i++) {
        s.l[i] = addc_co(a.l[i], b.l[i], ca);
        t.l[i - 1] = subb_co(s.l[i - 1], P::MOD[i - 1], cb);
    }
    t.l[7] = subb_co(s.l[7], P::MOD[7], cb);
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = sel_co(t.l[i], s.l[i], cb);  // borrow: s < p, keep s
    return r;
#else
    Fp<P> s;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)a.l[i] + b.l[i];
        s.l[i] = (uint32_t)c;
        c >>= 32;
    }
    return fp_reduce_once(s);  // p < 2^254: a + b < 2^255, no carry out of limb 7
#endif
}

template <class P>
H2_DEV Fp<P> fp_sub(const Fp<P>& a, const Fp<P>& b) {
#ifdef H2_ASM_CHAINS
    // d = a - b (chain A) and u = d + p (chain B, one limb behind); the final borrow of A picks u or d
    Fp<P> d, u, r;
    uint64_t ca, cb;
    d.l[0] = sub_co(a.l[0], b.l[0], ca);
    d.l[1] = subb_co(a.l[1], b.l[1], ca);
    u.l[0] = add_co(d.l[0], P::MOD[0], cb);
#pragma unroll
    for (int i = 2; i < 8; i++) {
        d.l[i] = subb_co(a.l[i], b.l[i], ca);
        u.l[i - 1] = addc_co(d.l[i - 1], P::MOD[i - 1], cb);
    }
    u.l[7] = addc_co(d.l[7], P::MOD[7], cb);
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = sel_co(d.l[i], u.l[i], ca);  // borrow: a < b, take d + p
    return r;
#else
    Fp<P> d;
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t t = (uint64_t)a.l[i] - b.l[i] - borrow;
        d.l[i] = (uint32_t)t;
        borrow = (t >> 32) & 1;
    }
    uint32_t mask = borrow ? 0xffffffffu : 0u;
    uint64_t c = 0;
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)d.l[i] + (P::MOD[i] & mask);
        r.l[i] = (uint32_t)c;
        c >>= 32;
    }
    return r;
#endif
}

template <class P>
H2_DEV Fp<P> fp_neg(const Fp<P>& a) {
    return fp_sub(fp_zero<P>(), a);
}
template <class P>
H2_DEV Fp<P> fp_dbl(const Fp<P>& a) {
    return fp_add(a, a);
}

// ---- the lazy domain of the NTT stage loops ------------------------------------------------------------------
// Between the load of a pass and its store nothing needs to be a canonical residue: 4p < 2^256, so values may stay anywhere
// below 4p.  A product by a (canonical) twiddle accepts ANY 256-bit first operand and returns a value below 2p without its
// final conditional subtraction (fp_mul_wide); a butterfly's sum of two values below 2p is a bare 8-limb addition, its
// difference u - t + 2p a subtraction and an addition (both below 4p); only a value that takes the NON-multiplied branch
// of the next butterfly is brought below 2p first (fp_lazy_red2p).  Per radix-2 butterfly that is ~40 instructions next
// to the product instead of 48 + the product's own 16-24 of reduction; the canonical residue comes back once, at the
// pass's store (fp_lazy_canon).  Same field elements as the canonical arithmetic at every point (mod p), hence bit-exact
// results after the final canonicalisation.
// a < 4p  ->  a or a - 2p, below 2p
template <class P>
H2_DEV Fp<P> fp_lazy_red2p(const Fp<P>& a) {
#ifdef H2_ASM_CHAINS
    Fp<P> t, r;
    uint64_t cb;
    t.l[0] = sub_co(a.l[0], P::MOD2[0], cb);
#pragma unroll
    for (int i = 1; i < 8; i++) t.l[i] = subb_co(a.l[i], P::MOD2[i], cb);
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = sel_co(t.l[i], a.l[i], cb);  // borrow: a < 2p, keep a
    return r;
#else
    Fp<P> d;
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t t = (uint64_t)a.l[i] - P::MOD2[i] - borrow;
        d.l[i] = (uint32_t)t;
        borrow = (t >> 32) & 1;
    }
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = borrow ? a.l[i] : d.l[i];
    return r;
#endif
}
// a + b as 256-bit integers (a, b < 2p -> below 4p: no carry out)
template <class P>
H2_DEV Fp<P> fp_lazy_add(const Fp<P>& a, const Fp<P>& b) {
    Fp<P> s;
#ifdef H2_ASM_CHAINS
    uint64_t c;
    s.l[0] = add_co(a.l[0], b.l[0], c);
#pragma unroll
    for (int i = 1; i < 8; i++) s.l[i] = addc_co(a.l[i], b.l[i], c);
#else
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)a.l[i] + b.l[i];
        s.l[i] = (uint32_t)c;
        c >>= 32;
    }
#endif
    return s;
}
// a - b + 2p as 256-bit integers (a, b < 2p -> in (0, 4p)); the two chains run one limb apart
template <class P>
H2_DEV Fp<P> fp_lazy_sub(const Fp<P>& a, const Fp<P>& b) {
    Fp<P> d, u;
#ifdef H2_ASM_CHAINS
    uint64_t ca, cb;
    d.l[0] = sub_co(a.l[0], b.l[0], ca);
    d.l[1] = subb_co(a.l[1], b.l[1], ca);
    u.l[0] = add_co(d.l[0], P::MOD2[0], cb);
#pragma unroll
    for (int i = 2; i < 8; i++) {
        d.l[i] = subb_co(a.l[i], b.l[i], ca);
        u.l[i - 1] = addc_co(d.l[i - 1], P::MOD2[i - 1], cb);
    }
    u.l[7] = addc_co(d.l[7], P::MOD2[7], cb);
#else
    uint64_t borrow = 0, c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t t = (uint64_t)a.l[i] - b.l[i] - borrow;
        d.l[i] = (uint32_t)t;
        borrow = (t >> 32) & 1;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)d.l[i] + P::MOD2[i];
        u.l[i] = (uint32_t)c;
        c >>= 32;
    }
#endif
    return u;  // the wrap of d and the carry out of u cancel: the true value is below 2^256
}
// (a + b) brought below 2p / (a - b) brought into [0, 2p)   (a, b < 2p): fp_add / fp_sub with the modulus 2p
template <class P>
H2_DEV Fp<P> fp_lazy_add_red(const Fp<P>& a, const Fp<P>& b) {
#ifdef H2_ASM_CHAINS
    Fp<P> s, t, r;
    uint64_t ca, cb;
    s.l[0] = add_co(a.l[0], b.l[0], ca);
    s.l[1] = addc_co(a.l[1], b.l[1], ca);
    t.l[0] = sub_co(s.l[0], P::MOD2[0], cb);
#pragma unroll
    for (int i = 2; i < 8; i++) {
        s.l[i] = addc_co(a.l[i], b.l[i], ca);
        t.l[i - 1] = subb_co(s.l[i - 1], P::MOD2[i - 1], cb);
    }
    t.l[7] = subb_co(s.l[7], P::MOD2[7], cb);
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = sel_co(t.l[i], s.l[i], cb);  // borrow: s < 2p, keep s
    return r;
#else
    return fp_lazy_red2p(fp_lazy_add(a, b));
#endif
}
template <class P>
H2_DEV Fp<P> fp_lazy_sub_red(const Fp<P>& a, const Fp<P>& b) {
#ifdef H2_ASM_CHAINS
    Fp<P> d, u, r;
    uint64_t ca, cb;
    d.l[0] = sub_co(a.l[0], b.l[0], ca);
    d.l[1] = subb_co(a.l[1], b.l[1], ca);
    u.l[0] = add_co(d.l[0], P::MOD2[0], cb);
#pragma unroll
    for (int i = 2; i < 8; i++) {
        d.l[i] = subb_co(a.l[i], b.l[i], ca);
        u.l[i - 1] = addc_co(d.l[i - 1], P::MOD2[i - 1], cb);
    }
    u.l[7] = addc_co(d.l[7], P::MOD2[7], cb);
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = sel_co(d.l[i], u.l[i], ca);  // borrow: a < b, take d + 2p
    return r;
#else
    return fp_lazy_red2p(fp_lazy_sub(a, b));
#endif
}
// a < 4p -> the canonical residue
template <class P>
H2_DEV Fp<P> fp_lazy_canon(const Fp<P>& a) {
    return fp_reduce_once(fp_lazy_red2p(a));
}

// acc(96 bit = lo64 : hi32) += a * b.
// v_mad_u64_u32 D, carry(SGPR pair), a, b, D ; v_addc_co_u32 hi, carry, hi, 0, carry
H2_DEV void mad_acc(uint64_t& lo, uint32_t& hi, uint32_t a, uint32_t b) {
#if defined(H2_PORTABLE_MUL) || !defined(__HIP_DEVICE_COMPILE__)
    uint64_t p = (uint64_t)a * b;
    uint64_t s = lo + p;
    hi += (s < p) ? 1u : 0u;
    lo = s;
#else
    uint64_t carry;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(lo), "=s"(carry) : "v"(a), "v"(b));
    asm("v_addc_co_u32 %0, %1, %0, 0, %1" : "+v"(hi), "+s"(carry));
#endif
}

// ---- the device multiplier: a generated straight-line schedule per field (tools/gen_fp_mul.py -> fp_mul_gen.hpp) that
// drops the add-with-carry after every multiply-add whose column sum provably still fits 64 bits.
template <class P>
__device__ __forceinline__ Fp<P> fp_mul_dev(const Fp<P>& a, const Fp<P>& b);
template <class P>
__device__ __forceinline__ Fp<P> fp_sqr_dev(const Fp<P>& a);  // a * a with 36 operand products instead of 64
template <class P>
__device__ __forceinline__ Fp<P> fp_mul2_dev(const Fp<P>& a, const Fp<P>& b, const Fp<P>& c, const Fp<P>& d);  // a b + c d, ONE reduction
template <class P>
__device__ __forceinline__ Fp<P> fp_mul_wide_dev(const Fp<P>& a, const Fp<P>& b);  // a < 2^256, b canonical -> < 2p, unreduced
template <class P>
__device__ __forceinline__ Fp<P> fp_mul_const_dev(const Fp<P>& a, const Fp<P>& w, const Fp<P>& wq);  // a < 2^256; see fp_mul_const
#if defined(__HIP_DEVICE_COMPILE__) && !defined(H2_PORTABLE_MUL)
#include "fp_mul_gen.hpp"
#endif

// Montgomery product a*b*R^-1 mod p, product scanning (FIPS) form.
template <class P>
H2_DEV Fp<P> fp_mul(const Fp<P>& a, const Fp<P>& b) {
#if !defined(__HIP_DEVICE_COMPILE__)
    // Host pass: the same Montgomery product on 4 x u64 limbs with 128-bit intermediates (CIOS) -- the host tails of
    // the MSM (window Horner, dominant-scalar multiplication) and the setup arithmetic run several times faster than
    // through the 32-bit product-scanning form below, which is shaped for the GPU's v_mad_u64_u32.
    {
        typedef unsigned __int128 u128;
        uint64_t x[4], y[4], p[4];
        for (int i = 0; i < 4; i++) {
            x[i] = (uint64_t)a.l[2 * i] | ((uint64_t)a.l[2 * i + 1] << 32);
            y[i] = (uint64_t)b.l[2 * i] | ((uint64_t)b.l[2 * i + 1] << 32);
            p[i] = (uint64_t)P::MOD[2 * i] | ((uint64_t)P::MOD[2 * i + 1] << 32);
        }
        uint64_t inv = p[0];                                   // p^-1 mod 2^64 by Newton (p odd), then negated
        for (int i = 0; i < 6; i++) inv *= 2 - p[0] * inv;
        inv = 0 - inv;
        uint64_t t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; i++) {
            u128 c = 0;
            for (int j = 0; j < 4; j++) {
                c += (u128)t[j] + (u128)x[j] * y[i];
                t[j] = (uint64_t)c;
                c >>= 64;
            }
            c += t[4];
            t[4] = (uint64_t)c;
            t[5] = (uint64_t)(c >> 64);
            const uint64_t m = t[0] * inv;
            c = ((u128)t[0] + (u128)m * p[0]) >> 64;
            for (int j = 1; j < 4; j++) {
                c += (u128)t[j] + (u128)m * p[j];
                t[j - 1] = (uint64_t)c;
                c >>= 64;
            }
            c += t[4];
            t[3] = (uint64_t)c;
            t[4] = t[5] + (uint64_t)(c >> 64);
        }
        Fp<P> r;
        for (int i = 0; i < 4; i++) {
            r.l[2 * i] = (uint32_t)t[i];
            r.l[2 * i + 1] = (uint32_t)(t[i] >> 32);
        }
        return fp_reduce_once(r);   // inputs < p => t < 2p < 2^255: t[4] == 0
    }
#elif !defined(H2_PORTABLE_MUL)
    return fp_mul_dev(a, b);
#endif
    Fp<P> r;
    uint64_t lo = 0;
    uint32_t hi = 0;
    uint32_t m[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
#pragma unroll
        for (int j = 0; j <= i; j++) mad_acc(lo, hi, a.l[j], b.l[i - j]);
#pragma unroll
        for (int j = 0; j < i; j++) mad_acc(lo, hi, m[j], P::MOD[i - j]);
        m[i] = (uint32_t)lo * P::INV;
        mad_acc(lo, hi, m[i], P::MOD[0]);  // low word becomes 0
        lo = (lo >> 32) | ((uint64_t)hi << 32);
        hi = 0;
    }
#pragma unroll
    for (int i = 8; i < 16; i++) {
#pragma unroll
        for (int j = i - 7; j < 8; j++) mad_acc(lo, hi, a.l[j], b.l[i - j]);
#pragma unroll
        for (int j = i - 7; j < 8; j++) mad_acc(lo, hi, m[j], P::MOD[i - j]);
        r.l[i - 8] = (uint32_t)lo;
        lo = (lo >> 32) | ((uint64_t)hi << 32);
        hi = 0;
    }
    // a, b < p  =>  result < 2p < 2^255, so the word above r.l[7] is zero here
    return fp_reduce_once(r);
}

template <class P>
H2_DEV Fp<P> fp_sqr(const Fp<P>& a) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(H2_PORTABLE_MUL) && !defined(H2_NO_SQR)
    return fp_sqr_dev(a);
#else
    return fp_mul(a, a);
#endif
}

// canonical integer <-> Montgomery (`batch_mont` / `batch_unmont`, arithmetic.rs:235-241,280-286)
// a * b + c * d (Montgomery form) with ONE reduction on the device: 128 operand products and 64 reduction products in a
// single column scan instead of 2 x (64 + 64) -- the `R (Q - X3) - Y1 PPP` of a point addition, with -Y1 for c.  Inputs
// below 2^254 as everywhere; the result is the canonical residue, identical to fp_add(fp_mul(a, b), fp_mul(c, d)).
template <class P>
H2_DEV Fp<P> fp_mul2(const Fp<P>& a, const Fp<P>& b, const Fp<P>& c, const Fp<P>& d) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(H2_PORTABLE_MUL) && !defined(H2_NO_MUL2)
    return fp_mul2_dev(a, b, c, d);
#else
    return fp_add(fp_mul(a, b), fp_mul(c, d));
#endif
}

// a * b / 2^256 mod p for ANY 256-bit a and a canonical b, as a value below 2p (not reduced further): the product of the
// lazy domain above.  Host / portable path: the generic column scan (its carries are complete) without the final step.
template <class P>
H2_DEV Fp<P> fp_mul_wide(const Fp<P>& a, const Fp<P>& b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(H2_PORTABLE_MUL)
    return fp_mul_wide_dev(a, b);
#else
    Fp<P> r;
    uint64_t lo = 0;
    uint32_t hi = 0;
    uint32_t m[8];
    for (int i = 0; i < 8; i++) {
        for (int j = 0; j <= i; j++) mad_acc(lo, hi, a.l[j], b.l[i - j]);
        for (int j = 0; j < i; j++) mad_acc(lo, hi, m[j], P::MOD[i - j]);
        m[i] = (uint32_t)lo * P::INV;
        mad_acc(lo, hi, m[i], P::MOD[0]);
        lo = (lo >> 32) | ((uint64_t)hi << 32);
        hi = 0;
    }
    for (int i = 8; i < 16; i++) {
        for (int j = i - 7; j < 8; j++) mad_acc(lo, hi, a.l[j], b.l[i - j]);
        for (int j = i - 7; j < 8; j++) mad_acc(lo, hi, m[j], P::MOD[i - j]);
        r.l[i - 8] = (uint32_t)lo;
        lo = (lo >> 32) | ((uint64_t)hi << 32);
        hi = 0;
    }
    return r;
#endif
}

// x * w for a CONSTANT w tabulated with its quotient (the twiddle factors of the NTT): w < p PLAIN (not Montgomery), wq =
// floor(w 2^256 / p), x ANY 256-bit value.  q = floor(x wq / 2^256) summed over the anti-diagonals >= 6 only (exact or one
// short), result x w - q p mod 2^256: 43 + 36 + 36 = 115 multiply-adds and no m = t n' steps (tools/gen_fp_mul.py
// schedule_const has the bounds).  Below 2p like fp_mul_wide's result, and congruent to x w: for x = X R (Montgomery data) that
// is (X w) R -- the data never leave Montgomery form.  The short quotient leaves a value in [2p, 2p + 2^-29 p): seen in the
// top limb, taken back by one subtraction of 2p (about once in 2^29 products).
template <class P>
H2_DEV Fp<P> fp_mul_const(const Fp<P>& x, const Fp<P>& w, const Fp<P>& wq) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(H2_PORTABLE_MUL)
    Fp<P> r = fp_mul_const_dev(x, w, wq);
    if (__builtin_expect(r.l[7] >= P::MOD2[7], 0)) r = fp_lazy_red2p(r);
    return r;
#else
    // portable form: the exact quotient (all 64 products of x wq), the same residue below 2p
    uint32_t q[8];
    {
        uint64_t lo = 0;
        uint32_t hi = 0;
        for (int i = 0; i < 15; i++) {
            for (int j = (i > 7 ? i - 7 : 0); j <= (i < 7 ? i : 7); j++) mad_acc(lo, hi, x.l[j], wq.l[i - j]);
            if (i >= 8) q[i - 8] = (uint32_t)lo;
            lo = (lo >> 32) | ((uint64_t)hi << 32);
            hi = 0;
        }
        q[7] = (uint32_t)lo;
    }
    Fp<P> r;
    uint64_t lo = 0;
    uint32_t hi = 0;
    for (int i = 0; i < 8; i++) {
        for (int j = 0; j <= i; j++) {
            mad_acc(lo, hi, x.l[j], w.l[i - j]);
            mad_acc(lo, hi, q[j], P::NMOD[i - j]);
        }
        r.l[i] = (uint32_t)lo;
        lo = (lo >> 32) | ((uint64_t)hi << 32);
        hi = 0;
    }
    return r;
#endif
}
// w (Montgomery form, as the table kernels compute their powers) -> the pair fp_mul_const reads: the plain value and
// floor(w_plain 2^256 / p).  w_plain 2^256 = wq p + w_mont exactly (w_mont = w_plain 2^256 mod p), so wq = -w_mont / p mod 2^256:
// one low-half product by the constant -p^-1 mod 2^256 (PINV256), no division.
template <class P>
H2_DEV void fp_const_pair(const Fp<P>& w_mont, Fp<P>& w_plain, Fp<P>& wq) {
    w_plain = fp_from_mont(w_mont);
    uint64_t lo = 0;
    uint32_t hi = 0;
    for (int i = 0; i < 8; i++) {
        for (int j = 0; j <= i; j++) mad_acc(lo, hi, w_mont.l[j], P::PINV256[i - j]);
        wq.l[i] = (uint32_t)lo;
        lo = (lo >> 32) | ((uint64_t)hi << 32);
        hi = 0;
    }
}

// The two conversions take RAW caller data (h2_batch_mont, the scalars of an MSM, point encodings): the wide-operand
// product is correct for ANY 256-bit first operand -- a non-canonical input comes out as its residue -- where the
// standard schedule's carry analysis assumes operands below 2^254.
template <class P>
H2_DEV Fp<P> fp_to_mont(const Fp<P>& canon) {
    Fp<P> rr;
#pragma unroll
    for (int i = 0; i < 8; i++) rr.l[i] = P::RR[i];
    return fp_reduce_once(fp_mul_wide(canon, rr));
}
template <class P>
H2_DEV Fp<P> fp_from_mont(const Fp<P>& a) {
    Fp<P> one = fp_zero<P>();
    one.l[0] = 1;
    return fp_reduce_once(fp_mul_wide(a, one));
}

// 1 / a in Montgomery form (a R -> a^-1 R; 0 -> 0) by Kaliski's almost-Montgomery inverse: a binary extended GCD on
// (u, v) = (p, a R) with cofactors (r, s) -- u s + v r = p throughout -- that ends after n <= k <= 2 n halvings (n = 254)
// with a R's inverse times 2^k, followed by 512 - k modular doublings (a^-1 R^-1 2^k * 2^(512 - k) = a^-1 R).  ~360
// rounds of 8-limb shifts / adds / subtractions instead of the 254 squarings + 127 products of a^(p-2).  The four cases
// of a round are branches: for a wave whose lanes hold the SAME value (k_batch_invert's shared inversion) that is 105 us
// against 230 us; lanes with different values serialise the cases (287 us) -- per-lane inversions keep a^(p-2).
template <class P>
H2_DEV Fp<P> fp_inv(const Fp<P>& a) {
    uint32_t u[8], v[8], r[8], s[8];
    uint32_t nz = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u[i] = P::MOD[i];
        v[i] = a.l[i];
        r[i] = 0;
        s[i] = i == 0 ? 1u : 0u;
        nz |= a.l[i];
    }
    if (nz == 0) return a;
    uint32_t k = 0;
    auto halve = [](uint32_t* x) {
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = (x[i] >> 1) | (i < 7 ? x[i + 1] << 31 : 0u);
    };
    auto dbl = [](uint32_t* x) {
#pragma unroll
        for (int i = 7; i >= 0; i--) x[i] = (x[i] << 1) | (i ? x[i - 1] >> 31 : 0u);
    };
    auto sub = [](uint32_t* x, const uint32_t* y) {  // x -= y
        uint32_t bw = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t d = x[i] - y[i], b1 = x[i] < y[i] ? 1u : 0u, e = d - bw, b2 = d < bw ? 1u : 0u;
            x[i] = e;
            bw = b1 | b2;
        }
    };
    auto add = [](uint32_t* x, const uint32_t* y) {  // x += y
        uint32_t cy = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t t = x[i] + y[i], c1 = t < y[i] ? 1u : 0u, e = t + cy, c2 = e < cy ? 1u : 0u;
            x[i] = e;
            cy = c1 | c2;
        }
    };
    auto greater = [](const uint32_t* x, const uint32_t* y) {  // x > y
#pragma unroll
        for (int i = 7; i >= 0; i--)
            if (x[i] != y[i]) return x[i] > y[i];
        return false;
    };
    for (;;) {
        uint32_t vnz = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) vnz |= v[i];
        if (vnz == 0) break;
        // the four cases as branches: a wave whose lanes hold the same value (the shared inversion of k_batch_invert)
        // runs one of them per round; lanes with different values serialise them
        if ((u[0] & 1) == 0) {
            halve(u);
            dbl(s);
        } else if ((v[0] & 1) == 0) {
            halve(v);
            dbl(r);
        } else if (greater(u, v)) {
            sub(u, v);
            halve(u);
            add(r, s);
            dbl(s);
        } else {
            sub(v, u);
            halve(v);
            add(s, r);
            dbl(r);
        }
        k++;
    }
    // r in [0, 2p): x = p - (r mod p) = (a R)^-1 2^k mod p
    Fp<P> x;
    {
        uint32_t d[8];
        uint64_t bw = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t e = (uint64_t)r[i] - P::MOD[i] - bw;
            d[i] = (uint32_t)e;
            bw = (e >> 32) & 1;
        }
        uint64_t bw2 = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t rr = bw ? r[i] : d[i];  // r mod p
            uint64_t e = (uint64_t)P::MOD[i] - rr - bw2;
            x.l[i] = (uint32_t)e;
            bw2 = (e >> 32) & 1;
        }
    }
    x = fp_reduce_once(x);  // r mod p = 0 cannot happen for a != 0, but p - 0 = p must not escape
    for (uint32_t i = k; i < 512; i++) x = fp_dbl(x);
    return x;
}

// a^e for a 32-bit exponent (square-and-multiply from the top set bit)
template <class P>
H2_DEV Fp<P> fp_pow_u32(const Fp<P>& a, uint32_t e) {
    if (e == 0) return fp_one<P>();
    Fp<P> acc = a;
    for (int i = 30 - __builtin_clz(e); i >= 0; i--) {
        acc = fp_sqr(acc);
        if ((e >> i) & 1) acc = fp_mul(acc, a);
    }
    return acc;
}

}  // namespace h2
