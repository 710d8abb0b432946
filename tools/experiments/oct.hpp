// oct.hpp -- EXPERIMENT: a 254-bit Montgomery product spread over the eight lanes of an "octet" (lane k holds limb k of every
// operand), for the latency-bound tails of the MSM (k_finish / k_reduce run chains of dependent point additions with few of
// them in flight: what counts there is the latency of ONE product, ~1600 cycles on one lane).
//
// CIOS with the column accumulators moving down one lane per step: at step j lane k adds a_k * b_j and m_j * p_k into the
// accumulator of column k + j; column j is complete in lane 0 at step j (that is where m_j = column * (-p^-1) comes from);
// afterwards every accumulator moves to the lane below (lane 0's upper words are its carry into the next column), lane 7
// starts a fresh column.  After 8 steps the lanes hold columns 8..15, un-normalised; one carry-lookahead pass over the
// octet resolves them and a second one does the conditional subtraction of p.
// Per lane: 16 multiply-adds instead of 136, ~150 instructions instead of ~400; cross-lane traffic is DPP only.
#pragma once
#include "../../halo2-gpu-specific_amd/csrc/field.hpp"

namespace h2 {

#define H2_DPP(old, src, ctrl, rmask, bmask, bound) \
    ((uint32_t)__builtin_amdgcn_update_dpp((int)(old), (int)(src), (ctrl), (rmask), (bmask), (bound)))

// lane J of every octet -> its eight lanes (octets = lanes 8o .. 8o + 7; two per DPP row)
template <int J>
__device__ __forceinline__ uint32_t oct_bcast(uint32_t v) {
    constexpr int q = J & 3;
    uint32_t t = H2_DPP(v, v, q * 0x55, 0xF, 0xF, true);       // every quad: its own lane q
    if (J < 4) return H2_DPP(t, t, 0x114, 0xF, 0xA, false);    // row_shr:4 into banks 1, 3: the upper quad takes the lower one's
    return H2_DPP(t, t, 0x104, 0xF, 0x5, false);               // row_shl:4 into banks 0, 2
}

struct OctCtx {
    uint32_t pk;     // limb (lane & 7) of the modulus
    uint32_t inv;    // -p^-1 mod 2^32
    uint32_t k;      // lane & 7
};

template <class P>
__device__ __forceinline__ OctCtx oct_ctx() {
    OctCtx c;
    c.k = threadIdx.x & 7;
    c.pk = 0;
#pragma unroll
    for (int i = 0; i < 8; i++)
        if (c.k == (uint32_t)i) c.pk = P::MOD[i];
    c.inv = P::INV;
    return c;
}

// lane k <- lane k + d of the same octet (0 beyond it)
template <int D>
__device__ __forceinline__ uint32_t oct_down(uint32_t v, uint32_t k) {
    uint32_t t = H2_DPP(0u, v, 0x100 + D, 0xF, 0xF, true);  // row_shl:D
    return (k + D > 7) ? 0u : t;
}
// lane k <- lane k - d of the same octet (0 below it)
template <int D>
__device__ __forceinline__ uint32_t oct_up(uint32_t v, uint32_t k) {
    uint32_t t = H2_DPP(0u, v, 0x110 + D, 0xF, 0xF, true);  // row_shr:D
    return (k < (uint32_t)D) ? 0u : t;
}

// carry-lookahead over the octet: g = this lane generates a carry, p = this lane propagates one; returns the carry INTO the lane
__device__ __forceinline__ uint32_t oct_carry_in(uint32_t g, uint32_t p, uint32_t k) {
    uint32_t gs = oct_up<1>(g, k), ps = oct_up<1>(p, k);
    g |= p & gs;
    p &= ps;
    gs = oct_up<2>(g, k);
    ps = oct_up<2>(p, k);
    g |= p & gs;
    p &= ps;
    gs = oct_up<4>(g, k);
    g |= p & gs;
    return oct_up<1>(g, k);
}

template <int J>
__device__ __forceinline__ void oct_step(uint32_t a, uint32_t b, const OctCtx& c, uint32_t& w0, uint32_t& w1, uint32_t& w2) {
    const uint32_t bj = oct_bcast<J>(b);
    uint64_t acc = ((uint64_t)w1 << 32) | w0;
    uint64_t t = (uint64_t)a * bj;
    acc += t;
    w2 += acc < t ? 1u : 0u;
    const uint32_t m = oct_bcast<0>((uint32_t)acc * c.inv);
    t = (uint64_t)m * c.pk;
    acc += t;
    w2 += acc < t ? 1u : 0u;
    // lane 0: low word is 0 now; its upper words are the carry into the next column
    const uint32_t lo = (uint32_t)acc, hi = (uint32_t)(acc >> 32);
    const uint32_t add0 = c.k == 0 ? hi : 0u, add1 = c.k == 0 ? w2 : 0u;
    const uint32_t n0 = oct_down<1>(lo, c.k), n1 = oct_down<1>(hi, c.k), n2 = oct_down<1>(w2, c.k);
    uint64_t s = (uint64_t)n0 + add0;
    w0 = (uint32_t)s;
    s = (s >> 32) + n1 + add1;
    w1 = (uint32_t)s;
    w2 = n2 + (uint32_t)(s >> 32);
}

// a * b * 2^-256 mod p; a, b, result: limb (lane & 7) of each
template <class P>
__device__ __forceinline__ uint32_t oct_mul(uint32_t a, uint32_t b, const OctCtx& c) {
    uint32_t w0 = 0, w1 = 0, w2 = 0;
    oct_step<0>(a, b, c, w0, w1, w2);
    oct_step<1>(a, b, c, w0, w1, w2);
    oct_step<2>(a, b, c, w0, w1, w2);
    oct_step<3>(a, b, c, w0, w1, w2);
    oct_step<4>(a, b, c, w0, w1, w2);
    oct_step<5>(a, b, c, w0, w1, w2);
    oct_step<6>(a, b, c, w0, w1, w2);
    oct_step<7>(a, b, c, w0, w1, w2);
    // lanes hold columns 8 .. 15: limb k = w0_k + w1_(k-1) + w2_(k-2) + carries
    uint64_t s = (uint64_t)w0 + oct_up<1>(w1, c.k) + oct_up<2>(w2, c.k);
    uint32_t r = (uint32_t)s;
    uint32_t cin = oct_up<1>((uint32_t)(s >> 32), c.k);  // 0 .. 2 from the lane below
    uint32_t t = r + cin;
    uint32_t g = t < r ? 1u : 0u;
    // ripple the single-bit carries: lane k receives g_(k-1); generate / propagate on t
    const uint32_t y = oct_up<1>(g, c.k);
    const uint32_t z = t + y;
    const uint32_t gen = z < t ? 1u : 0u, prop = z == 0xffffffffu ? 1u : 0u;
    r = z + oct_carry_in(gen, prop, c.k);
    // conditional subtraction of p (result < 2p): borrow-lookahead
    const uint32_t bg = r < c.pk ? 1u : 0u, bp = r == c.pk ? 1u : 0u;
    uint32_t G = bg, Pp = bp;
    uint32_t gs = oct_up<1>(G, c.k), ps = oct_up<1>(Pp, c.k);
    G |= Pp & gs;
    Pp &= ps;
    gs = oct_up<2>(G, c.k);
    ps = oct_up<2>(Pp, c.k);
    G |= Pp & gs;
    Pp &= ps;
    gs = oct_up<4>(G, c.k);
    G |= Pp & gs;
    const uint32_t bin = oct_up<1>(G, c.k);
    const uint32_t d = r - c.pk - bin;
    const uint32_t below = oct_bcast<7>(G);  // borrow out of the top limb: r < p
    return below ? r : d;
}

}  // namespace h2
