// ec_quad.hpp -- the XYZZ group law spread over the four lanes of a quad (device only).
//
// The tail of an MSM (folding slice partials into buckets, the per-window running sums, their trees) is bound by the
// DEPTH of dependent point additions, not by their count: one 254-bit Montgomery product is ~0.5 us on a wave however
// few lanes are active, an addition is 14 of them back to back.  The 14 products of add-2008-s form only four dependent
// levels (four products each), those of dbl-2008-s-1 three: lanes 4k .. 4k+3 hold the SAME operands, each computes one
// product of the level, and the quad exchanges them with `v_mov_b32 quad_perm` (8 moves per field element).  A chain of
// additions then advances in 4 product times instead of 14.
//
// Every function expects its point arguments replicated over the quad and returns a replicated result, so every
// exceptional-case branch (identity operands, P + P, P - P) is taken by whole quads and the cross-lane moves always
// see four active lanes.  The formulas are those of ec.hpp: results are limb-for-limb what xyzz_add / xyzz_double give.
#pragma once
#include "ec.hpp"

namespace h2 {

template <int K>
__device__ __forceinline__ Fq quad_bcast(const Fq& v) {  // lane K of the quad -> all four
    Fq r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)v.l[i], K * 0x55, 0xF, 0xF, true);
    return r;
}

__device__ __forceinline__ Fq quad_pick(uint32_t q, const Fq& a0, const Fq& a1, const Fq& a2, const Fq& a3) {
    Fq r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t lo = (q & 1) ? a1.l[i] : a0.l[i];
        uint32_t hi = (q & 1) ? a3.l[i] : a2.l[i];
        r.l[i] = (q & 2) ? hi : lo;
    }
    return r;
}

// 2 * P, three product levels
__device__ __forceinline__ XYZZ xyzz_double_q(const XYZZ& p, uint32_t q) {
    if (xyzz_is_identity(p)) return p;
    const Fq u = fp_dbl(p.y);
    Fq t = fp_mul(quad_pick(q, u, p.x, u, p.x), quad_pick(q, u, p.x, u, p.x));
    const Fq v = quad_bcast<0>(t), xx = quad_bcast<1>(t);
    const Fq m = fp_add(fp_dbl(xx), xx);
    t = fp_mul(quad_pick(q, u, p.x, m, v), quad_pick(q, v, v, m, p.zz));
    const Fq w = quad_bcast<0>(t), s = quad_bcast<1>(t), mm = quad_bcast<2>(t);
    XYZZ r;
    r.zz = quad_bcast<3>(t);
    r.x = fp_sub(mm, fp_dbl(s));
    t = fp_mul(quad_pick(q, m, w, w, w), quad_pick(q, fp_sub(s, r.x), p.y, p.zzz, p.zzz));
    r.y = fp_sub(quad_bcast<0>(t), quad_bcast<1>(t));
    r.zzz = quad_bcast<2>(t);
    return r;
}

// a + b, four product levels
__device__ __forceinline__ XYZZ xyzz_add_q(const XYZZ& a, const XYZZ& b, uint32_t q) {
    if (xyzz_is_identity(a)) return b;
    if (xyzz_is_identity(b)) return a;
    Fq t = fp_mul(quad_pick(q, a.x, b.x, a.y, b.y), quad_pick(q, b.zz, a.zz, b.zzz, a.zzz));
    const Fq u1 = quad_bcast<0>(t), u2 = quad_bcast<1>(t), s1 = quad_bcast<2>(t), s2 = quad_bcast<3>(t);
    const Fq p = fp_sub(u2, u1);
    const Fq r_ = fp_sub(s2, s1);
    if (fp_is_zero(p)) {
        if (fp_is_zero(r_)) return xyzz_double_q(a, q);
        return xyzz_identity();
    }
    t = fp_mul(quad_pick(q, p, r_, a.zz, a.zzz), quad_pick(q, p, r_, b.zz, b.zzz));
    const Fq pp = quad_bcast<0>(t), rr = quad_bcast<1>(t), zz12 = quad_bcast<2>(t), zzz12 = quad_bcast<3>(t);
    t = fp_mul(quad_pick(q, p, u1, zz12, zz12), pp);
    const Fq ppp = quad_bcast<0>(t), qq = quad_bcast<1>(t);
    XYZZ r;
    r.zz = quad_bcast<2>(t);
    r.x = fp_sub(fp_sub(rr, ppp), fp_dbl(qq));
    t = fp_mul(quad_pick(q, r_, s1, zzz12, zzz12), quad_pick(q, fp_sub(qq, r.x), ppp, ppp, ppp));
    r.y = fp_sub(quad_bcast<0>(t), quad_bcast<1>(t));
    r.zzz = quad_bcast<2>(t);
    return r;
}

// [k] P, double-and-add from the top set bit
__device__ __forceinline__ XYZZ xyzz_mul_u32_q(const XYZZ& p, uint32_t k, uint32_t q) {
    XYZZ acc = xyzz_identity();
    if (k == 0 || xyzz_is_identity(p)) return acc;
    for (int i = 31 - __clz(k); i >= 0; i--) {
        acc = xyzz_double_q(acc, q);
        if ((k >> i) & 1) acc = xyzz_add_q(acc, p, q);
    }
    return acc;
}

}  // namespace h2
