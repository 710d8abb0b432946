// ec.hpp -- BN254 G1 (y^2 = x^3 + 3) group law for the MSM, host + device.
//
// Replaces the `pairing_bn256::bn256::{G1Affine, G1}` operations the reference uses in
// multiexp_serial (/root/reference/halo2_proofs/src/arithmetic.rs:55-106: `a + *other`,
// `+=`, `double`) and ec-gpu-gen's generated POINT_* CUDA source.
//
// Bucket accumulators use extended Jacobian ("XYZZ": x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2)
// coordinates: mixed addition is 8M + 2S with no inversion.  Identity <=> ZZ == 0.
// Every exceptional case (identity operands, P + P, P + (-P)) is handled explicitly: SRS
// bases and sparse witness columns make them common, not adversarial.
#pragma once
#include "field.hpp"

namespace h2 {

struct alignas(16) Affine {  // G1Affine: 64 B, identity = (0, 0)
    Fq x, y;
};
struct alignas(16) XYZZ {  // 128 B
    Fq x, y, zz, zzz;
};
struct alignas(16) Jacobian {  // G1: 96 B, identity has z = 0
    Fq x, y, z;
};

H2_DEV bool affine_is_identity(const Affine& p) { return fp_is_zero(p.x) && fp_is_zero(p.y); }
H2_DEV bool xyzz_is_identity(const XYZZ& p) { return fp_is_zero(p.zz); }

H2_DEV XYZZ xyzz_identity() {
    XYZZ r;
    r.x = fp_zero<FqParams>();
    r.y = fp_one<FqParams>();
    r.zz = fp_zero<FqParams>();
    r.zzz = fp_zero<FqParams>();
    return r;
}

H2_DEV XYZZ xyzz_from_affine(const Affine& p, bool negate) {
    if (affine_is_identity(p)) return xyzz_identity();
    XYZZ r;
    r.x = p.x;
    r.y = negate ? fp_neg(p.y) : p.y;
    r.zz = fp_one<FqParams>();
    r.zzz = fp_one<FqParams>();
    return r;
}

// 2 * (affine point)   [mdbl-2008-s-1]
H2_DEV XYZZ xyzz_double_affine(const Fq& x1, const Fq& y1) {
    Fq u = fp_dbl(y1);
    Fq v = fp_sqr(u);
    Fq w = fp_mul(u, v);
    Fq s = fp_mul(x1, v);
    Fq xx = fp_sqr(x1);
    Fq m = fp_add(fp_dbl(xx), xx);
    XYZZ r;
    r.x = fp_sub(fp_sqr(m), fp_dbl(s));
    r.y = fp_sub(fp_mul(m, fp_sub(s, r.x)), fp_mul(w, y1));
    r.zz = v;
    r.zzz = w;
    return r;
}

// 2 * P   [dbl-2008-s-1, a = 0]
H2_DEV XYZZ xyzz_double(const XYZZ& p) {
    if (xyzz_is_identity(p)) return p;
    Fq u = fp_dbl(p.y);
    Fq v = fp_sqr(u);
    Fq w = fp_mul(u, v);
    Fq s = fp_mul(p.x, v);
    Fq xx = fp_sqr(p.x);
    Fq m = fp_add(fp_dbl(xx), xx);
    XYZZ r;
    r.x = fp_sub(fp_sqr(m), fp_dbl(s));
    r.y = fp_sub(fp_mul(m, fp_sub(s, r.x)), fp_mul(w, p.y));
    r.zz = fp_mul(v, p.zz);
    r.zzz = fp_mul(w, p.zzz);
    return r;
}

// acc + (+-)q, q affine   [madd-2008-s]
H2_DEV XYZZ xyzz_madd(const XYZZ& acc, const Affine& q, bool negate) {
    if (affine_is_identity(q)) return acc;
    Fq qy = negate ? fp_neg(q.y) : q.y;
    if (xyzz_is_identity(acc)) {
        XYZZ r;
        r.x = q.x;
        r.y = qy;
        r.zz = fp_one<FqParams>();
        r.zzz = fp_one<FqParams>();
        return r;
    }
    Fq u2 = fp_mul(q.x, acc.zz);
    Fq s2 = fp_mul(qy, acc.zzz);
    Fq p = fp_sub(u2, acc.x);
    Fq r_ = fp_sub(s2, acc.y);
    if (fp_is_zero(p)) {
        if (fp_is_zero(r_)) return xyzz_double_affine(q.x, qy);  // acc == q
        return xyzz_identity();                                  // acc == -q
    }
    Fq pp = fp_sqr(p);
    Fq ppp = fp_mul(p, pp);
    Fq qq = fp_mul(acc.x, pp);
    XYZZ r;
    r.x = fp_sub(fp_sub(fp_sqr(r_), ppp), fp_dbl(qq));
    r.y = fp_mul2(r_, fp_sub(qq, r.x), fp_neg(acc.y), ppp);  // R (Q - X3) - Y1 PPP under one reduction
    r.zz = fp_mul(acc.zz, pp);
    r.zzz = fp_mul(acc.zzz, ppp);
    return r;
}

// a + b   [add-2008-s]
H2_DEV XYZZ xyzz_add(const XYZZ& a, const XYZZ& b) {
    if (xyzz_is_identity(a)) return b;
    if (xyzz_is_identity(b)) return a;
    Fq u1 = fp_mul(a.x, b.zz);
    Fq u2 = fp_mul(b.x, a.zz);
    Fq s1 = fp_mul(a.y, b.zzz);
    Fq s2 = fp_mul(b.y, a.zzz);
    Fq p = fp_sub(u2, u1);
    Fq r_ = fp_sub(s2, s1);
    if (fp_is_zero(p)) {
        if (fp_is_zero(r_)) return xyzz_double(a);
        return xyzz_identity();
    }
    Fq pp = fp_sqr(p);
    Fq ppp = fp_mul(p, pp);
    Fq qq = fp_mul(u1, pp);
    XYZZ r;
    r.x = fp_sub(fp_sub(fp_sqr(r_), ppp), fp_dbl(qq));
    r.y = fp_sub(fp_mul(r_, fp_sub(qq, r.x)), fp_mul(s1, ppp));
    r.zz = fp_mul(fp_mul(a.zz, b.zz), pp);
    r.zzz = fp_mul(fp_mul(a.zzz, b.zzz), ppp);
    return r;
}

// [k] P for a small unsigned k (double-and-add, MSB first)
H2_DEV XYZZ xyzz_mul_u32(const XYZZ& p, uint32_t k) {
    XYZZ acc = xyzz_identity();
    for (int i = 31; i >= 0; i--) {
        acc = xyzz_double(acc);
        if ((k >> i) & 1) acc = xyzz_add(acc, p);
    }
    return acc;
}

// XYZZ -> Jacobian (X*ZZ, Y*ZZZ, ZZ): x = X*ZZ/ZZ^2, y = Y*ZZZ/ZZ^3 since ZZ^3 = ZZZ^2
H2_DEV Jacobian xyzz_to_jacobian(const XYZZ& p) {
    Jacobian r;
    if (xyzz_is_identity(p)) {
        r.x = fp_zero<FqParams>();
        r.y = fp_one<FqParams>();
        r.z = fp_zero<FqParams>();
        return r;
    }
    r.x = fp_mul(p.x, p.zz);
    r.y = fp_mul(p.y, p.zzz);
    r.z = p.zz;
    return r;
}

H2_DEV XYZZ xyzz_load(const XYZZ* p) {
    XYZZ r;
    r.x = fp_load(&p->x);
    r.y = fp_load(&p->y);
    r.zz = fp_load(&p->zz);
    r.zzz = fp_load(&p->zzz);
    return r;
}
H2_DEV void xyzz_store(XYZZ* p, const XYZZ& v) {
    fp_store(&p->x, v.x);
    fp_store(&p->y, v.y);
    fp_store(&p->zz, v.zz);
    fp_store(&p->zzz, v.zzz);
}
H2_DEV Affine affine_load(const Affine* p) {
    Affine r;
    r.x = fp_load(&p->x);
    r.y = fp_load(&p->y);
    return r;
}

}  // namespace h2
