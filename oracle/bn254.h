/*
 * oracle/bn254.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * BN254 ("bn256") scalar field Fr, base field Fq and the G1 group
 * y^2 = x^3 + 3, restating the arithmetic the reference obtains from the
 * un-vendored crate `pairing_bn256` (github.com/lanbones/pairing @30b052f,
 * /root/reference/Cargo.lock:1284-1286; re-exported at
 * /root/reference/halo2_proofs/src/arithmetic.rs:16-17).
 *
 * PARITY UNPINNED at the byte level: the reference tree holds no golden vectors
 * for this path and cannot be built here (no Rust toolchain).  The values are
 * pinned to mathematics instead: tests/golden/ holds vectors produced by an
 * independent Python big-integer computation (tests/golden/gen_golden.py).
 *
 * Layout assumptions:
 *   Fr / Fq     4 x u64 LE limbs, Montgomery form, R = 2^256
 *   G1Affine    {x: Fq, y: Fq} = 64 B, identity = (0, 0)
 *   G1          Jacobian {x, y, z: Fq} = 96 B, identity has z = 0
 */
#ifndef ORACLE_BN254_H
#define ORACLE_BN254_H

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint64_t l[4];
} u256;

/* ---- Fr: r = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001 ---- */
#define FP fr
#define FP_MOD0 0x43e1f593f0000001ULL
#define FP_MOD1 0x2833e84879b97091ULL
#define FP_MOD2 0xb85045b68181585dULL
#define FP_MOD3 0x30644e72e131a029ULL
#define FP_INV 0xc2e1f593efffffffULL
#define FP_R0 0xac96341c4ffffffbULL
#define FP_R1 0x36fc76959f60cd29ULL
#define FP_R2 0x666ea36f7879462eULL
#define FP_R3 0x0e0a77c19a07df2fULL
#define FP_R2_0 0x1bb8e645ae216da7ULL
#define FP_R2_1 0x53fe3ab1e35c59e3ULL
#define FP_R2_2 0x8c49833d53bb8085ULL
#define FP_R2_3 0x0216d0b17f4e44a5ULL
#include "fp_tmpl.h"

/* ---- Fq: q = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47 ---- */
#define FP fq
#define FP_MOD0 0x3c208c16d87cfd47ULL
#define FP_MOD1 0x97816a916871ca8dULL
#define FP_MOD2 0xb85045b68181585dULL
#define FP_MOD3 0x30644e72e131a029ULL
#define FP_INV 0x87d20782e4866389ULL
#define FP_R0 0xd35d438dc58f0d9dULL
#define FP_R1 0x0a78eb28f5c70b3dULL
#define FP_R2 0x666ea36f7879462cULL
#define FP_R3 0x0e0a77c19a07df2fULL
#define FP_R2_0 0xf32cfc5b538afa89ULL
#define FP_R2_1 0xb5e71911d44501fbULL
#define FP_R2_2 0x47ab1eff0a417ff6ULL
#define FP_R2_3 0x06d89f71cab8351fULL
#include "fp_tmpl.h"

/* Fr constants (canonical integers; `PrimeField::S`, `root_of_unity`, `FieldExt::ZETA`).
 * ROOT_OF_UNITY = 7^((r-1)/2^28); ZETA is the halo2curves value (a primitive cube
 * root of unity) -- which of the two cube roots pairing_bn256@30b052f uses is
 * "parity unpinned" (SURVEY.md 8(c) item 1); the product takes zeta as an argument. */
#define FR_S 28
static const u256 FR_ROOT_OF_UNITY_CANON = {{0xd34f1ed960c37c9cULL, 0x3215cf6dd39329c8ULL,
                                             0x98865ea93dd31f74ULL, 0x03ddb9f5166d18b7ULL}};
static const u256 FR_ZETA_CANON = {{0xb8ca0b2d36636f23ULL, 0xcc37a73fec2bc5e9ULL,
                                    0x048b6e193fd84104ULL, 0x30644e72e131a029ULL}};

/* ------------------------------- G1 ------------------------------------- */
typedef struct {
    u256 x, y;
} g1_affine;
typedef struct {
    u256 x, y, z;
} g1_jac;

static inline int g1a_is_identity(const g1_affine *p) { return fq_is_zero(&p->x) && fq_is_zero(&p->y); }
static inline int g1j_is_identity(const g1_jac *p) { return fq_is_zero(&p->z); }
static inline void g1j_set_identity(g1_jac *p) {
    p->x = fq_ZERO;
    p->y = fq_ONE;
    p->z = fq_ZERO;
}
static inline void g1j_from_affine(g1_jac *r, const g1_affine *p) {
    if (g1a_is_identity(p)) {
        g1j_set_identity(r);
        return;
    }
    r->x = p->x;
    r->y = p->y;
    r->z = fq_ONE;
}

/* dbl-2009-l (a = 0) */
static inline void g1j_double(g1_jac *r, const g1_jac *p) {
    if (g1j_is_identity(p)) {
        g1j_set_identity(r);
        return;
    }
    u256 a, b, c, d, e, f, t, x3, y3, z3;
    fq_sqr(&a, &p->x);
    fq_sqr(&b, &p->y);
    fq_sqr(&c, &b);
    fq_add(&d, &p->x, &b);
    fq_sqr(&d, &d);
    fq_sub(&d, &d, &a);
    fq_sub(&d, &d, &c);
    fq_dbl(&d, &d);
    fq_dbl(&e, &a);
    fq_add(&e, &e, &a);
    fq_sqr(&f, &e);
    fq_dbl(&t, &d);
    fq_sub(&x3, &f, &t);
    fq_sub(&t, &d, &x3);
    fq_mul(&y3, &e, &t);
    fq_dbl(&c, &c);
    fq_dbl(&c, &c);
    fq_dbl(&c, &c);
    fq_sub(&y3, &y3, &c);
    fq_mul(&z3, &p->y, &p->z);
    fq_dbl(&z3, &z3);
    r->x = x3;
    r->y = y3;
    r->z = z3;
}

/* add-2007-bl with the exceptional cases handled explicitly */
static inline void g1j_add(g1_jac *r, const g1_jac *p, const g1_jac *q) {
    if (g1j_is_identity(p)) {
        *r = *q;
        return;
    }
    if (g1j_is_identity(q)) {
        *r = *p;
        return;
    }
    u256 z1z1, z2z2, u1, u2, s1, s2, h, i, j, rr, v, t, x3, y3, z3;
    fq_sqr(&z1z1, &p->z);
    fq_sqr(&z2z2, &q->z);
    fq_mul(&u1, &p->x, &z2z2);
    fq_mul(&u2, &q->x, &z1z1);
    fq_mul(&s1, &p->y, &q->z);
    fq_mul(&s1, &s1, &z2z2);
    fq_mul(&s2, &q->y, &p->z);
    fq_mul(&s2, &s2, &z1z1);
    if (fq_eq(&u1, &u2)) {
        if (fq_eq(&s1, &s2))
            g1j_double(r, p);
        else
            g1j_set_identity(r);
        return;
    }
    fq_sub(&h, &u2, &u1);
    fq_dbl(&i, &h);
    fq_sqr(&i, &i);
    fq_mul(&j, &h, &i);
    fq_sub(&rr, &s2, &s1);
    fq_dbl(&rr, &rr);
    fq_mul(&v, &u1, &i);
    fq_sqr(&x3, &rr);
    fq_sub(&x3, &x3, &j);
    fq_sub(&x3, &x3, &v);
    fq_sub(&x3, &x3, &v);
    fq_sub(&t, &v, &x3);
    fq_mul(&y3, &rr, &t);
    fq_mul(&t, &s1, &j);
    fq_dbl(&t, &t);
    fq_sub(&y3, &y3, &t);
    fq_add(&z3, &p->z, &q->z);
    fq_sqr(&z3, &z3);
    fq_sub(&z3, &z3, &z1z1);
    fq_sub(&z3, &z3, &z2z2);
    fq_mul(&z3, &z3, &h);
    r->x = x3;
    r->y = y3;
    r->z = z3;
}

/* madd-2007-bl (q affine) with the exceptional cases handled explicitly */
static inline void g1j_add_affine(g1_jac *r, const g1_jac *p, const g1_affine *q) {
    if (g1a_is_identity(q)) {
        *r = *p;
        return;
    }
    if (g1j_is_identity(p)) {
        g1j_from_affine(r, q);
        return;
    }
    u256 z1z1, u2, s2, h, hh, i, j, rr, v, t, x3, y3, z3;
    fq_sqr(&z1z1, &p->z);
    fq_mul(&u2, &q->x, &z1z1);
    fq_mul(&s2, &q->y, &p->z);
    fq_mul(&s2, &s2, &z1z1);
    if (fq_eq(&p->x, &u2)) {
        if (fq_eq(&p->y, &s2))
            g1j_double(r, p);
        else
            g1j_set_identity(r);
        return;
    }
    fq_sub(&h, &u2, &p->x);
    fq_sqr(&hh, &h);
    fq_dbl(&i, &hh);
    fq_dbl(&i, &i);
    fq_mul(&j, &h, &i);
    fq_sub(&rr, &s2, &p->y);
    fq_dbl(&rr, &rr);
    fq_mul(&v, &p->x, &i);
    fq_sqr(&x3, &rr);
    fq_sub(&x3, &x3, &j);
    fq_sub(&x3, &x3, &v);
    fq_sub(&x3, &x3, &v);
    fq_sub(&t, &v, &x3);
    fq_mul(&y3, &rr, &t);
    fq_mul(&t, &p->y, &j);
    fq_dbl(&t, &t);
    fq_sub(&y3, &y3, &t);
    fq_add(&z3, &p->z, &h);
    fq_sqr(&z3, &z3);
    fq_sub(&z3, &z3, &z1z1);
    fq_sub(&z3, &z3, &hh);
    r->x = x3;
    r->y = y3;
    r->z = z3;
}

static inline void g1j_to_affine(g1_affine *r, const g1_jac *p) {
    if (g1j_is_identity(p)) {
        r->x = fq_ZERO;
        r->y = fq_ZERO;
        return;
    }
    u256 zi, zi2, zi3;
    fq_inv(&zi, &p->z);
    fq_sqr(&zi2, &zi);
    fq_mul(&zi3, &zi2, &zi);
    fq_mul(&r->x, &p->x, &zi2);
    fq_mul(&r->y, &p->y, &zi3);
}

static inline void g1a_neg(g1_affine *r, const g1_affine *p) {
    r->x = p->x;
    fq_neg(&r->y, &p->y);
}

/* projective equality (the reference's `==` on C::Curve, e.g. poly/commitment.rs:494) */
static inline int g1j_eq(const g1_jac *p, const g1_jac *q) {
    int pi = g1j_is_identity(p), qi = g1j_is_identity(q);
    if (pi || qi) return pi && qi;
    u256 z1z1, z2z2, a, b;
    fq_sqr(&z1z1, &p->z);
    fq_sqr(&z2z2, &q->z);
    fq_mul(&a, &p->x, &z2z2);
    fq_mul(&b, &q->x, &z1z1);
    if (!fq_eq(&a, &b)) return 0;
    fq_mul(&a, &p->y, &q->z);
    fq_mul(&a, &a, &z2z2);
    fq_mul(&b, &q->y, &p->z);
    fq_mul(&b, &b, &z1z1);
    return fq_eq(&a, &b);
}

/* [k]P, k = canonical 4 x u64 integer (double-and-add, MSB first) */
static inline void g1j_mul_canon(g1_jac *r, const g1_jac *p, const u256 *k) {
    g1_jac acc;
    g1j_set_identity(&acc);
    for (int i = 255; i >= 0; i--) {
        g1j_double(&acc, &acc);
        if ((k->l[i / 64] >> (i % 64)) & 1) g1j_add(&acc, &acc, p);
    }
    *r = acc;
}

/* on-curve check y^2 == x^3 + 3 (identity accepted) */
static inline int g1a_on_curve(const g1_affine *p) {
    if (g1a_is_identity(p)) return 1;
    u256 y2, x3, b;
    fq_sqr(&y2, &p->y);
    fq_sqr(&x3, &p->x);
    fq_mul(&x3, &x3, &p->x);
    fq_from_u64(&b, 3);
    fq_add(&x3, &x3, &b);
    return fq_eq(&y2, &x3);
}

#endif /* ORACLE_BN254_H */
