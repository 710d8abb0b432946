/*
 * oracle/fp_tmpl.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * 254-bit prime-field arithmetic, 4 x u64 little-endian limbs, Montgomery form
 * (R = 2^256).  Included twice by bn254.h: once for Fr (BN254 scalar field) and
 * once for Fq (BN254 base field).
 *
 * What it restates: the arithmetic of `pairing_bn256::bn256::{Fr,Fq}` -- an
 * UN-VENDORED dependency of the reference (github.com/lanbones/pairing, rev
 * 30b052f29d7ec3e68e20528584d9495de678ea05; /root/reference/Cargo.lock:1284-1286).
 * Its source is not in /root/reference, so this is a restatement of the PUBLISHED
 * algorithm (Montgomery CIOS over the public BN254 moduli) and is pinned against
 * Python big-integer arithmetic (tests/golden/, tests/test_oracle_golden.py).
 * What the reference tree itself establishes: Fr is 32 B = 4 x u64 in Montgomery
 * form in memory (halo2_proofs/src/plonk/prover.rs:176,183; helpers.rs:185-194).
 *
 * Before including define:
 *   FP            name prefix (fr / fq)
 *   FP_MOD0..3    modulus limbs
 *   FP_INV        -p^{-1} mod 2^64
 *   FP_R0..3      R   mod p   (Montgomery one)
 *   FP_R2_0..3    R^2 mod p
 */
#define FP_CAT_(a, b) a##_##b
#define FP_CAT(a, b) FP_CAT_(a, b)
#define FN(name) FP_CAT(FP, name)

static const uint64_t FN(MOD)[4] = {FP_MOD0, FP_MOD1, FP_MOD2, FP_MOD3};
static const u256 FN(ONE) = {{FP_R0, FP_R1, FP_R2, FP_R3}};
static const u256 FN(RR) = {{FP_R2_0, FP_R2_1, FP_R2_2, FP_R2_3}};
static const u256 FN(ZERO) = {{0, 0, 0, 0}};

static inline int FN(is_zero)(const u256 *a) {
    return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0;
}

static inline int FN(eq)(const u256 *a, const u256 *b) {
    return a->l[0] == b->l[0] && a->l[1] == b->l[1] && a->l[2] == b->l[2] && a->l[3] == b->l[3];
}

/* a >= p ? */
static inline int FN(geq_mod)(const uint64_t *a) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] > FN(MOD)[i]) return 1;
        if (a[i] < FN(MOD)[i]) return 0;
    }
    return 1;
}

static inline void FN(sub_mod_raw)(uint64_t *a) {
    unsigned __int128 borrow = 0;
    for (int i = 0; i < 4; i++) {
        unsigned __int128 d = (unsigned __int128)a[i] - FN(MOD)[i] - (uint64_t)borrow;
        a[i] = (uint64_t)d;
        borrow = (d >> 64) & 1;
    }
}

static inline void FN(add)(u256 *r, const u256 *a, const u256 *b) {
    unsigned __int128 c = 0;
    uint64_t t[4];
    for (int i = 0; i < 4; i++) {
        c += (unsigned __int128)a->l[i] + b->l[i];
        t[i] = (uint64_t)c;
        c >>= 64;
    }
    /* p < 2^254 so a + b < 2^255: no carry out of limb 3 */
    if (FN(geq_mod)(t)) FN(sub_mod_raw)(t);
    memcpy(r->l, t, 32);
}

static inline void FN(sub)(u256 *r, const u256 *a, const u256 *b) {
    uint64_t t[4];
    unsigned __int128 borrow = 0;
    for (int i = 0; i < 4; i++) {
        unsigned __int128 d = (unsigned __int128)a->l[i] - b->l[i] - (uint64_t)borrow;
        t[i] = (uint64_t)d;
        borrow = (d >> 64) & 1;
    }
    if (borrow) {
        unsigned __int128 c = 0;
        for (int i = 0; i < 4; i++) {
            c += (unsigned __int128)t[i] + FN(MOD)[i];
            t[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    memcpy(r->l, t, 32);
}

static inline void FN(neg)(u256 *r, const u256 *a) {
    if (FN(is_zero)(a)) {
        *r = *a;
        return;
    }
    u256 z = FN(ZERO);
    FN(sub)(r, &z, a);
}

static inline void FN(dbl)(u256 *r, const u256 *a) { FN(add)(r, a, a); }

/* Montgomery product a*b*R^-1 mod p  (CIOS, 4 limbs) */
static inline void FN(mul)(u256 *r, const u256 *a, const u256 *b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        unsigned __int128 acc;
        uint64_t carry = 0;
        for (int j = 0; j < 4; j++) {
            acc = (unsigned __int128)a->l[j] * b->l[i] + t[j] + carry;
            t[j] = (uint64_t)acc;
            carry = (uint64_t)(acc >> 64);
        }
        acc = (unsigned __int128)t[4] + carry;
        t[4] = (uint64_t)acc;
        t[5] = (uint64_t)(acc >> 64);

        uint64_t m = t[0] * FP_INV;
        acc = (unsigned __int128)m * FN(MOD)[0] + t[0];
        carry = (uint64_t)(acc >> 64);
        for (int j = 1; j < 4; j++) {
            acc = (unsigned __int128)m * FN(MOD)[j] + t[j] + carry;
            t[j - 1] = (uint64_t)acc;
            carry = (uint64_t)(acc >> 64);
        }
        acc = (unsigned __int128)t[4] + carry;
        t[3] = (uint64_t)acc;
        t[4] = t[5] + (uint64_t)(acc >> 64);
    }
    if (t[4] || FN(geq_mod)(t)) FN(sub_mod_raw)(t);
    memcpy(r->l, t, 32);
}

static inline void FN(sqr)(u256 *r, const u256 *a) { FN(mul)(r, a, a); }

/* canonical integer (4 limbs, < p) -> Montgomery */
static inline void FN(from_repr)(u256 *r, const u256 *canon) { FN(mul)(r, canon, &FN(RR)); }

/* Montgomery -> canonical little-endian integer (`PrimeField::to_repr`,
 * used at /root/reference/halo2_proofs/src/arithmetic.rs:21) */
static inline void FN(to_repr)(u256 *r, const u256 *a) {
    u256 one = {{1, 0, 0, 0}};
    FN(mul)(r, a, &one);
}

static inline void FN(from_u64)(u256 *r, uint64_t v) {
    u256 c = {{v, 0, 0, 0}};
    FN(from_repr)(r, &c);
}

/* a^e, e = 4 x u64 little-endian plain integer (`pow_vartime`) */
static inline void FN(pow)(u256 *r, const u256 *a, const uint64_t e[4]) {
    u256 acc = FN(ONE);
    int started = 0;
    for (int i = 255; i >= 0; i--) {
        if (started) FN(sqr)(&acc, &acc);
        if ((e[i / 64] >> (i % 64)) & 1) {
            FN(mul)(&acc, &acc, a);
            started = 1;
        }
    }
    *r = acc;
}

static inline void FN(pow_u64)(u256 *r, const u256 *a, uint64_t e) {
    uint64_t ee[4] = {e, 0, 0, 0};
    FN(pow)(r, a, ee);
}

/* a^-1 = a^(p-2)  (0 -> 0) */
static inline void FN(inv)(u256 *r, const u256 *a) {
    uint64_t e[4] = {FP_MOD0 - 2, FP_MOD1, FP_MOD2, FP_MOD3}; /* low limb of both moduli is >= 2 */
    FN(pow)(r, a, e);
}

/* canonical-integer comparison of two Montgomery values (`Ord for Fr`, used by
 * find_max_scalar_bits at /root/reference/halo2_proofs/src/plonk/prover.rs:252-254;
 * assumed canonical order -- SURVEY.md section 8(c) "parity unpinned" item 3) */
static inline int FN(cmp)(const u256 *a, const u256 *b) {
    u256 ca, cb;
    FN(to_repr)(&ca, a);
    FN(to_repr)(&cb, b);
    for (int i = 3; i >= 0; i--) {
        if (ca.l[i] > cb.l[i]) return 1;
        if (ca.l[i] < cb.l[i]) return -1;
    }
    return 0;
}

/* Montgomery's trick, in place; zeros are left as zero (`ff::BatchInvert`) */
static inline void FN(batch_invert)(u256 *v, size_t n) {
    if (n == 0) return;
    u256 *prefix = (u256 *)malloc(n * sizeof(u256));
    u256 acc = FN(ONE);
    for (size_t i = 0; i < n; i++) {
        prefix[i] = acc;
        if (!FN(is_zero)(&v[i])) FN(mul)(&acc, &acc, &v[i]);
    }
    u256 inv;
    FN(inv)(&inv, &acc);
    for (size_t i = n; i-- > 0;) {
        if (FN(is_zero)(&v[i])) continue;
        u256 t;
        FN(mul)(&t, &inv, &prefix[i]);
        FN(mul)(&inv, &inv, &v[i]);
        v[i] = t;
    }
    free(prefix);
}

#undef FN
#undef FP
#undef FP_MOD0
#undef FP_MOD1
#undef FP_MOD2
#undef FP_MOD3
#undef FP_INV
#undef FP_R0
#undef FP_R1
#undef FP_R2
#undef FP_R3
#undef FP_R2_0
#undef FP_R2_1
#undef FP_R2_2
#undef FP_R2_3
