// evalh_gen.cpp -- the evaluate_h code generator and its hipRTC / cache plumbing (host C++ only; see evalh_gen.hpp).
//
// Reference: Evaluator::evaluate_h, plonk/evaluation.rs:778-1226 (CPU twin) / :1229-1985 (cuda), whose value is
//     h = sum_j T_j y^(N-1-j)
// over the N terms T_j in the reference's order: the gate value parts (:891-901), the permutation argument's terms
// (:1004-1085), every lookup's (:1138-1182) and every shuffle's (:1197-1219).  All of it is exact arithmetic in F_r on
// canonical residues, so ANY algebraically equal evaluation order produces the same bits.  The generator uses that:
//   * the program is rebuilt as a hash-consed DAG (products by 0 / 1 / -1 / 2 disappear into nothing / a negation / a
//     doubling, a x a is a squaring) and emitted on demand, each intermediate right before its first use -- a lookup's
//     compressed product is computed when the lookup's term needs it, not 100 statements earlier;
//   * terms that share a factor F (l_0, l_last, l_active_row, or a gate selector that is the top-level factor of several
//     value parts) are Horner-summed in powers of y FIRST and multiplied by F once:
//         sum_{j in G} (X_j F) y^e_j  =  F * y^e_min * Horner_{y^gaps}(X_j)
//     -- one product per term instead of two (the fold by y and the product by F);
//   * column pointers and every uniform scalar (challenges and their powers, y^e, constants) are kernel ARGUMENTS:
//     scalar loads, SGPR-based addressing, nothing staged through device memory per call;
//   * loads are value-numbered and issued one scheduling group ahead of their first use;
//   * a program too wide for one kernel (register budget, 4 KiB of arguments) is cut into stages that each ADD their
//     share of the sum into `values`.
#include "evalh_gen.hpp"

#include <dirent.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <atomic>
#include <mutex>
#include <stdexcept>

// The field layer the generated sources include, embedded byte for byte (hipRTC has no include path on a proving machine)
__asm__(
    ".section .rodata\n"
    ".global h2_embed_field_hpp\n"
    "h2_embed_field_hpp:\n"
    ".incbin \"field.hpp\"\n"
    ".byte 0\n"
    ".global h2_embed_fp_mul_gen_hpp\n"
    "h2_embed_fp_mul_gen_hpp:\n"
    ".incbin \"fp_mul_gen.hpp\"\n"
    ".byte 0\n"
    ".previous\n");
extern "C" const char h2_embed_field_hpp[];
extern "C" const char h2_embed_fp_mul_gen_hpp[];

namespace h2 {
namespace evgen {

namespace {

constexpr const char* GENERATOR_VERSION = "h2-evalh-gen 5.14";

[[noreturn]] void fail(const std::string& what) { throw std::runtime_error("evaluate_h generator: " + what); }

// ------------------------------------------------------------------------------------------------ SHA-256 (cache identity)
struct Sha256 {
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    uint8_t buf[64];
    size_t fill = 0;
    uint64_t total = 0;
    static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
    void block(const uint8_t* p) {
        static const uint32_t K[64] = {
            0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
            0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
            0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
            0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
            0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
            0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
            0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
        uint32_t w[64];
        for (int i = 0; i < 16; i++) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
        for (int i = 16; i < 64; i++) {
            uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
            uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; i++) {
            uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
            uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    void update(const void* data, size_t n) {
        const uint8_t* p = (const uint8_t*)data;
        total += n;
        while (n) {
            size_t take = std::min(n, 64 - fill);
            memcpy(buf + fill, p, take);
            fill += take; p += take; n -= take;
            if (fill == 64) { block(buf); fill = 0; }
        }
    }
    void u32(uint32_t v) { update(&v, 4); }
    void finish(uint8_t out[32]) {
        uint64_t bits = total * 8;
        uint8_t pad = 0x80;
        update(&pad, 1);
        pad = 0;
        while (fill != 56) update(&pad, 1);
        uint8_t len[8];
        for (int i = 0; i < 8; i++) len[i] = (uint8_t)(bits >> (56 - 8 * i));
        update(len, 8);
        for (int i = 0; i < 8; i++) { out[4 * i] = h[i] >> 24; out[4 * i + 1] = h[i] >> 16; out[4 * i + 2] = h[i] >> 8; out[4 * i + 3] = h[i]; }
    }
};

// ------------------------------------------------------------------------------------------------ the DAG
// BN254 F_r in Montgomery form (field.hpp FrParams): what a constant has to equal to be 1, -1 or 2
const uint64_t FR_MOD[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
const uint64_t FR_ONE[4] = {0xac96341c4ffffffbull, 0x36fc76959f60cd29ull, 0x666ea36f7879462eull, 0x0e0a77c19a07df2full};
enum ConstClass { CC_OTHER, CC_ZERO, CC_ONE, CC_MINUS_ONE, CC_TWO };

ConstClass classify(const uint64_t v[4]) {
    uint64_t m1[4], two[4];
    unsigned __int128 bw = 0, cy = 0;
    for (int i = 0; i < 4; i++) {
        unsigned __int128 dd = (unsigned __int128)FR_MOD[i] - FR_ONE[i] - (uint64_t)bw;
        m1[i] = (uint64_t)dd;
        bw = (dd >> 64) & 1;
        unsigned __int128 ss = (unsigned __int128)FR_ONE[i] + FR_ONE[i] + (uint64_t)cy;
        two[i] = (uint64_t)ss;
        cy = ss >> 64;
    }
    auto eq = [&](const uint64_t* w) { return v[0] == w[0] && v[1] == w[1] && v[2] == w[2] && v[3] == w[3]; };
    const uint64_t zero[4] = {0, 0, 0, 0};
    if (eq(zero)) return CC_ZERO;
    if (eq(FR_ONE)) return CC_ONE;
    if (eq(m1)) return CC_MINUS_ONE;
    if (eq(two)) return CC_TWO;   // 2 R < r: no reduction
    return CC_OTHER;
}

enum NodeOp : uint8_t { N_ZERO, N_LOAD, N_SCALAR, N_OMEGA, N_ADD, N_SUB, N_MUL, N_NEG, N_SQR, N_DBL };

struct Node {
    uint8_t op;
    int32_t a, b;   // operands (interior nodes)
    uint32_t x;     // LOAD: column id; SCALAR: scalar id
    int32_t y;      // LOAD: rotation (rows)
};

struct Term {
    int x;  // the term is X * F ...
    int f;  // ... or just X when f < 0
};

struct Builder {
    const h2_evalh_desc* d;
    const Options& opt;
    std::vector<Node> nodes;
    std::map<std::array<int64_t, 5>, int> memo;
    std::vector<ColRef> cols;
    std::map<std::pair<int, uint32_t>, int> col_ids;
    std::vector<ScalarRef> scalars;
    std::map<std::pair<int, uint32_t>, int> scalar_ids;
    std::vector<ConstClass> cclass;
    std::vector<int> inter;
    std::vector<Term> terms;
    std::vector<int> need_;  // Sethi-Ullman register need per node (lazy)
    uint32_t ref_products = 0;
    int zero_, one_, beta_, gamma_;

    Builder(const h2_evalh_desc* desc, const Options& o) : d(desc), opt(o) {}

    int mk(uint8_t op, int a, int b, uint32_t x = 0, int32_t y = 0) {
        int ka = a, kb = b;
        if ((op == N_ADD || op == N_MUL) && ka > kb) std::swap(ka, kb);
        std::array<int64_t, 5> key = {op, ka, kb, (int64_t)x, y};
        auto it = memo.find(key);
        if (it != memo.end()) return it->second;
        nodes.push_back(Node{op, a, b, x, y});
        memo[key] = (int)nodes.size() - 1;
        return (int)nodes.size() - 1;
    }
    int scalar(uint8_t kind, uint32_t arg = 0) {
        auto key = std::make_pair((int)kind, arg);
        auto it = scalar_ids.find(key);
        int id;
        if (it == scalar_ids.end()) {
            id = (int)scalars.size();
            scalars.push_back(ScalarRef{kind, arg});
            scalar_ids[key] = id;
        } else {
            id = it->second;
        }
        return mk(N_SCALAR, -1, -1, (uint32_t)id);
    }
    int load(uint8_t table, uint32_t index, int32_t rot) {
        auto key = std::make_pair((int)table, index);
        auto it = col_ids.find(key);
        int id;
        if (it == col_ids.end()) {
            id = (int)cols.size();
            cols.push_back(ColRef{table, index});
            col_ids[key] = id;
        } else {
            id = it->second;
        }
        return mk(N_LOAD, -1, -1, (uint32_t)id, rot);
    }
    bool is(int n, uint8_t op) const { return nodes[n].op == op; }
    bool is_const(int n, ConstClass c) const {
        if (c == CC_ZERO) return n == zero_;
        if (c == CC_ONE) return n == one_;
        if (!is(n, N_SCALAR)) return false;
        const ScalarRef& s = scalars[nodes[n].x];
        return s.kind == SC_CONST && cclass[s.arg] == c;
    }
    // ---- the field operations, simplified where the result is known to be the same field element
    int neg(int a) {
        if (a == zero_) return zero_;
        if (is(a, N_NEG)) return nodes[a].a;
        if (is(a, N_SUB)) return mk(N_SUB, nodes[a].b, nodes[a].a);
        return mk(N_NEG, a, -1);
    }
    int dbl(int a) { return a == zero_ ? zero_ : mk(N_DBL, a, -1); }
    int add(int a, int b) {
        if (a == zero_) return b;
        if (b == zero_) return a;
        if (a == b) return dbl(a);
        if (is(b, N_NEG)) return sub(a, nodes[b].a);
        if (is(a, N_NEG)) return sub(b, nodes[a].a);
        return mk(N_ADD, a, b);
    }
    int sub(int a, int b) {
        if (b == zero_) return a;
        if (a == zero_) return neg(b);
        if (a == b) return zero_;
        if (is(b, N_NEG)) return add(a, nodes[b].a);
        return mk(N_SUB, a, b);
    }
    int mul(int a, int b) {
        if (a == zero_ || b == zero_) return zero_;
        if (a == one_) return b;
        if (b == one_) return a;
        if (is_const(a, CC_MINUS_ONE)) return neg(b);
        if (is_const(b, CC_MINUS_ONE)) return neg(a);
        if (is_const(a, CC_TWO)) return dbl(b);
        if (is_const(b, CC_TWO)) return dbl(a);
        if (a == b) return mk(N_SQR, a, -1);
        return mk(N_MUL, a, b);
    }
    int constant(uint32_t index) {
        if (index >= d->n_constants) fail("constant index out of range");
        switch (cclass[index]) {
            case CC_ZERO: return zero_;
            case CC_ONE: return one_;
            default: return scalar(SC_CONST, index);
        }
    }
    int challenge(uint32_t which, uint32_t power) {
        if (which != H2_CHALLENGE_BETA && which != H2_CHALLENGE_GAMMA) fail("unknown challenge");
        return scalar(which == H2_CHALLENGE_BETA ? SC_BETA_POW : SC_GAMMA_POW, power > 1 ? power : 1);
    }
    int source(const h2_value_source& v, uint32_t inter_limit) {
        switch (v.kind) {
            case H2_VS_CONSTANT: return constant(v.index);
            case H2_VS_INTERMEDIATE:
                if (v.index >= inter_limit) fail("intermediate used before it is defined");
                return inter[v.index];
            case H2_VS_FIXED:
            case H2_VS_ADVICE:
            case H2_VS_INSTANCE: {
                const uint32_t lim = v.kind == H2_VS_FIXED ? d->n_fixed : (v.kind == H2_VS_ADVICE ? d->n_advice : d->n_instance);
                if (v.index >= lim) fail("column index out of range");
                if (v.rot >= d->n_rotations) fail("rotation index out of range");
                const uint8_t table = v.kind == H2_VS_FIXED ? T_FIXED : (v.kind == H2_VS_ADVICE ? T_ADVICE : T_INSTANCE);
                return load(table, v.index, d->rotations[v.rot]);
            }
            default: fail("unknown value source");
        }
    }
    // one `Calculation` (evaluation.rs:95-266; the interpreter's Interp::eval in evalh.hip)
    int calculation(const h2_calculation& c, uint32_t inter_limit) {
        const int a = source(c.a, inter_limit);
        switch (c.op) {
            case H2_CALC_ADD: return add(a, source(c.b, inter_limit));
            case H2_CALC_SUB: return sub(a, source(c.b, inter_limit));
            case H2_CALC_MUL: ref_products++; return mul(a, source(c.b, inter_limit));
            case H2_CALC_NEGATE: return neg(a);
            case H2_CALC_LC_CHALLENGE: ref_products++; return mul(add(a, challenge(c.challenge, c.power)), source(c.b, inter_limit));
            case H2_CALC_LC_THETA: ref_products++; return add(mul(a, scalar(SC_THETA)), source(c.b, inter_limit));
            case H2_CALC_ADD_CHALLENGE: return add(a, challenge(c.challenge, 1));
            case H2_CALC_STORE: return a;
            default: fail("unknown calculation");
        }
    }
    void term(int x, int f) {
        if (opt.factor || f < 0) {
            terms.push_back(Term{x, f});
        } else {
            terms.push_back(Term{mul(x, f), -1});
        }
    }

    void build() {
        if (d->n_perm_sets && d->chunk_len == 0) fail("chunk_len must be non-zero when permutation sets are present");
        zero_ = mk(N_ZERO, -1, -1);
        one_ = scalar(SC_ONE);
        cclass.resize(d->n_constants);
        for (uint32_t i = 0; i < d->n_constants; i++) cclass[i] = classify(d->constants + 4 * (size_t)i);
        inter.assign(d->n_calculations, -1);
        for (uint32_t i = 0; i < d->n_calculations; i++) inter[i] = calculation(d->calculations[i], i);
        // ---- gate value parts (evaluation.rs:891-901)
        for (uint32_t i = 0; i < d->n_value_parts; i++) terms.push_back(Term{source(d->value_parts[i], d->n_calculations), -1});
        ref_products += d->n_value_parts;
        const int32_t last = -((int32_t)d->blinding_factors + 1);
        auto l0 = [&] { return load(T_L0, 0, 0); };
        auto l_last = [&] { return load(T_L_LAST, 0, 0); };
        auto l_active = [&] { return load(T_L_ACTIVE, 0, 0); };
        // ---- permutation argument (evaluation.rs:1004-1085; k_evalh_perm)
        if (d->n_perm_sets) {
            const uint32_t ns = d->n_perm_sets, nc = d->n_perm_columns, chunk = d->chunk_len;
            auto pz = [&](uint32_t s, int32_t rot) { return load(T_PERM_Z, s, rot); };
            const int beta = scalar(SC_BETA_POW, 1), gamma = scalar(SC_GAMMA_POW, 1);
            term(sub(one_, pz(0, 0)), l0());
            const int zl = pz(ns - 1, 0);
            term(sub(mul(zl, zl), zl), l_last());
            for (uint32_t s = 1; s < ns; s++) term(sub(pz(s, 0), pz(s - 1, last)), l0());
            const int omega = mk(N_OMEGA, -1, -1);
            ref_products += 2 + 3 + (ns - 1) * 2;
            for (uint32_t s = 0; s < ns; s++) {
                const uint32_t c0 = s * chunk, c1 = std::min(c0 + chunk, nc);
                int left = pz(s, 1), right = pz(s, 0);
                for (uint32_t j = c0; j < c1; j++) {
                    const uint32_t ty = d->perm_col_type[j], ix = d->perm_col_index[j];
                    const uint32_t lim = ty == H2_ANY_ADVICE ? d->n_advice : (ty == H2_ANY_FIXED ? d->n_fixed : (ty == H2_ANY_INSTANCE ? d->n_instance : 0));
                    if (ix >= lim) fail("permutation column index out of range");
                    const int v = load(ty == H2_ANY_ADVICE ? T_ADVICE : (ty == H2_ANY_FIXED ? T_FIXED : T_INSTANCE), ix, 0);
                    left = mul(left, add(add(v, mul(beta, load(T_PERM_SIGMA, j, 0))), gamma));
                    // current_delta of column j = beta ZETA DELTA^j extended_omega^idx (evaluation.rs:1012,1074): the uniform part
                    // is a kernel argument, one product per column as in the running form
                    right = mul(right, add(add(v, mul(omega, scalar(SC_DELTA_TERM, j))), gamma));
                    ref_products += 4;
                }
                term(sub(left, right), l_active());
                ref_products += 2;
            }
        }
        // ---- logup lookups (evaluation.rs:1138-1182; k_evalh_lookup)
        size_t zoff = 0, slot = 0;
        for (uint32_t t = 0; t < d->n_lookups; t++) {
            const uint32_t nset = d->lookup_sets[t];
            if (nset == 0) fail("a lookup without input sets");
            auto lz = [&](uint32_t i, int32_t rot) { return load(T_LOOKUP_Z, (uint32_t)(zoff + i), rot); };
            const int table = calculation(d->lookup_calcs[slot], d->n_calculations);
            std::vector<int> prod(nset), sum(nset);
            for (uint32_t i = 0; i < nset; i++) {
                prod[i] = calculation(d->lookup_calcs[slot + 1 + 2 * i], d->n_calculations);
                sum[i] = calculation(d->lookup_calcs[slot + 2 + 2 * i], d->n_calculations);
            }
            term(lz(0, 0), l0());
            term(lz(nset - 1, 0), l_last());
            term(sub(mul(add(mul(sub(lz(0, 1), lz(0, 0)), table), load(T_LOOKUP_M, t, 0)), prod[0]), mul(table, sum[0])), l_active());
            for (uint32_t i = 1; i < nset; i++) term(sub(lz(i, 0), lz(i - 1, last)), l0());
            for (uint32_t i = 1; i < nset; i++) term(sub(mul(sub(lz(i, 1), lz(i, 0)), prod[i]), sum[i]), l_active());
            ref_products += 2 + 2 + 5 + (nset - 1) * (2 + 3);
            zoff += nset;
            slot += 1 + 2 * (size_t)nset;
        }
        // ---- shuffles (evaluation.rs:1197-1219; k_evalh_shuffle)
        for (uint32_t s = 0; s < d->n_shuffles; s++) {
            const int input = calculation(d->shuffle_calcs[2 * s], d->n_calculations);
            const int shuffle = calculation(d->shuffle_calcs[2 * s + 1], d->n_calculations);
            const int z = load(T_SHUFFLE_Z, s, 0);
            term(sub(one_, z), l0());
            term(sub(mul(z, z), z), l_last());
            term(sub(mul(load(T_SHUFFLE_Z, s, 1), shuffle), mul(z, input)), l_active());
            ref_products += 2 + 3 + 4;
        }
        if (ref_products) ref_products--;  // the first term starts the fold
        if (opt.factor) {
            factor_gates();
            dissolve_small_groups();
        }
        count_uses();
    }

    // A group pays two closing products (y^e_min and F) for saving one per member: with fewer than three members (two when
    // products cannot be paired) the plain fold `S y + X F` -- one fp_mul2 -- is as cheap or cheaper: those terms go back to
    // the factor-less group (mini-PLONK: every argument group is a singleton; factored 7.33 ms, plain 7.15 ms at 2^25)
    void dissolve_small_groups() {
        std::map<int, int> size;
        for (const Term& t : terms)
            if (t.f >= 0) size[t.f]++;
        const int least = (int)(opt.mul2 ? opt.min_group : 2);
        for (Term& t : terms)
            if (t.f >= 0 && size[t.f] < least) t = Term{mul(t.x, t.f), -1};
    }

    // value parts that are products containing the same column value S (a selector: `q * (...)`, `q * (...) * (...)`) get S
    // as their group factor: the product tree is flattened (a b c d = any association), S is taken out, the rest is multiplied
    // back together -- no more products than the gate had, one fewer whenever S sat inside a nested product
    void flatten(int n, std::vector<int>& out) {
        if (is(n, N_MUL)) {
            flatten(nodes[n].a, out);
            flatten(nodes[n].b, out);
        } else if (is(n, N_SQR)) {
            flatten(nodes[n].a, out);
            flatten(nodes[n].a, out);
        } else {
            out.push_back(n);
        }
    }
    void factor_gates() {
        std::map<int, int> freq;
        std::vector<std::vector<int>> factors(d->n_value_parts);
        for (uint32_t i = 0; i < d->n_value_parts; i++) {
            flatten(terms[i].x, factors[i]);
            if (factors[i].size() < 2) continue;
            std::vector<int> seen;
            for (int f : factors[i])
                if (is(f, N_LOAD) && std::find(seen.begin(), seen.end(), f) == seen.end()) {
                    seen.push_back(f);
                    freq[f]++;
                }
        }
        for (uint32_t i = 0; i < d->n_value_parts; i++) {
            if (factors[i].size() < 2) continue;
            int best = -1;
            for (int f : factors[i]) {
                if (!is(f, N_LOAD) || freq[f] < 2) continue;
                if (best < 0 || freq[f] > freq[best] ||
                    (freq[f] == freq[best] && cols[nodes[f].x].table == T_FIXED && cols[nodes[best].x].table != T_FIXED))
                    best = f;
            }
            if (best < 0) continue;
            int rest = -1;
            bool taken = false;
            for (int f : factors[i]) {
                if (f == best && !taken) {
                    taken = true;
                    continue;
                }
                rest = rest < 0 ? f : mul(rest, f);
            }
            terms[i] = Term{rest, best};
        }
    }

    // how many nodes / terms consume each node (a product consumed once can be fused into its consumer: fp_mul2)
    std::vector<int> uses_;
    void count_uses() {
        uses_.assign(nodes.size(), 0);
        std::vector<char> seen(nodes.size(), 0);
        std::vector<int> stack;
        auto root = [&](int n) {
            if (n < 0) return;
            uses_[n]++;
            if (!seen[n]) {
                seen[n] = 1;
                stack.push_back(n);
            }
        };
        for (const Term& t : terms) {
            root(t.x);
            root(t.f);
        }
        while (!stack.empty()) {
            const int n = stack.back();
            stack.pop_back();
            const Node& nd = nodes[n];
            for (int c : {nd.a, nd.b}) {
                if (c < 0) continue;
                uses_[c]++;
                if (!seen[c]) {
                    seen[c] = 1;
                    stack.push_back(c);
                }
            }
        }
    }
    int uses(int n) const { return n >= 0 && (size_t)n < uses_.size() ? uses_[n] : 2; }   // (nodes made later: never fused)

    int need(int n) {
        if (need_.size() < nodes.size()) need_.resize(nodes.size(), -1);
        if (need_[n] >= 0) return need_[n];
        const Node& nd = nodes[n];
        int r;
        switch (nd.op) {
            case N_ZERO: case N_SCALAR: r = 0; break;
            case N_LOAD: case N_OMEGA: r = 1; break;
            case N_NEG: case N_SQR: case N_DBL: r = std::max(need(nd.a), 1); break;
            default: {
                const int x = need(nd.a), y = need(nd.b);
                r = x == y ? x + 1 : std::max(x, y);
            }
        }
        return need_[n] = r;
    }
};

// ------------------------------------------------------------------------------------------------ one stage
struct Operand {
    enum Kind { VAR, LOAD, TEXT } kind = TEXT;
    int id = -1;       // VAR: variable; LOAD: load key
    std::string text;  // TEXT
};

struct Stmt {
    int def;  // the variable it defines
    std::string fmt;  // "@0", "@1", ... stand for the operands
    std::vector<Operand> args;
    uint32_t products;
};

struct StageEmitter {
    Builder& B;
    const Options& opt;
    std::vector<Stmt> stmts;
    std::map<int, int> var_of;  // node -> variable
    int nvars = 0;
    std::vector<ColRef> cols;
    std::map<int, int> col_slot;  // program column id -> Args::cols slot
    std::vector<ScalarRef> scalars;
    std::map<int, int> scalar_slot;
    static constexpr int CONST_KEY = 1 << 20;    // load keys (-(CONST_KEY + slot), 0): entry `slot` of the module's constant table
    std::vector<uint32_t> consts;                // constants[] indices the stage reads, in table order
    std::map<uint32_t, int> const_slot;
    std::vector<std::pair<int, int>> load_keys;  // (cols slot, rotation); (-(scalar slot + 1), 0) for a uniform scalar
    std::map<std::pair<int, int>, int> load_key_id;
    uint32_t products = 0, n_terms = 0;
    bool uses_omega = false;
    bool has_total = false;
    Operand total;
    // the open group
    bool in_group = false, group_empty = true;
    int group_f = -1;
    Operand S;
    uint32_t e_prev = 0;

    StageEmitter(Builder& b, const Options& o) : B(b), opt(o) {}

    Operand text(const std::string& t) {
        Operand o;
        o.kind = Operand::TEXT;
        o.text = t;
        return o;
    }
    Operand stmt(const std::string& fmt, std::vector<Operand> args, uint32_t nprod) {
        Stmt s{nvars++, fmt, std::move(args), nprod};
        products += nprod;
        stmts.push_back(std::move(s));
        Operand o;
        o.kind = Operand::VAR;
        o.id = stmts.back().def;
        return o;
    }
    // a value that comes from memory: one load key per (source, rotation), interned
    Operand load_operand(const std::pair<int, int>& key) {
        Operand o;
        o.kind = Operand::LOAD;
        auto kt = load_key_id.find(key);
        if (kt == load_key_id.end()) {
            o.id = (int)load_keys.size();
            load_keys.push_back(key);
            load_key_id[key] = o.id;
        } else {
            o.id = kt->second;
        }
        return o;
    }
    Operand scalar_operand(int scalar_id) {
        if (B.scalars[scalar_id].kind == SC_CONST) {
            // a constant of the circuit is part of the program (and of its hash): it lives in the code object as initialised
            // data (`h2_consts`), read like any other value where the schedule wants it -- not a kernel argument: a gate set with
            // a constant per gate would fill the 4 KiB of arguments with them and be cut into stages for that alone
            const uint32_t index = B.scalars[scalar_id].arg;
            auto ct = const_slot.find(index);
            int slot;
            if (ct == const_slot.end()) {
                slot = (int)consts.size();
                consts.push_back(index);
                const_slot[index] = slot;
            } else {
                slot = ct->second;
            }
            return load_operand(std::make_pair(-(CONST_KEY + slot), 0));
        }
        auto it = scalar_slot.find(scalar_id);
        int slot;
        if (it == scalar_slot.end()) {
            slot = (int)scalars.size();
            scalars.push_back(B.scalars[scalar_id]);
            scalar_slot[scalar_id] = slot;
        } else {
            slot = it->second;
        }
        // a uniform scalar is a LOAD like a column value -- from the kernel arguments or their LDS copy (finish() decides),
        // value-numbered, kept within the live budget, issued a group ahead: key (-(slot + 1), 0)
        return load_operand(std::make_pair(-(slot + 1), 0));
    }
    Operand ypow(uint32_t e) {
        const int n = B.scalar(SC_Y_POW, e);
        return scalar_operand((int)B.nodes[n].x);
    }
    // a*b + c*d under ONE Montgomery reduction (fp_mul2: 128 operand multiply-adds + 64 of reduction instead of 2 x 128): for a
    // product that nothing else reads and that has not been emitted yet
    uint32_t fused = 0;
    bool fusible(int n) const {
        if (!opt.mul2 || n < 0 || fused >= opt.mul2) return false;
        const Node& nd = B.nodes[n];
        return (nd.op == N_MUL || nd.op == N_SQR) && B.uses(n) == 1 && !var_of.count(n);
    }
    void product_operands(int n, Operand& a, Operand& b) {
        const Node nd = B.nodes[n];
        a = emit(nd.a);
        b = nd.op == N_SQR ? a : emit(nd.b);
    }
    Operand emit(int n) {
        const Node nd = B.nodes[n];
        switch (nd.op) {
            case N_ZERO: return text("fp_zero<FrParams>()");
            case N_SCALAR: return scalar_operand((int)nd.x);
            case N_LOAD: {
                auto it = col_slot.find((int)nd.x);
                int slot;
                if (it == col_slot.end()) {
                    slot = (int)cols.size();
                    cols.push_back(B.cols[nd.x]);
                    col_slot[(int)nd.x] = slot;
                } else {
                    slot = it->second;
                }
                return load_operand(std::make_pair(slot, (int)nd.y));
            }
            default: break;
        }
        auto vt = var_of.find(n);
        if (vt != var_of.end()) {
            Operand o;
            o.kind = Operand::VAR;
            o.id = vt->second;
            return o;
        }
        Operand r;
        if (nd.op == N_OMEGA) {
            // extended_omega^idx from the NTT plan's two-level power table (k_evalh_perm's beta_term)
            uses_omega = true;
            r = stmt("(a.extended_k <= 12) ? fp_load(a.tw_lo + idx) : jmul(fp_load(a.tw_lo + (idx & 4095)), fp_load(a.tw_hi + (idx >> 12)))", {}, 1);
        } else if (nd.op == N_NEG || nd.op == N_SQR || nd.op == N_DBL) {
            Operand a = emit(nd.a);
            r = stmt(nd.op == N_NEG ? "fp_neg(@0)" : (nd.op == N_SQR ? "jsqr(@0)" : "fp_dbl(@0)"), {a}, nd.op == N_SQR ? 1 : 0);
        } else if ((nd.op == N_ADD || nd.op == N_SUB) && fusible(nd.a) && fusible(nd.b)) {
            Operand a, b, c, d;
            product_operands(nd.a, a, b);
            product_operands(nd.b, c, d);
            r = stmt(nd.op == N_ADD ? "jmul2(@0, @1, @2, @3)" : "jmul2(@0, @1, fp_neg(@2), @3)", {a, b, c, d}, 2);
            fused++;
        } else {
            // the operand that needs more registers first (Sethi-Ullman); program order on a tie
            Operand a, b;
            if (B.need(nd.b) > B.need(nd.a)) {
                b = emit(nd.b);
                a = emit(nd.a);
            } else {
                a = emit(nd.a);
                b = emit(nd.b);
            }
            r = stmt(nd.op == N_ADD ? "fp_add(@0, @1)" : (nd.op == N_SUB ? "fp_sub(@0, @1)" : "jmul(@0, @1)"), {a, b}, nd.op == N_MUL ? 1 : 0);
        }
        var_of[n] = r.id;
        return r;
    }

    // ---- groups: sum_{j in G} X_j y^e_j as a Horner chain in the gaps between the exponents, closed by y^e_last and F
    void begin_group(int f) {
        in_group = true;
        group_empty = true;
        group_f = f;
    }
    void add_member(int x, uint32_t e) {
        if (group_empty) {
            S = emit(x);
            group_empty = false;
        } else {
            // S y^gap + X: when X is (or ends in) a product nobody else reads, the Horner step and that product share a reduction
            const Node nx = B.nodes[x];
            const Operand Y = ypow(e_prev - e);
            Operand p, q;
            if (fusible(x)) {
                product_operands(x, p, q);
                S = stmt("jmul2(@0, @1, @2, @3)", {S, Y, p, q}, 2);
                fused++;
            } else if ((nx.op == N_ADD || nx.op == N_SUB) && B.uses(x) == 1 && !var_of.count(x) && (fusible(nx.a) || fusible(nx.b))) {
                const bool first = fusible(nx.a);           // X = P (+/-) R  or  R (+/-) P, P the product
                const Operand R = emit(first ? nx.b : nx.a);
                product_operands(first ? nx.a : nx.b, p, q);
                if (nx.op == N_ADD) S = stmt("fp_add(jmul2(@0, @1, @2, @3), @4)", {S, Y, p, q, R}, 2);
                else if (first) S = stmt("fp_sub(jmul2(@0, @1, @2, @3), @4)", {S, Y, p, q, R}, 2);
                else S = stmt("fp_add(jmul2(@0, @1, fp_neg(@2), @3), @4)", {S, Y, p, q, R}, 2);
                fused++;
            } else {
                const Operand X = emit(x);
                S = stmt("fp_add(jmul(@0, @1), @2)", {S, Y, X}, 1);
            }
        }
        e_prev = e;
        n_terms++;
    }
    void end_group() {
        if (!in_group) return;
        in_group = false;
        if (group_empty) return;
        Operand C = S;
        if (e_prev > 0) C = stmt("jmul(@0, @1)", {C, ypow(e_prev)}, 1);
        if (group_f >= 0) C = stmt("jmul(@0, @1)", {C, emit(group_f)}, 1);
        total = has_total ? stmt("fp_add(@0, @1)", {total, C}, 0) : C;
        has_total = true;
    }
    size_t args_bytes() const { return ARGS_FIXED_BYTES + 32 * std::max<size_t>(scalars.size(), 1) + 8 * std::max<size_t>(cols.size(), 1); }

    bool args_in_lds = false;   // decided in finish(): see Options::lds_args
    std::string load_text(const std::pair<int, int>& key) const {
        if (key.first <= -CONST_KEY) return "fp_load(h2_consts + " + std::to_string(-key.first - CONST_KEY) + ")";
        if (key.first < 0) return std::string(args_in_lds ? "fp_load(sh_sc + " : "fp_load(a.sc + ") + std::to_string(-key.first - 1) + ")";
        return std::string(args_in_lds ? "fp_load(sh_cols[" : "fp_load(a.cols[") + std::to_string(key.first) + "] + " + rot_var(key.second) + ")";
    }
    static std::string rot_var(int rot) {
        if (rot == 0) return "idx";
        return rot > 0 ? "rp" + std::to_string(rot) : "rm" + std::to_string(-rot);
    }

    // place the loads (a (column, rotation) or scalar is loaded once for as long as it is kept -- see below -- and issued at the
    // start of the scheduling group BEFORE the one that first needs it) and write the translation unit
    void finish(bool accumulate, uint32_t stage_index, uint32_t stage_count, Stage& out) {
        const uint32_t n = (uint32_t)stmts.size();
        // Where the lanes read the argument block from.  Field by field from the kernel arguments (scalar loads, SGPR-based
        // addressing) is the fastest form -- 3-4 % on mini-PLONK and the wide circuit -- as long as everything fits: the
        // compiler loads EVERY argument it will ever need at the kernel's entry and parks what the ~100 SGPRs cannot hold in
        // AGPRs and VGPR lanes for the whole kernel (a 40-gate set with a constant per gate: 474 dwords -> 464 registers, one
        // wave per SIMD, scratch).  Above `lds_args` dwords of scalars + pointers the workgroup copies the block to LDS first
        // (a lane-indexed copy the compiler cannot take apart) and each value is read where the schedule wants it.
        const size_t arg_dwords = 8 * scalars.size() + 2 * cols.size();
        args_in_lds = opt.lds_args == 1 || (opt.lds_args > 1 && arg_dwords > opt.lds_args);
        std::vector<std::vector<uint32_t>> uses(load_keys.size());
        for (uint32_t i = 0; i < n; i++)
            for (const Operand& o : stmts[i].args)
                if (o.kind == Operand::LOAD && (uses[o.id].empty() || uses[o.id].back() != i)) uses[o.id].push_back(i);
        // A loaded value is a register octet for as long as it is kept: value numbering must not outgrow the register file
        // (a gate set that reads 30 columns at three rotations in no particular order would keep them all; the compiler then
        // spills or drops to one wave per SIMD).  One forward sweep: a value stays in its variable while it is used again
        // within `gap` statements AND the values alive at that point -- the statements' own results plus the loads kept --
        // fit `live_budget`; beyond it the kept load whose next use is farthest away is dropped and loaded again there
        // (Belady's rule; a second load of a line this lane read a few statements ago hits the cache).
        struct LoadVar { int key; uint32_t first, last; std::string name; };
        std::vector<LoadVar> lvars;
        std::map<std::pair<int, uint32_t>, int> lvar_at;  // (key, statement) -> load variable
        {
            std::vector<uint32_t> last_use(nvars, 0);
            for (uint32_t i = 0; i < n; i++)
                for (const Operand& o : stmts[i].args)
                    if (o.kind == Operand::VAR) last_use[o.id] = i;
            if (has_total && total.kind == Operand::VAR) last_use[total.id] = n;
            std::vector<int> delta(n + 2, 0);
            for (uint32_t i = 0; i < n; i++) {
                delta[i] += 1;
                delta[std::max(last_use[stmts[i].def], i) + 1] -= 1;
            }
            std::vector<int> ssa_live(n + 1, 0);
            for (uint32_t i = 0; i <= n; i++) ssa_live[i] = (i ? ssa_live[i - 1] : 0) + delta[i];
            std::vector<size_t> next_at(uses.size(), 0);   // per key: position in uses[key] of its next use
            std::map<int, int> kept;                        // key -> load variable currently holding it
            const int reserve = (int)std::min<uint32_t>(opt.max_ahead, 4);  // loads issued ahead of their group
            for (uint32_t i = 0; i < n; i++) {
                for (const Operand& o : stmts[i].args) {
                    if (o.kind != Operand::LOAD) continue;
                    const int k = o.id;
                    if (lvar_at.count(std::make_pair(k, i))) continue;
                    auto it = kept.find(k);
                    int lv;
                    if (it == kept.end()) {
                        lv = (int)lvars.size();
                        lvars.push_back(LoadVar{k, i, i, "v" + std::to_string(lvars.size())});
                        kept[k] = lv;
                    } else {
                        lv = it->second;
                    }
                    lvars[lv].last = i;
                    lvar_at[std::make_pair(k, i)] = lv;
                    while (next_at[k] < uses[k].size() && uses[k][next_at[k]] <= i) next_at[k]++;
                }
                // what is kept past statement i
                for (auto it = kept.begin(); it != kept.end();) {
                    const int k = it->first;
                    const bool again = next_at[k] < uses[k].size();
                    if (!again || uses[k][next_at[k]] - i > opt.gap) it = kept.erase(it);
                    else ++it;
                }
                while (!kept.empty() && ssa_live[i + 1] + (int)kept.size() + reserve > (int)opt.live_budget) {
                    auto far = kept.begin();
                    for (auto it = kept.begin(); it != kept.end(); ++it)
                        if (uses[it->first][next_at[it->first]] > uses[far->first][next_at[far->first]]) far = it;
                    kept.erase(far);
                }
            }
        }
        std::vector<int> order(lvars.size());
        for (size_t i = 0; i < order.size(); i++) order[i] = (int)i;
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return lvars[x].first < lvars[y].first; });
        const uint32_t G = std::max<uint32_t>(opt.group, 1);
        std::map<uint32_t, std::vector<int>> issue;  // group -> load variables issued at its start
        std::vector<uint32_t> issued_at(lvars.size());
        for (int lv : order) {
            uint32_t grp = lvars[lv].first / G;
            grp = grp ? grp - 1 : 0;
            if (issue[grp].size() >= opt.max_ahead) grp = lvars[lv].first / G;  // a full look-ahead window: with its own group
            issue[grp].push_back(lv);
            issued_at[lv] = grp * G;
        }
        // ---- the generator's own estimate of the live field values (the compiler's allocation is read back after hipRTC)
        {
            std::vector<uint32_t> last_use(nvars, 0);
            for (uint32_t i = 0; i < n; i++)
                for (const Operand& o : stmts[i].args)
                    if (o.kind == Operand::VAR) last_use[o.id] = i;
            if (has_total && total.kind == Operand::VAR) last_use[total.id] = n;
            std::vector<int> delta(n + 2, 0);
            for (uint32_t i = 0; i < n; i++) {
                delta[i] += 1;
                delta[std::max(last_use[stmts[i].def], i) + 1] -= 1;
            }
            for (size_t lv = 0; lv < lvars.size(); lv++) {
                delta[issued_at[lv]] += 1;
                delta[lvars[lv].last + 1] -= 1;
            }
            int live = 0, peak = 0;
            for (uint32_t i = 0; i <= n; i++) {
                live += delta[i];
                peak = std::max(peak, live);
            }
            out.max_live = (uint32_t)peak;
        }
        std::string body;
        std::vector<int> rots;
        for (auto& key : load_keys)
            if (key.second != 0 && std::find(rots.begin(), rots.end(), key.second) == rots.end()) rots.push_back(key.second);
        std::sort(rots.begin(), rots.end());
        for (int r : rots)
            body += "        const size_t " + rot_var(r) + " = (size_t)(((long long)idx + (long long)(" + std::to_string(r) +
                    ") * (long long)a.rot_scale) & mask);\n";
        auto substitute = [&](const Stmt& s, uint32_t i) {
            std::string t;
            for (size_t p = 0; p < s.fmt.size(); p++) {
                if (s.fmt[p] == '@') {
                    const Operand& o = s.args[s.fmt[p + 1] - '0'];
                    if (o.kind == Operand::VAR) t += "x" + std::to_string(o.id);
                    else if (o.kind == Operand::LOAD) t += lvars[lvar_at[std::make_pair(o.id, i)]].name;
                    else t += o.text;
                    p++;
                } else {
                    t += s.fmt[p];
                }
            }
            return t;
        };
        for (uint32_t i = 0; i < n; i++) {
            if (i % G == 0) {
                auto it = issue.find(i / G);
                if (it != issue.end())
                    for (int lv : it->second) {
                        const auto& key = load_keys[lvars[lv].key];
                        body += "        const Fr " + lvars[lv].name + " = " + load_text(key) + ";\n";
                    }
                body += "        __builtin_amdgcn_sched_barrier(0);\n";
            }
            body += "        const Fr x" + std::to_string(stmts[i].def) + " = " + substitute(stmts[i], i) + "; __builtin_amdgcn_sched_barrier(0);\n";
        }
        // a term list that reduces to one loaded value or a scalar (no statement at all) still has to be materialised
        std::string result;
        if (!has_total) {
            result = "fp_zero<FrParams>()";
        } else if (total.kind == Operand::VAR) {
            result = "x" + std::to_string(total.id);
        } else if (total.kind == Operand::LOAD) {
            const auto& key = load_keys[total.id];
            result = load_text(key);
            if (key.second != 0 && std::find(rots.begin(), rots.end(), key.second) == rots.end())
                body = "        const size_t " + rot_var(key.second) + " = (size_t)(((long long)idx + (long long)(" + std::to_string(key.second) +
                       ") * (long long)a.rot_scale) & mask);\n" + body;
        } else {
            result = total.text;
        }
        if (accumulate) result = "fp_add(fp_load(a.values + idx), " + result + ")";
        body += "        fp_store(a.values + idx, " + result + ");\n";

        const bool inline_mul = products <= opt.inline_muls;
        const size_t ns = std::max<size_t>(scalars.size(), 1), nc = std::max<size_t>(cols.size(), 1);
        std::string src;
        src += "// generated by libhalo2_hip.so (csrc/evalh_gen.cpp): evaluate_h of one circuit as straight-line code, stage " +
               std::to_string(stage_index + 1) + " of " + std::to_string(stage_count) + "\n";
        src += "// " + std::to_string(n_terms) + " terms, " + std::to_string(products) + " products per row, " + std::to_string(cols.size()) +
               " vectors read, " + std::to_string(scalars.size()) + " uniform scalars, loads issued one group ahead\n";
        src += "#include \"field.hpp\"\nusing namespace h2;\n\n";
        src += "struct Args {\n    Fr* values;\n    const Fr* tw_lo;\n    const Fr* tw_hi;\n    unsigned long long row_begin, row_end;\n"
               "    unsigned int extended_k, rot_scale;\n    Fr sc[" + std::to_string(ns) + "];\n    const Fr* cols[" + std::to_string(nc) + "];\n};\n\n";
        if (!consts.empty()) {
            src += "// the circuit's constants this stage reads (Montgomery form)\n__device__ Fr h2_consts[" + std::to_string(consts.size()) + "] = {\n";
            for (uint32_t index : consts) {
                const uint64_t* v = B.d->constants + 4 * (size_t)index;
                char buf[160];
                snprintf(buf, sizeof buf, "    {{0x%08xu, 0x%08xu, 0x%08xu, 0x%08xu, 0x%08xu, 0x%08xu, 0x%08xu, 0x%08xu}},\n", (uint32_t)v[0],
                         (uint32_t)(v[0] >> 32), (uint32_t)v[1], (uint32_t)(v[1] >> 32), (uint32_t)v[2], (uint32_t)(v[2] >> 32),
                         (uint32_t)v[3], (uint32_t)(v[3] >> 32));
                src += buf;
            }
            src += "};\n\n";
        }
        if (inline_mul) {
            src += "__device__ __forceinline__ Fr jmul(const Fr& x, const Fr& y) { return fp_mul(x, y); }\n"
                   "__device__ __forceinline__ Fr jsqr(const Fr& x) { return fp_sqr(x); }\n"
                   "__device__ __forceinline__ Fr jmul2(const Fr& x, const Fr& y, const Fr& z, const Fr& w) { return fp_mul2(x, y, z, w); }\n\n";
        } else {
            src += "// one out-of-line multiplier: the kernel stays a few instructions per product instead of ~460\n"
                   "__device__ __noinline__ Fr jmul(Fr x, Fr y) { return fp_mul(x, y); }\n"
                   "__device__ __forceinline__ Fr jsqr(const Fr& x) { return jmul(x, x); }\n";
            if (fused)
                // INLINE: four operands fill v0-v31, the schedule needs 67 registers, so an out-of-line fp_mul2 saves and
                // restores 16 callee-saved VGPRs through scratch on every call -- more than the 64 multiply-adds it spares
                src += "// x y + z w under one reduction (fp_mul2): three quarters of the multiply-adds of two products\n"
                       "__device__ __forceinline__ Fr jmul2(const Fr& x, const Fr& y, const Fr& z, const Fr& w) { return fp_mul2(x, y, z, w); }\n";
            src += "\n";
        }
        const std::string waves = opt.waves ? " __attribute__((amdgpu_waves_per_eu(" + std::to_string(opt.waves) + ")))" : "";
        src += "extern \"C\" __global__ void __launch_bounds__(256)" + waves + " " + std::string(KERNEL_NAME) + "(Args a) {\n"
               "    // ONE row per lane, no loop: inside a loop the compiler hoists the vector copies of every uniform scalar operand\n"
               "    // (loop-invariant) and keeps them all alive -- 8 registers per scalar, 350 for a gate set with 40 constants\n"
               "    const size_t size = (size_t)1 << a.extended_k;\n"
               "    const long long mask = (long long)size - 1;\n"
               + (args_in_lds ?
               "    // the argument block goes to LDS first, through a lane-indexed copy (evalh_gen.cpp, StageEmitter::finish)\n"
               "    __shared__ Fr sh_sc[" + std::to_string(ns) + "];\n"
               "    __shared__ const Fr* sh_cols[" + std::to_string(nc) + "];\n"
               "    for (unsigned t = threadIdx.x; t < " + std::to_string(ns) + "u; t += 256) fp_store(sh_sc + t, fp_load(a.sc + t));\n"
               "    for (unsigned t = threadIdx.x; t < " + std::to_string(nc) + "u; t += 256) sh_cols[t] = a.cols[t];\n"
               "    __syncthreads();\n" : std::string()) +
               "    const size_t idx = a.row_begin + (size_t)blockIdx.x * 256 + threadIdx.x;\n"
               "    if (idx >= a.row_end) return;\n"
               "    {\n";
        src += body;
        src += "    }\n}\n";
        out.source = std::move(src);
        out.cols = cols;
        out.scalars = scalars;
        out.accumulate = accumulate;
        out.uses_omega = uses_omega;
        out.products = products;
        out.fused_pairs = fused;
        out.statements = n;
    }
};

uint32_t env_u32(const char* name, uint32_t dflt) {
    const char* v = getenv(name);
    return v && *v ? (uint32_t)strtoul(v, nullptr, 10) : dflt;
}

}  // namespace

Options Options::from_env() {
    Options o;
    o.group = std::max<uint32_t>(env_u32("H2_JIT_GROUP", o.group), 1);
    o.max_ahead = env_u32("H2_JIT_MAX_AHEAD", o.max_ahead);
    o.gap = env_u32("H2_JIT_GAP", o.gap);
    o.live_budget = std::max<uint32_t>(env_u32("H2_JIT_LIVE", o.live_budget), 4);
    o.lds_args = env_u32("H2_JIT_LDS_ARGS", o.lds_args);
    o.mul2 = env_u32("H2_JIT_MUL2", o.mul2);
    o.min_group = env_u32("H2_JIT_MIN_GROUP", o.min_group);
    o.inline_muls = env_u32("H2_JIT_INLINE_MULS", o.inline_muls);
    o.stage_products = env_u32("H2_JIT_STAGE_PRODUCTS", o.stage_products);
    o.max_regs = env_u32("H2_JIT_MAX_REGS", o.max_regs);
    o.waves = env_u32("H2_JIT_WAVES", o.waves);
    o.factor = env_u32("H2_JIT_FACTOR", 1) != 0;
    return o;
}

Generated generate(const h2_evalh_desc* d, const Options& opt) {
    if (!d) fail("null descriptor");
    Builder B(d, opt);
    B.build();
    const uint32_t N = (uint32_t)B.terms.size();
    // groups in order of first appearance; without a factor: one group, the plain Horner fold
    struct Group { int f; std::vector<uint32_t> members; };
    std::vector<Group> groups;
    {
        std::map<int, size_t> where;
        for (uint32_t j = 0; j < N; j++) {
            const int f = B.terms[j].f;
            auto it = where.find(f);
            if (it == where.end()) {
                where[f] = groups.size();
                groups.push_back(Group{f, {j}});
            } else {
                groups[it->second].members.push_back(j);
            }
        }
    }
    Generated out;
    out.terms = N;
    out.reference_products_per_row = B.ref_products;
    out.vectors_read = 0;
    std::vector<StageEmitter> emitters;
    emitters.emplace_back(B, opt);
    auto stage_full = [&](const StageEmitter& e) {
        if (opt.stage_products && e.products >= opt.stage_products) return true;
        return e.args_bytes() > 3072 || e.cols.size() >= opt.max_cols;
    };
    for (const Group& g : groups) {
        emitters.back().begin_group(g.f);
        for (size_t m = 0; m < g.members.size(); m++) {
            const uint32_t j = g.members[m];
            emitters.back().add_member(B.terms[j].x, N - 1 - j);
            if (stage_full(emitters.back()) && !(m + 1 == g.members.size() && &g == &groups.back())) {
                emitters.back().end_group();
                emitters.emplace_back(B, opt);
                if (m + 1 < g.members.size()) emitters.back().begin_group(g.f);
            }
        }
        emitters.back().end_group();
    }
    if (emitters.size() > 1 && !emitters.back().has_total && emitters.back().stmts.empty()) emitters.pop_back();
    out.stages.resize(emitters.size());
    std::vector<bool> seen(B.cols.size(), false);
    for (size_t s = 0; s < emitters.size(); s++) {
        if (emitters[s].args_bytes() > 4096) fail("a single term reads more columns than one kernel's arguments hold");
        emitters[s].finish(s > 0, (uint32_t)s, (uint32_t)emitters.size(), out.stages[s]);
        out.products_per_row += out.stages[s].products;
        out.fused_pairs_per_row += out.stages[s].fused_pairs;
        for (auto& kv : emitters[s].col_slot)
            if (!seen[kv.first]) {
                seen[kv.first] = true;
                out.vectors_read++;
            }
    }
    return out;
}

void program_hash(const h2_evalh_desc* d, const Options& opt, uint8_t out[32]) {
    Sha256 h;
    h.update(GENERATOR_VERSION, strlen(GENERATOR_VERSION));
    h.update(h2_embed_field_hpp, strlen(h2_embed_field_hpp));
    h.update(h2_embed_fp_mul_gen_hpp, strlen(h2_embed_fp_mul_gen_hpp));
    const uint32_t o[] = {opt.group, opt.max_ahead, opt.gap, opt.inline_muls, opt.stage_products, opt.max_cols, opt.max_regs,
                          (uint32_t)opt.factor, opt.waves, opt.live_budget, opt.lds_args, (uint32_t)opt.mul2, opt.min_group};
    h.update(o, sizeof o);
    h.u32(d->blinding_factors);
    h.u32(d->chunk_len);
    h.u32(d->n_constants);
    h.update(d->constants, (size_t)d->n_constants * 32);
    h.u32(d->n_rotations);
    h.update(d->rotations, (size_t)d->n_rotations * 4);
    h.u32(d->n_calculations);
    h.update(d->calculations, (size_t)d->n_calculations * sizeof(h2_calculation));
    h.u32(d->n_value_parts);
    h.update(d->value_parts, (size_t)d->n_value_parts * sizeof(h2_value_source));
    h.u32(d->n_lookups);
    size_t n_lookup_calcs = 0;
    for (uint32_t t = 0; t < d->n_lookups; t++) n_lookup_calcs += 1 + 2 * (size_t)d->lookup_sets[t];
    h.update(d->lookup_sets, (size_t)d->n_lookups * 4);
    h.update(d->lookup_calcs, n_lookup_calcs * sizeof(h2_calculation));
    h.u32(d->n_shuffles);
    h.update(d->shuffle_calcs, 2 * (size_t)d->n_shuffles * sizeof(h2_calculation));
    h.u32(d->n_fixed);
    h.u32(d->n_advice);
    h.u32(d->n_instance);
    h.u32(d->n_perm_sets);
    h.u32(d->n_perm_columns);
    if (d->n_perm_sets) {
        h.update(d->perm_col_type, (size_t)d->n_perm_columns * 4);
        h.update(d->perm_col_index, (size_t)d->n_perm_columns * 4);
    }
    h.finish(out);
}

// ------------------------------------------------------------------------------------------------ hipRTC
namespace {

struct Rtc {
    void* handle = nullptr;
    int (*create)(void**, const char*, const char*, int, const char**, const char**) = nullptr;
    int (*compile)(void*, int, const char**) = nullptr;
    int (*log_size)(void*, size_t*) = nullptr;
    int (*log)(void*, char*) = nullptr;
    int (*code_size)(void*, size_t*) = nullptr;
    int (*code)(void*, char*) = nullptr;
    int (*destroy)(void**) = nullptr;
    int (*version)(int*, int*) = nullptr;
    std::string error;
};

Rtc& rtc() {
    static Rtc r;
    static std::once_flag once;
    std::call_once(once, [] {
        // H2_HIPRTC_LIB names the library to load instead of the default search (a test points it at a file that is not
        // there: the caller then keeps the interpreter kernels, with one warning)
        const char* forced = getenv("H2_HIPRTC_LIB");
        std::string why = "?";
        for (const char* name : {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"}) {
            r.handle = dlopen(forced && *forced ? forced : name, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
            const char* e = dlerror();  // (the call clears the message: read it once)
            if (e) why = e;
            if (forced && *forced) break;
        }
        if (!r.handle) {
            r.error = std::string("libhiprtc.so could not be loaded (") + why + ")";
            return;
        }
        auto sym = [&](const char* n) { return dlsym(r.handle, n); };
        r.create = (decltype(r.create))sym("hiprtcCreateProgram");
        r.compile = (decltype(r.compile))sym("hiprtcCompileProgram");
        r.log_size = (decltype(r.log_size))sym("hiprtcGetProgramLogSize");
        r.log = (decltype(r.log))sym("hiprtcGetProgramLog");
        r.code_size = (decltype(r.code_size))sym("hiprtcGetCodeSize");
        r.code = (decltype(r.code))sym("hiprtcGetCode");
        r.destroy = (decltype(r.destroy))sym("hiprtcDestroyProgram");
        r.version = (decltype(r.version))sym("hiprtcVersion");
        if (!r.create || !r.compile || !r.log_size || !r.log || !r.code_size || !r.code || !r.destroy) r.error = "libhiprtc.so lacks the hiprtc* entry points";
    });
    return r;
}

// the compiler a cached code object must come from: (major << 8 | minor) of hipRTC, 0 when it cannot be loaded or does not say
uint16_t rtc_version() {
    Rtc& r = rtc();
    int major = 0, minor = 0;
    if (!r.error.empty() || !r.version || r.version(&major, &minor) != 0) return 0;
    return (uint16_t)(((major & 0xff) << 8) | (minor & 0xff));
}

std::vector<char> rtc_compile(const std::string& source) {
    Rtc& r = rtc();
    if (!r.error.empty()) fail(r.error);
    const char* headers[] = {h2_embed_field_hpp, h2_embed_fp_mul_gen_hpp};
    const char* names[] = {"field.hpp", "fp_mul_gen.hpp"};
    void* prog = nullptr;
    if (r.create(&prog, source.c_str(), "h2_evalh_gen.hip", 2, headers, names) != 0) fail("hiprtcCreateProgram failed");
    const char* options[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
    const int rc = r.compile(prog, 3, options);
    if (rc != 0) {
        size_t n = 0;
        r.log_size(prog, &n);
        std::string log(n, '\0');
        if (n) r.log(prog, &log[0]);
        r.destroy(&prog);
        if (log.size() > 1500) log.resize(1500);
        fail("hipRTC rejected the generated source: " + log);
    }
    size_t n = 0;
    r.code_size(prog, &n);
    std::vector<char> code(n);
    r.code(prog, code.data());
    r.destroy(&prog);
    return code;
}

// `.vgpr_count` & co. of the kernel descriptor in the code object's msgpack metadata: <fixstr key><unsigned value>
uint32_t metadata_uint(const std::vector<char>& code, const char* key) {
    const size_t klen = strlen(key);
    uint32_t best = 0;
    for (size_t i = 0; i + klen + 2 < code.size(); i++) {
        const uint8_t tag = (uint8_t)code[i];
        const bool fix = klen < 32 && tag == (0xa0 | klen);
        const bool str8 = tag == 0xd9 && (uint8_t)code[i + 1] == klen;
        if (!fix && !str8) continue;
        const size_t at = i + (fix ? 1 : 2);
        if (at + klen >= code.size() || memcmp(&code[at], key, klen) != 0) continue;
        const uint8_t* v = (const uint8_t*)&code[at + klen];
        const size_t left = code.size() - (at + klen);
        uint32_t val = 0;
        if (v[0] < 0x80) val = v[0];
        else if (v[0] == 0xcc && left > 1) val = v[1];
        else if (v[0] == 0xcd && left > 2) val = (uint32_t)v[1] << 8 | v[2];
        else if (v[0] == 0xce && left > 4) val = (uint32_t)v[1] << 24 | (uint32_t)v[2] << 16 | (uint32_t)v[3] << 8 | v[4];
        else continue;
        best = std::max(best, val);
    }
    return best;
}

std::mutex g_dir_mu;
std::string g_private_dir;

bool is_private_dir(const std::string& dir) {
    struct stat st;
    if (lstat(dir.c_str(), &st) != 0) return false;
    return S_ISDIR(st.st_mode) && st.st_uid == getuid() && (st.st_mode & 022) == 0;
}

std::string hex(const uint8_t* p, size_t n) {
    static const char* digits = "0123456789abcdef";
    std::string s;
    for (size_t i = 0; i < n; i++) {
        s += digits[p[i] >> 4];
        s += digits[p[i] & 15];
    }
    return s;
}

bool read_file(const std::string& path, std::vector<char>& out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t)n : 0);
    const bool ok = n >= 0 && fread(out.data(), 1, out.size(), f) == out.size();
    fclose(f);
    return ok;
}

// file: magic, layout word, stage count, {length, code object} per stage, SHA-256 of everything before it
// file header: 6 bytes of magic + the hipRTC version the code objects were compiled by (2 bytes).  A file from ANOTHER compiler
// version is a miss when this process can compile (after a ROCm upgrade the objects are rebuilt, not kept for ever); a process
// without hipRTC takes what is there -- it could not rebuild it.
constexpr char CACHE_MAGIC[6] = {'H', '2', 'E', 'V', 'G', '3'};

}  // namespace

std::string cache_dir() {
    std::lock_guard<std::mutex> g(g_dir_mu);
    const char* env = getenv("H2_JIT_CACHE");
    std::string dir;
    if (env && *env) {
        dir = env;
    } else {
        const char* tmp = getenv("TMPDIR");
        dir = std::string(tmp && *tmp ? tmp : "/tmp") + "/halo2_hip_jit_" + std::to_string((unsigned)getuid());
    }
    mkdir(dir.c_str(), 0700);
    if (is_private_dir(dir)) {
        // once per process: temporary files of writers that were killed before their rename (older than an hour) go
        static bool swept = false;
        if (!swept) {
            swept = true;
            if (DIR* dp = opendir(dir.c_str())) {
                const time_t now = time(nullptr);
                while (struct dirent* e = readdir(dp)) {
                    if (!strstr(e->d_name, ".h2ev.tmp.")) continue;
                    const std::string f = dir + "/" + e->d_name;
                    struct stat st;
                    if (stat(f.c_str(), &st) == 0 && now - st.st_mtime > 3600) unlink(f.c_str());
                }
                closedir(dp);
            }
        }
        return dir;
    }
    if (g_private_dir.empty()) {
        fprintf(stderr, "libhalo2_hip: code-object cache directory %s is not private to this user: using a per-process directory\n", dir.c_str());
        const char* tmp = getenv("TMPDIR");
        std::string templ = std::string(tmp && *tmp ? tmp : "/tmp") + "/halo2_hip_jit_XXXXXX";
        std::vector<char> buf(templ.begin(), templ.end());
        buf.push_back('\0');
        if (!mkdtemp(buf.data())) fail("no private cache directory could be made");
        g_private_dir = buf.data();
        // a directory of this process alone: it goes when the process does (its files first)
        atexit([] {
            if (g_private_dir.empty()) return;
            if (DIR* dp = opendir(g_private_dir.c_str())) {
                while (struct dirent* e = readdir(dp))
                    if (strcmp(e->d_name, ".") && strcmp(e->d_name, "..")) unlink((g_private_dir + "/" + e->d_name).c_str());
                closedir(dp);
            }
            rmdir(g_private_dir.c_str());
        });
    }
    return g_private_dir;
}

Generated compile(const h2_evalh_desc* d, const Options& opt_in) {
    Options opt = opt_in;
    uint8_t hash[32];
    program_hash(d, opt_in, hash);
    const bool use_disk = env_u32("H2_JIT_DISK_CACHE", 1) != 0;
    const std::string path = use_disk ? cache_dir() + "/" + hex(hash, 16) + ".h2ev" : std::string();
    for (int attempt = 0;; attempt++) {
        Generated g = generate(d, opt);
        // ---- the disk cache holds the code objects of the FINAL stage layout under the hash of the requested options
        if (attempt == 0 && use_disk) {
            std::vector<char> blob;
            bool intact = read_file(path, blob) && blob.size() >= 16 + 32 && memcmp(blob.data(), CACHE_MAGIC, 6) == 0;
            if (intact) {
                uint16_t made_by = 0;
                memcpy(&made_by, &blob[6], 2);
                const uint16_t mine = rtc_version();
                if (mine != 0 && made_by != mine) intact = false;
            }
            if (intact) {   // a torn or damaged file is a miss (and is overwritten by the rebuild below)
                uint8_t sum[32];
                Sha256 hs;
                hs.update(blob.data(), blob.size() - 32);
                hs.finish(sum);
                intact = memcmp(sum, blob.data() + blob.size() - 32, 32) == 0;
                blob.resize(blob.size() - 32);
            }
            if (intact) {
                uint32_t stage_products = 0, nstages = 0;
                memcpy(&stage_products, &blob[8], 4);
                memcpy(&nstages, &blob[12], 4);
                Options o2 = opt;
                o2.stage_products = stage_products & 0x7fffffffu;   // (top bit: built with the argument block through LDS)
                if (stage_products >> 31) o2.lds_args = 1;
                Generated g2 = (o2.stage_products == opt.stage_products && o2.lds_args == opt.lds_args) ? std::move(g) : generate(d, o2);
                size_t at = 16;
                bool ok = nstages == g2.stages.size();
                for (uint32_t s = 0; ok && s < nstages; s++) {
                    uint32_t len = 0;
                    if (at + 4 > blob.size()) { ok = false; break; }
                    memcpy(&len, &blob[at], 4);
                    at += 4;
                    if (at + len > blob.size()) { ok = false; break; }
                    g2.stages[s].code.assign(blob.begin() + at, blob.begin() + at + len);
                    at += len;
                }
                if (ok && at == blob.size()) {
                    for (Stage& st : g2.stages) {
                        st.vgprs = metadata_uint(st.code, ".vgpr_count");
                        st.agprs = metadata_uint(st.code, ".agpr_count");
                        st.scratch = metadata_uint(st.code, ".private_segment_fixed_size");
                    }
                    g2.from_disk = true;
                    return g2;
                }
                g = generate(d, opt);
            }
        }
        uint32_t worst_products = 0;
        bool over = false;
        for (Stage& st : g.stages) {
            st.code = rtc_compile(st.source);
            st.vgprs = metadata_uint(st.code, ".vgpr_count");
            st.agprs = metadata_uint(st.code, ".agpr_count");
            st.scratch = metadata_uint(st.code, ".private_segment_fixed_size");
            if (st.vgprs + st.agprs > opt.max_regs || st.scratch) {
                over = true;
                worst_products = std::max(worst_products, st.products);
            }
        }
        // a stage the compiler could not keep within the register budget (one wave per SIMD, or spills): cut the
        // program into stages of half that many products and build again -- each stage keeps fewer values live
        if (over && opt.lds_args != 1 && attempt < 6) {
            opt.lds_args = 1;   // first remedy: the argument block through LDS (fewer registers, the same stages)
            continue;
        }
        if (over && attempt < 6 && worst_products >= 16) {
            opt.stage_products = (worst_products + 1) / 2;
            continue;
        }
        if (use_disk) {
            // (a name of its own per writer: h2_evalh_compile / h2_evalh_prepare may run on several threads at once, for the
            // same program too -- the last complete file renamed into place wins, and they are all the same bytes)
            static std::atomic<unsigned> writer{0};
            std::string tmp = path + ".tmp." + std::to_string((long)getpid()) + "." + std::to_string(writer.fetch_add(1));
            FILE* f = fopen(tmp.c_str(), "wb");
            if (f) {
                const uint32_t ns = (uint32_t)g.stages.size();
                const uint32_t layout = opt.stage_products | (opt.lds_args == 1 && opt_in.lds_args != 1 ? 0x80000000u : 0u);
                Sha256 hs;
                auto put = [&](const void* p_, size_t n_) {
                    hs.update(p_, n_);
                    return fwrite(p_, 1, n_, f) == n_;
                };
                const uint16_t made_by = rtc_version();
                bool ok = put(CACHE_MAGIC, 6) && put(&made_by, 2) && put(&layout, 4) && put(&ns, 4);
                for (const Stage& st : g.stages) {
                    const uint32_t len = (uint32_t)st.code.size();
                    ok = ok && put(&len, 4) && put(st.code.data(), len);
                }
                uint8_t sum[32];
                hs.finish(sum);
                ok = ok && fwrite(sum, 1, 32, f) == 32;
                ok = fclose(f) == 0 && ok;
                if (!ok || rename(tmp.c_str(), path.c_str()) != 0) unlink(tmp.c_str());
            }
        }
        return g;
    }
}

}  // namespace evgen
}  // namespace h2
