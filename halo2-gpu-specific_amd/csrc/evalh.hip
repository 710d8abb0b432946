// evalh.hip -- the quotient numerator h(X) on the extended coset: Evaluator::evaluate_h.
//
// Reference: CPU twin plonk/evaluation.rs:778-1226; cuda driver plonk/evaluation.rs:1229-1985, which
// walks ProveExpression trees launching one elementwise kernel per AST node (evaluation_gpu.rs:594-803)
// and re-derives extended cosets on demand through a 5-entry cache.  Here every coset stays resident
// (288 GB of HBM) and the whole gate program runs in ONE pass over the extended domain:
//   k_evalh_expr     per index: interpret the flattened `Calculation` program (evaluation.rs:95-266),
//                    Horner-accumulate the gate value parts in y (:891-901), and emit the lookup /
//                    shuffle compressed expressions (:903-997)
//   k_evalh_perm     permutation argument terms (:1004-1085; cuda kernels eval_h_permutation_part1/2/
//                    left_prepare/left_right/part3, :1341-1486) fused into one kernel
//   k_evalh_lookup   logup terms (:1138-1182; cuda eval_h_logup / _z / _extra, :1655-1788)
//   k_evalh_shuffle  shuffle terms (:1197-1219; cuda eval_h_shuffles, :1935-1952)
// Intermediates of the interpreter live in a [calculation][thread] global array (coalesced, L2-resident);
// the program itself is read with wave-uniform (scalar) loads, so there is no divergence.
// These four kernels are the fallback: a program is normally run as kernels GENERATED for it (evalh_gen.cpp: straight-line
// code, intermediates in registers, all terms in one or a few launches), built by hipRTC on first use -- `EvalhPlan` below.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <set>
#include <condition_variable>
#include <chrono>
#include <map>
#include <memory>
#include <stdexcept>
#include <thread>
#include <vector>

#include "common.hpp"
#include "evalh.hpp"
#include "msm.hpp"
#include "evalh_gen.hpp"
#include "ntt.hpp"
#include "poly.hpp"

namespace h2 {

struct EvalhProgram {  // everything the kernels need, device pointers
    const Fr* constants;
    const int32_t* rotations;
    const h2_calculation* calcs;
    const h2_value_source* value_parts;
    const h2_calculation* lookup_calcs;
    const uint32_t* lookup_sets;
    const h2_calculation* shuffle_calcs;
    const Fr* const* fixed;
    const Fr* const* advice;
    const Fr* const* instance;
    uint32_t n_calcs, n_value_parts, n_lookups, n_shuffles;
    uint32_t extended_k, rot_scale;
    size_t row_begin, row_end;   // rows to evaluate
    Fr y, beta, gamma, theta;
};

__device__ __forceinline__ size_t rot_idx(size_t idx, int32_t rot, uint32_t rot_scale, uint32_t extended_k) {
    // (idx + rot * rot_scale) mod 2^extended_k  (get_rotation_idx, evaluation.rs:40-42)
    long long v = (long long)idx + (long long)rot * (long long)rot_scale;
    return (size_t)(v & (((long long)1 << extended_k) - 1));
}

struct Interp {
    const EvalhProgram& p;
    const Fr* inter;  // this thread's column of the intermediates array
    size_t stride;    // distance between consecutive intermediates of one thread
    size_t idx;
    // the most recent intermediate stays in registers: expression trees flattened depth-first consume the previous
    // result in the very next calculation most of the time, which then skips the round trip through memory
    uint32_t last_index = 0xffffffffu;
    Fr last{};

    __device__ __forceinline__ Fr get(const h2_value_source& v) const {
        switch (v.kind) {
            case H2_VS_CONSTANT: return fp_load(p.constants + v.index);
            case H2_VS_INTERMEDIATE:
                if (v.index == last_index) return last;
                return fp_load(inter + (size_t)v.index * stride);
            case H2_VS_FIXED: return fp_load(p.fixed[v.index] + rot_idx(idx, p.rotations[v.rot], p.rot_scale, p.extended_k));
            case H2_VS_ADVICE: return fp_load(p.advice[v.index] + rot_idx(idx, p.rotations[v.rot], p.rot_scale, p.extended_k));
            default: return fp_load(p.instance[v.index] + rot_idx(idx, p.rotations[v.rot], p.rot_scale, p.extended_k));
        }
    }

    __device__ __forceinline__ Fr eval(const h2_calculation& k) const {
        Fr a = get(k.a);
        switch (k.op) {
            case H2_CALC_ADD: return fp_add(a, get(k.b));
            case H2_CALC_SUB: return fp_sub(a, get(k.b));
            case H2_CALC_MUL: return fp_mul(a, get(k.b));
            case H2_CALC_NEGATE: return fp_neg(a);
            case H2_CALC_LC_CHALLENGE: {
                Fr x = (k.challenge == H2_CHALLENGE_BETA) ? p.beta : p.gamma;
                if (k.power > 1) x = fp_pow_u32(x, k.power);
                return fp_mul(fp_add(a, x), get(k.b));
            }
            case H2_CALC_LC_THETA: return fp_add(fp_mul(a, p.theta), get(k.b));
            case H2_CALC_ADD_CHALLENGE: return fp_add(a, (k.challenge == H2_CHALLENGE_BETA) ? p.beta : p.gamma);
            default: return a;  // Store
        }
    }
};

// lookup tables layout: [slot][idx], slots per lookup t: table, product_0, sum_0, product_1, sum_1, ...
// (slot base of lookup t = sum_{t' < t} (1 + 2 * sets[t'])) -- the same order as lookup_calcs
__global__ void __launch_bounds__(256) k_evalh_expr(EvalhProgram p, Fr* inter, Fr* values, Fr* lk_out, Fr* sh_out) {
    const size_t size = (size_t)1 << p.extended_k;
    const size_t nthreads = (size_t)gridDim.x * blockDim.x;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fr* my = inter + t;
    for (size_t idx = p.row_begin + t; idx < p.row_end; idx += nthreads) {
        Interp in{p, my, nthreads, idx};
        for (uint32_t i = 0; i < p.n_calcs; i++) {
            Fr r = in.eval(p.calcs[i]);
            fp_store(my + (size_t)i * nthreads, r);
            in.last = r;
            in.last_index = i;
        }
        Fr value = fp_zero<FrParams>();
        for (uint32_t i = 0; i < p.n_value_parts; i++) value = fp_add(fp_mul(value, p.y), in.get(p.value_parts[i]));
        fp_store(values + idx, value);
        uint32_t slot = 0;
        for (uint32_t lk = 0; lk < p.n_lookups; lk++) {
            uint32_t cnt = 1 + 2 * p.lookup_sets[lk];
            for (uint32_t j = 0; j < cnt; j++, slot++) fp_store(lk_out + (size_t)slot * size + idx, in.eval(p.lookup_calcs[slot]));
        }
        for (uint32_t i = 0; i < 2 * p.n_shuffles; i++) fp_store(sh_out + (size_t)i * size + idx, in.eval(p.shuffle_calcs[i]));
    }
}

struct PermArgs {
    Fr* values;
    const Fr* const* perm_z;
    const Fr* const* perm_cols;   // resolved column coset per permutation column
    const Fr* const* perm_sigma;
    const Fr *l0, *l_last, *l_active_row;
    const Fr *tw_lo, *tw_hi;      // extended_omega^i tables (NTT plan)
    uint32_t n_sets, n_columns, chunk_len, extended_k, rot_scale;
    int32_t last_rotation;
    size_t row_begin, row_end;
    Fr y, beta, gamma, delta, delta_start;  // delta_start = beta * ZETA
};

__global__ void __launch_bounds__(256) k_evalh_perm(PermArgs a) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const Fr one = fp_one<FrParams>();
    for (size_t idx = a.row_begin + (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < a.row_end; idx += stride) {
        const size_t r_next = rot_idx(idx, 1, a.rot_scale, a.extended_k);
        const size_t r_last = rot_idx(idx, a.last_rotation, a.rot_scale, a.extended_k);
        Fr value = fp_load(a.values + idx);
        const Fr l0 = fp_load(a.l0 + idx);
        {   // l_0(X) * (1 - z_0(X))
            Fr z0 = fp_load(a.perm_z[0] + idx);
            value = fp_add(fp_mul(value, a.y), fp_mul(fp_sub(one, z0), l0));
            // l_last(X) * (z_l(X)^2 - z_l(X))
            Fr zl = fp_load(a.perm_z[a.n_sets - 1] + idx);
            value = fp_add(fp_mul(value, a.y), fp_mul(fp_sub(fp_sqr(zl), zl), fp_load(a.l_last + idx)));
        }
        for (uint32_t s = 1; s < a.n_sets; s++) {  // l_0(X) * (z_i(X) - z_{i-1}(w^last X))
            Fr d = fp_sub(fp_load(a.perm_z[s] + idx), fp_load(a.perm_z[s - 1] + r_last));
            value = fp_add(fp_mul(value, a.y), fp_mul(d, l0));
        }
        // beta_term = extended_omega^idx
        Fr beta_term;
        if (a.extended_k <= 12) {
            beta_term = fp_load(a.tw_lo + idx);
        } else {
            beta_term = fp_mul(fp_load(a.tw_lo + (idx & 4095)), fp_load(a.tw_hi + (idx >> 12)));
        }
        Fr current_delta = fp_mul(a.delta_start, beta_term);
        const Fr lar = fp_load(a.l_active_row + idx);
        for (uint32_t s = 0; s < a.n_sets; s++) {
            uint32_t c0 = s * a.chunk_len, c1 = c0 + a.chunk_len;
            if (c1 > a.n_columns) c1 = a.n_columns;
            Fr left = fp_load(a.perm_z[s] + r_next), right = fp_load(a.perm_z[s] + idx);
            for (uint32_t j = c0; j < c1; j++) {
                Fr v = fp_load(a.perm_cols[j] + idx);
                Fr sg = fp_load(a.perm_sigma[j] + idx);
                left = fp_mul(left, fp_add(fp_add(v, fp_mul(a.beta, sg)), a.gamma));
                right = fp_mul(right, fp_add(fp_add(v, current_delta), a.gamma));
                current_delta = fp_mul(current_delta, a.delta);
            }
            value = fp_add(fp_mul(value, a.y), fp_mul(fp_sub(left, right), lar));
        }
        fp_store(a.values + idx, value);
    }
}

struct LookupArgs {
    Fr* values;
    const Fr* const* zs;     // sets_len z cosets of this lookup
    const Fr* m;
    const Fr* table;         // lk_out slots of this lookup: table, product_0, sum_0, product_1, ...
    const Fr *l0, *l_last, *l_active_row;
    uint32_t sets_len, extended_k, rot_scale;
    int32_t last_rotation;
    size_t row_begin, row_end;
    Fr y;
};

__global__ void __launch_bounds__(256) k_evalh_lookup(LookupArgs a) {
    const size_t size = (size_t)1 << a.extended_k;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t idx = a.row_begin + (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < a.row_end; idx += stride) {
        const size_t r_next = rot_idx(idx, 1, a.rot_scale, a.extended_k);
        const size_t r_last = rot_idx(idx, a.last_rotation, a.rot_scale, a.extended_k);
        Fr value = fp_load(a.values + idx);
        const Fr l0 = fp_load(a.l0 + idx), lar = fp_load(a.l_active_row + idx);
        const Fr z0 = fp_load(a.zs[0] + idx);
        value = fp_add(fp_mul(value, a.y), fp_mul(z0, l0));
        value = fp_add(fp_mul(value, a.y), fp_mul(fp_load(a.zs[a.sets_len - 1] + idx), fp_load(a.l_last + idx)));
        {
            Fr table = fp_load(a.table + idx);
            Fr prod = fp_load(a.table + size + idx), sum = fp_load(a.table + 2 * size + idx);
            Fr d = fp_sub(fp_load(a.zs[0] + r_next), z0);
            Fr t = fp_mul(fp_add(fp_mul(d, table), fp_load(a.m + idx)), prod);
            t = fp_sub(t, fp_mul(table, sum));
            value = fp_add(fp_mul(value, a.y), fp_mul(t, lar));
        }
        for (uint32_t i = 1; i < a.sets_len; i++) {
            Fr d = fp_sub(fp_load(a.zs[i] + idx), fp_load(a.zs[i - 1] + r_last));
            value = fp_add(fp_mul(value, a.y), fp_mul(d, l0));
        }
        for (uint32_t i = 1; i < a.sets_len; i++) {
            Fr d = fp_sub(fp_load(a.zs[i] + r_next), fp_load(a.zs[i] + idx));
            Fr t = fp_sub(fp_mul(d, fp_load(a.table + (size_t)(1 + 2 * i) * size + idx)),
                          fp_load(a.table + (size_t)(2 + 2 * i) * size + idx));
            value = fp_add(fp_mul(value, a.y), fp_mul(t, lar));
        }
        fp_store(a.values + idx, value);
    }
}

struct ShuffleArgs {
    Fr* values;
    const Fr* z;
    const Fr *input, *shuffle;
    const Fr *l0, *l_last, *l_active_row;
    uint32_t extended_k, rot_scale;
    size_t row_begin, row_end;
    Fr y;
};

__global__ void __launch_bounds__(256) k_evalh_shuffle(ShuffleArgs a) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const Fr one = fp_one<FrParams>();
    for (size_t idx = a.row_begin + (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < a.row_end; idx += stride) {
        const size_t r_next = rot_idx(idx, 1, a.rot_scale, a.extended_k);
        Fr value = fp_load(a.values + idx);
        const Fr z = fp_load(a.z + idx);
        value = fp_add(fp_mul(value, a.y), fp_mul(fp_sub(one, z), fp_load(a.l0 + idx)));
        value = fp_add(fp_mul(value, a.y), fp_mul(fp_sub(fp_sqr(z), z), fp_load(a.l_last + idx)));
        Fr t = fp_sub(fp_mul(fp_load(a.z + r_next), fp_load(a.shuffle + idx)), fp_mul(z, fp_load(a.input + idx)));
        value = fp_add(fp_mul(value, a.y), fp_mul(t, fp_load(a.l_active_row + idx)));
        fp_store(a.values + idx, value);
    }
}


// ---------------------------------------------------------------- generated kernels: plans per (program, device)
struct EvalhPlan {
    evgen::Generated gen;               // stage metadata (sources and code objects dropped after loading)
    std::vector<hipModule_t> modules;
    std::vector<hipFunction_t> functions;
    int device = -1;
    bool uses_omega = false;
    int from_cache = 0;                 // 0 compiled now, 2 from the disk cache (1 = memory is decided by the caller)
    mutable uint64_t last_use = 0;      // g_plan_mu: the cache's clock at the last lookup
    mutable uint32_t users = 0;         // g_plan_mu: EvalhPlanRef holders -- a held plan is never evicted
};

namespace {
struct PlanKey {
    uint64_t h[2];
    int device;
    bool operator<(const PlanKey& o) const {
        if (h[0] != o.h[0]) return h[0] < o.h[0];
        if (h[1] != o.h[1]) return h[1] < o.h[1];
        return device < o.device;
    }
};
std::mutex g_plan_mu;
std::map<PlanKey, std::unique_ptr<EvalhPlan>> g_plans;   // a null entry: generation failed for this program (said once)
// programs some caller is generating / compiling / loading right now, OUTSIDE g_plan_mu (a hipRTC build takes seconds): callers
// of the SAME program wait on the condition variable instead of compiling it again; lookups of every other program -- cache
// hits included, on any device or host-API slot -- go on meanwhile
std::set<PlanKey> g_building;
std::condition_variable g_plan_cv;
std::map<PlanKey, std::chrono::steady_clock::time_point> g_failed_at;   // when a null entry was made: retried after a minute
constexpr int FAILED_RETRY_SECONDS = 60;
std::atomic<uint64_t> g_generated_launches{0};
std::atomic<uint64_t> g_plan_evictions{0};
uint64_t g_plan_clock = 0;                               // g_plan_mu

size_t plans_max() {
    static const size_t v = [] {
        const char* e = getenv("H2_EVALH_PLANS_MAX");
        const long n = e ? atol(e) : 0;
        return (size_t)(n > 0 ? n : 128);
    }();
    return v;
}

// g_plan_mu held.  UNHOOKS least-recently-used plans nobody holds until the cache is back at its bound and hands them to the
// caller, who unloads them with `unload_plans` AFTER releasing the lock: a plan's device is drained first (its launches are
// asynchronous and the code object has to outlive them), and that wait must not stall every other lookup.
void evict_plans(const PlanKey& keep, std::vector<std::unique_ptr<EvalhPlan>>& gone) {
    while (g_plans.size() > plans_max()) {
        auto victim = g_plans.end();
        for (auto it = g_plans.begin(); it != g_plans.end(); ++it) {
            if (!(it->first < keep) && !(keep < it->first)) continue;
            if (it->second && it->second->users) continue;
            const uint64_t t = it->second ? it->second->last_use : 0;   // (failed generations first: they hold nothing)
            if (victim == g_plans.end() || t < (victim->second ? victim->second->last_use : 0)) victim = it;
        }
        if (victim == g_plans.end()) return;   // everything else is in use: over the bound until a holder lets go
        if (victim->second) gone.push_back(std::move(victim->second));
        g_failed_at.erase(victim->first);
        g_plans.erase(victim);
        g_plan_evictions.fetch_add(1);
    }
}
// no lock held: nobody can reach these plans any more
void unload_plans(std::vector<std::unique_ptr<EvalhPlan>>& gone) {
    for (auto& pl : gone) {
        int current = 0;
        const bool have = hipGetDevice(&current) == hipSuccess;
        if (hipSetDevice(pl->device) == hipSuccess) {
            (void)hipDeviceSynchronize();
            for (hipModule_t m : pl->modules) (void)hipModuleUnload(m);
        }
        if (have) (void)hipSetDevice(current);
    }
    gone.clear();
}

// in-memory identity of a program: two multiply-xorshift lanes over the same bytes program_hash covers (the SHA-256 is
// for the file name on disk, where a collision would load foreign code; here it would at worst pick the wrong cached plan
// of THIS process's own programs, at 2^-128)
struct FastHash {
    uint64_t a = 0x9e3779b97f4a7c15ull, b = 0xc2b2ae3d27d4eb4full;
    void word(uint64_t w) {
        a = (a ^ w) * 0xff51afd7ed558ccdull;
        a ^= a >> 32;
        b = (b + w) * 0x9fb21c651e98df25ull;
        b ^= b >> 29;
    }
    void bytes(const void* p, size_t n) {
        const uint8_t* q = (const uint8_t*)p;
        word(n);
        for (; n >= 8; n -= 8, q += 8) {
            uint64_t w;
            memcpy(&w, q, 8);
            word(w);
        }
        uint64_t w = 0;
        memcpy(&w, q, n);
        word(w);
    }
};

PlanKey plan_key(const h2_evalh_desc* d, const evgen::Options& opt, int device) {
    FastHash h;
    const uint32_t o[] = {opt.group, opt.max_ahead, opt.gap, opt.inline_muls, opt.stage_products, opt.max_cols, opt.max_regs,
                          (uint32_t)opt.factor, opt.waves, opt.live_budget, opt.lds_args, (uint32_t)opt.mul2, opt.min_group, d->blinding_factors, d->chunk_len, d->n_fixed, d->n_advice,
                          d->n_instance, d->n_perm_sets, d->n_perm_columns, d->n_lookups, d->n_shuffles};
    h.bytes(o, sizeof o);
    h.bytes(d->constants, (size_t)d->n_constants * 32);
    h.bytes(d->rotations, (size_t)d->n_rotations * 4);
    h.bytes(d->calculations, (size_t)d->n_calculations * sizeof(h2_calculation));
    h.bytes(d->value_parts, (size_t)d->n_value_parts * sizeof(h2_value_source));
    size_t n_lookup_calcs = 0;
    for (uint32_t t = 0; t < d->n_lookups; t++) n_lookup_calcs += 1 + 2 * (size_t)d->lookup_sets[t];
    h.bytes(d->lookup_sets, (size_t)d->n_lookups * 4);
    h.bytes(d->lookup_calcs, n_lookup_calcs * sizeof(h2_calculation));
    h.bytes(d->shuffle_calcs, 2 * (size_t)d->n_shuffles * sizeof(h2_calculation));
    if (d->n_perm_sets) {
        h.bytes(d->perm_col_type, (size_t)d->n_perm_columns * 4);
        h.bytes(d->perm_col_index, (size_t)d->n_perm_columns * 4);
    }
    return PlanKey{{h.a, h.b}, device};
}

bool generation_enabled() {
    const char* v = getenv("H2_EVALH_JIT");
    return !(v && v[0] == '0');
}
}  // namespace

void evalh_plan_release(const EvalhPlan* plan) {
    std::lock_guard<std::mutex> g(g_plan_mu);
    if (plan->users) plan->users--;
}

uint64_t evalh_plan_evictions() { return g_plan_evictions.load(); }

EvalhPlanRef evalh_plan_get(const h2_evalh_desc* d, int* cached) {
    if (cached) *cached = 0;
    if (!generation_enabled()) return EvalhPlanRef();
    int device = 0;
    H2_HIP(hipGetDevice(&device));
    const evgen::Options opt = evgen::Options::from_env();
    const PlanKey key = plan_key(d, opt, device);
    // one builder per PROGRAM: two callers with the same new program would otherwise both compile it.  The build itself runs
    // without the lock.
    std::unique_lock<std::mutex> g(g_plan_mu);
    for (;;) {
        auto it = g_plans.find(key);
        if (it != g_plans.end()) {
            if (!it->second) {
                // a failed generation is not for ever: a transient failure (a full disk, a compiler that was being replaced)
                // is tried again after a minute
                auto f = g_failed_at.find(key);
                if (f != g_failed_at.end() && std::chrono::steady_clock::now() - f->second > std::chrono::seconds(FAILED_RETRY_SECONDS)) {
                    g_failed_at.erase(f);
                    g_plans.erase(it);
                    continue;
                }
            }
            if (cached) *cached = 1;
            if (const EvalhPlan* hit = it->second.get()) {
                hit->last_use = ++g_plan_clock;
                hit->users++;
            }
            return EvalhPlanRef(it->second.get());
        }
        if (!g_building.count(key)) break;
        g_plan_cv.wait(g);
    }
    g_building.insert(key);
    g.unlock();
    struct Done {   // whatever happens below, the waiters of this program are released
        const PlanKey& key;
        std::unique_lock<std::mutex>& g;
        ~Done() {
            if (!g.owns_lock()) g.lock();
            g_building.erase(key);
            g_plan_cv.notify_all();
        }
    } done{key, g};
    std::unique_ptr<EvalhPlan> plan;
    try {
        plan.reset(new EvalhPlan);
        plan->gen = evgen::compile(d, opt);
        plan->device = device;
        plan->from_cache = plan->gen.from_disk ? 2 : 0;
        for (evgen::Stage& st : plan->gen.stages) {
            hipModule_t mod = nullptr;
            H2_HIP(hipModuleLoadData(&mod, st.code.data()));
            plan->modules.push_back(mod);
            hipFunction_t fn = nullptr;
            H2_HIP(hipModuleGetFunction(&fn, mod, evgen::KERNEL_NAME));
            plan->functions.push_back(fn);
            plan->uses_omega = plan->uses_omega || st.uses_omega;
            // (st.code stays for the life of the plan: hipModuleLoadData does not copy the image -- freeing it here ended in
            //  memory faults at the first launch)
            std::string().swap(st.source);
        }
        if (cached) *cached = plan->from_cache;
    } catch (const std::exception& e) {
        fprintf(stderr, "libhalo2_hip: evaluate_h runs on the interpreter kernels for this program: %s\n", e.what());
        plan.reset();
    } catch (const HipError& e) {
        fprintf(stderr, "libhalo2_hip: evaluate_h runs on the interpreter kernels for this program: HIP error %d loading the generated code (%s)\n",
                (int)e.code, e.expr);
        plan.reset();
    }
    g.lock();
    const EvalhPlan* out = plan.get();
    if (out) {
        out->last_use = ++g_plan_clock;
        out->users = 1;
    } else {
        g_failed_at[key] = std::chrono::steady_clock::now();
    }
    g_plans[key] = std::move(plan);
    std::vector<std::unique_ptr<EvalhPlan>> gone;
    evict_plans(key, gone);
    if (!gone.empty()) {
        g.unlock();
        unload_plans(gone);
    }
    return EvalhPlanRef(out);
}

void evalh_plan_info(const EvalhPlan* plan, h2_evalh_info* info) {
    if (!info) return;
    memset(info, 0, sizeof *info);
    if (!plan) return;
    evalh_gen_info(plan->gen, info);
}

void evalh_gen_info(const evgen::Generated& g, h2_evalh_info* info) {
    memset(info, 0, sizeof *info);
    info->stages = (uint32_t)g.stages.size();
    info->terms = g.terms;
    info->products_per_row = g.products_per_row;
    info->fused_pairs_per_row = g.fused_pairs_per_row;
    info->reference_products_per_row = g.reference_products_per_row;
    info->vectors_read = g.vectors_read;
    for (const evgen::Stage& st : g.stages) {
        info->max_registers = std::max(info->max_registers, st.vgprs + st.agprs);
        info->scratch_bytes = std::max(info->scratch_bytes, st.scratch);
    }
    info->from_cache = g.from_disk ? 2 : 0;
}

uint64_t evalh_generated_launches() { return g_generated_launches.load(); }

// The argument block of one generated stage (evalh_gen.hpp: fixed part, uniform scalars, column pointers) for descriptor `d`:
// what the kernel receives by value.  Host arithmetic only -- also behind h2_evalh_stage_args, through which the CPU tests run
// the generated source (compiled for the host) against the oracle.  Returns the block's size; throws on a program / descriptor
// mismatch.
size_t evalh_fill_stage_args(const evgen::Stage& st, const h2_evalh_desc* d, Fr* values, const Fr* tw_lo, const Fr* tw_hi,
                             size_t row_begin, size_t row_end, unsigned char* buf, size_t cap) {
    const Fr y = fr_from_u64x4(d->y), beta = fr_from_u64x4(d->beta), gamma = fr_from_u64x4(d->gamma), theta = fr_from_u64x4(d->theta);
    const Fr delta = fr_from_u64x4(d->delta), delta_start = fp_mul(beta, fr_from_u64x4(d->zeta));  // evaluation.rs:1012
    size_t n_lookup_z = 0;
    for (uint32_t t = 0; t < d->n_lookups; t++) n_lookup_z += d->lookup_sets[t];
    const size_t ns = std::max<size_t>(st.scalars.size(), 1), nc = std::max<size_t>(st.cols.size(), 1);
    const size_t bytes = evgen::ARGS_FIXED_BYTES + 32 * ns + 8 * nc;
    if (bytes > cap) throw std::runtime_error("h2_evaluate_h: generated stage with oversized arguments");
    memset(buf, 0, bytes);
    struct Fixed {
        Fr* values;
        const Fr* tw_lo;
        const Fr* tw_hi;
        unsigned long long row_begin, row_end;
        unsigned int extended_k, rot_scale;
    } fx{values, tw_lo, tw_hi, (unsigned long long)row_begin, (unsigned long long)row_end, d->extended_k, 1u << (d->extended_k - d->k)};
    static_assert(sizeof(Fixed) == evgen::ARGS_FIXED_BYTES, "argument layout");
    memcpy(buf, &fx, sizeof fx);
    Fr* sc = (Fr*)(buf + evgen::ARGS_FIXED_BYTES);
    std::vector<Fr> delta_pow;  // DELTA^j, grown on demand
    for (size_t i = 0; i < st.scalars.size(); i++) {
        const evgen::ScalarRef& r = st.scalars[i];
        switch (r.kind) {
            case evgen::SC_ONE: sc[i] = fp_one<FrParams>(); break;
            case evgen::SC_CONST: sc[i] = fr_from_u64x4(d->constants + 4 * (size_t)r.arg); break;
            case evgen::SC_Y_POW: sc[i] = fp_pow_u32(y, r.arg); break;
            case evgen::SC_BETA_POW: sc[i] = fp_pow_u32(beta, r.arg); break;
            case evgen::SC_GAMMA_POW: sc[i] = fp_pow_u32(gamma, r.arg); break;
            case evgen::SC_THETA: sc[i] = theta; break;
            default: {  // SC_DELTA_TERM: beta ZETA DELTA^arg
                if (delta_pow.empty()) delta_pow.push_back(fp_one<FrParams>());
                while (delta_pow.size() <= r.arg) delta_pow.push_back(fp_mul(delta_pow.back(), delta));
                sc[i] = fp_mul(delta_start, delta_pow[r.arg]);
            }
        }
    }
    const void** cols = (const void**)(buf + evgen::ARGS_FIXED_BYTES + 32 * ns);
    for (size_t i = 0; i < st.cols.size(); i++) {
        const evgen::ColRef& c = st.cols[i];
        const void* p = nullptr;
        switch (c.table) {
            case evgen::T_FIXED: p = c.index < d->n_fixed ? d->fixed[c.index] : nullptr; break;
            case evgen::T_ADVICE: p = c.index < d->n_advice ? d->advice[c.index] : nullptr; break;
            case evgen::T_INSTANCE: p = c.index < d->n_instance ? d->instance[c.index] : nullptr; break;
            case evgen::T_PERM_Z: p = c.index < d->n_perm_sets ? d->perm_z[c.index] : nullptr; break;
            case evgen::T_PERM_SIGMA: p = c.index < d->n_perm_columns ? d->perm_sigma[c.index] : nullptr; break;
            case evgen::T_LOOKUP_Z: p = c.index < n_lookup_z ? d->lookup_z[c.index] : nullptr; break;
            case evgen::T_LOOKUP_M: p = c.index < d->n_lookups ? d->lookup_m[c.index] : nullptr; break;
            case evgen::T_SHUFFLE_Z: p = c.index < d->n_shuffles ? d->shuffle_z[c.index] : nullptr; break;
            case evgen::T_L0: p = d->l0; break;
            case evgen::T_L_LAST: p = d->l_last; break;
            default: p = d->l_active_row;
        }
        if (!p) throw std::runtime_error("h2_evaluate_h: a column the program reads has a null pointer in the descriptor");
        cols[i] = p;
    }
    return bytes;
}

// Fills each stage's argument block and launches it.
static void evalh_plan_launch(const EvalhPlan* plan, const h2_evalh_desc* d, Fr* d_values, const Fr* tw_lo, const Fr* tw_hi,
                              size_t row_begin, size_t row_end, hipStream_t stream) {
    const size_t rows = row_end - row_begin;
    const unsigned blocks = (unsigned)std::min<size_t>((rows + 255) / 256, 0x7fffffffu);
    for (size_t s = 0; s < plan->gen.stages.size(); s++) {
        const evgen::Stage& st = plan->gen.stages[s];
        alignas(16) unsigned char buf[4096];
        const size_t bytes = evalh_fill_stage_args(st, d, d_values, tw_lo, tw_hi, row_begin, row_end, buf, sizeof buf);
        static const bool debug = getenv("H2_JIT_DEBUG") != nullptr;
        if (debug)
            fprintf(stderr, "h2_evalh_gen stage %zu: %u blocks, rows [%zu, %zu), %zu scalars, %zu columns, %zu argument bytes, values %p\n", s,
                    blocks, row_begin, row_end, st.scalars.size(), st.cols.size(), bytes, (void*)d_values);
        if (getenv("H2_JIT_LAUNCH_EXTRA")) {
            size_t arg_bytes = (bytes + 15) & ~(size_t)15;
            void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, buf, HIP_LAUNCH_PARAM_BUFFER_SIZE, &arg_bytes, HIP_LAUNCH_PARAM_END};
            H2_HIP(hipModuleLaunchKernel(plan->functions[s], blocks, 1, 1, 256, 1, 1, 0, stream, nullptr, config));
        } else {
            void* kargs[] = {buf};  // the kernel's one parameter: the argument block by value (copied at the call)
            H2_HIP(hipModuleLaunchKernel(plan->functions[s], blocks, 1, 1, 256, 1, 1, 0, stream, kargs, nullptr));
        }
        if (debug) H2_HIP(hipStreamSynchronize(stream));
        g_generated_launches.fetch_add(1);
    }
}

// ---------------------------------------------------------------- host driver
namespace {
struct Arena {  // bump allocator over one host staging block mirrored to one device block
    std::vector<char> host;
    char* dev = nullptr;
    size_t off = 0;
    template <class T>
    const T* put(const T* src, size_t count) {
        off = (off + 15) & ~(size_t)15;
        size_t bytes = count * sizeof(T);
        if (host.size() < off + bytes + 16) host.resize(off + bytes + 16);
        if (bytes) memcpy(host.data() + off, src, bytes);
        const T* d = reinterpret_cast<const T*>(dev + off);
        off += bytes;
        return d;
    }
};
}  // namespace

int evalh_device(DeviceCtx* ctx, const h2_evalh_desc* d, Fr* d_values, hipStream_t stream, bool have_lock) {
    if (!d || !d_values) {
        set_last_error("h2_evaluate_h: null argument");
        return H2_ERR_INVALID;
    }
    if (d->extended_k < d->k || d->extended_k > 28) {
        set_last_error("h2_evaluate_h: bad k / extended_k");
        return H2_ERR_INVALID;
    }
    if (d->n_perm_sets && d->chunk_len == 0) {
        set_last_error("h2_evaluate_h: chunk_len must be non-zero when permutation sets are present");
        return H2_ERR_INVALID;
    }
    if (d->reserved != nullptr) {
        set_last_error("h2_evaluate_h: `reserved` must be NULL (round 4's caller-supplied kernel slot: the library generates the "
                       "program's kernels itself now, see h2_evalh_prepare)");
        return H2_ERR_INVALID;
    }
    const size_t size = (size_t)1 << d->extended_k;
    const uint32_t rot_scale = 1u << (d->extended_k - d->k);
    if ((size_t)d->row_begin + d->row_count > size) {
        set_last_error("h2_evaluate_h: row range outside the domain");
        return H2_ERR_INVALID;
    }
    const size_t row_begin = d->row_count ? d->row_begin : 0, row_end = d->row_count ? (size_t)d->row_begin + d->row_count : size;
    const size_t rows = row_end - row_begin;

    // ---- the program as generated straight-line kernels (evalh_gen.cpp), unless switched off or unavailable
    if (!(d->flags & H2_EVALH_INTERPRET)) {
        if (EvalhPlanRef held = evalh_plan_get(d, nullptr)) {
            const EvalhPlan* plan = held.get();
            PlanRef pl;  // the power tables of extended_omega: pinned until the kernels that read them are launched
            if (plan->uses_omega) {
                if (have_lock) {
                    pl = ntt_get_plan(ctx, d->extended_k, d->extended_omega, stream);
                } else {
                    std::lock_guard<std::mutex> g(ctx->mu);
                    pl = ntt_get_plan(ctx, d->extended_k, d->extended_omega, stream);
                }
            }
            evalh_plan_launch(plan, d, d_values, plan->uses_omega ? pl->tw_lo : nullptr, plan->uses_omega ? pl->tw_hi : nullptr,
                              row_begin, row_end, stream);
            return H2_OK;
        }
    }

    // ---- the interpreter: stage the program and pointer tables (a few KB) into one device block
    size_t n_lookup_calcs = 0, n_lookup_z = 0;
    for (uint32_t t = 0; t < d->n_lookups; t++) {
        n_lookup_calcs += 1 + 2 * (size_t)d->lookup_sets[t];
        n_lookup_z += d->lookup_sets[t];
    }
    size_t need = 4096 + d->n_constants * 32 + d->n_rotations * 4 + (d->n_calculations + n_lookup_calcs + 2 * d->n_shuffles) * sizeof(h2_calculation) +
                  d->n_value_parts * sizeof(h2_value_source) + d->n_lookups * 4 +
                  ((size_t)d->n_fixed + d->n_advice + d->n_instance + d->n_perm_sets + 2 * (size_t)d->n_perm_columns + n_lookup_z +
                   d->n_lookups + d->n_shuffles) * 8 + 64 * 16;
    // ---- work space: interpreter intermediates + lookup / shuffle compressed expressions
    const unsigned blocks = 256 * 8, threads = 256;
    const size_t nthreads = (size_t)blocks * threads;
    size_t inter_bytes = (size_t)(d->n_calculations ? d->n_calculations : 1) * nthreads * sizeof(Fr);
    size_t lk_bytes = n_lookup_calcs ? n_lookup_calcs * size * sizeof(Fr) : 256;
    size_t sh_bytes = d->n_shuffles ? 2 * (size_t)d->n_shuffles * size * sizeof(Fr) : 256;
    auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t total = align(need) + align(inter_bytes) + align(lk_bytes) + align(sh_bytes);
    char* block = (char*)ctx->evalh_scratch.get(total);
    Arena ar;
    ar.dev = block;
    Fr* d_inter = (Fr*)(block + align(need));
    Fr* d_lk = (Fr*)((char*)d_inter + align(inter_bytes));
    Fr* d_sh = (Fr*)((char*)d_lk + align(lk_bytes));

    EvalhProgram p{};
    p.constants = (const Fr*)ar.put(d->constants, (size_t)d->n_constants * 4);
    p.rotations = ar.put(d->rotations, d->n_rotations);
    p.calcs = ar.put(d->calculations, d->n_calculations);
    p.value_parts = ar.put(d->value_parts, d->n_value_parts);
    p.lookup_calcs = ar.put(d->lookup_calcs, n_lookup_calcs);
    p.lookup_sets = ar.put(d->lookup_sets, d->n_lookups);
    p.shuffle_calcs = ar.put(d->shuffle_calcs, 2 * (size_t)d->n_shuffles);
    p.fixed = (const Fr* const*)ar.put(d->fixed, d->n_fixed);
    p.advice = (const Fr* const*)ar.put(d->advice, d->n_advice);
    p.instance = (const Fr* const*)ar.put(d->instance, d->n_instance);
    const Fr* const* d_perm_z = (const Fr* const*)ar.put(d->perm_z, d->n_perm_sets);
    const Fr* const* d_perm_sigma = (const Fr* const*)ar.put(d->perm_sigma, d->n_perm_columns);
    std::vector<const uint64_t*> cols(d->n_perm_columns);
    for (uint32_t j = 0; j < d->n_perm_columns; j++) {  // evaluation.rs:1060-1064
        uint32_t ty = d->perm_col_type[j], ix = d->perm_col_index[j];
        const uint64_t* const* tab = ty == H2_ANY_ADVICE ? d->advice : (ty == H2_ANY_FIXED ? d->fixed : d->instance);
        uint32_t lim = ty == H2_ANY_ADVICE ? d->n_advice : (ty == H2_ANY_FIXED ? d->n_fixed : d->n_instance);
        if (ix >= lim) {
            set_last_error("h2_evaluate_h: permutation column index out of range");
            return H2_ERR_INVALID;
        }
        cols[j] = tab[ix];
    }
    const Fr* const* d_perm_cols = (const Fr* const*)ar.put(cols.data(), cols.size());
    const Fr* const* d_lookup_z = (const Fr* const*)ar.put(d->lookup_z, n_lookup_z);
    if (ar.off > need) {
        set_last_error("h2_evaluate_h: internal staging overflow");
        return H2_ERR_INVALID;
    }
    H2_HIP(hipMemcpyAsync(block, ar.host.data(), ar.off, hipMemcpyHostToDevice, stream));
    H2_HIP(hipStreamSynchronize(stream));  // the staging vector dies with this frame

    p.n_calcs = d->n_calculations;
    p.n_value_parts = d->n_value_parts;
    p.n_lookups = d->n_lookups;
    p.n_shuffles = d->n_shuffles;
    p.extended_k = d->extended_k;
    p.rot_scale = rot_scale;
    p.row_begin = row_begin;
    p.row_end = row_end;
    p.y = fr_from_u64x4(d->y);
    p.beta = fr_from_u64x4(d->beta);
    p.gamma = fr_from_u64x4(d->gamma);
    p.theta = fr_from_u64x4(d->theta);

    const int32_t last_rotation = -((int32_t)d->blinding_factors + 1);
    PlanRef pl;  // the power tables of extended_omega: pinned until the kernels that read them are launched
    if (d->n_perm_sets) {
        if (have_lock) {
            pl = ntt_get_plan(ctx, d->extended_k, d->extended_omega, stream);
        } else {
            std::lock_guard<std::mutex> g(ctx->mu);
            pl = ntt_get_plan(ctx, d->extended_k, d->extended_omega, stream);
        }
    }
    {
        hipLaunchKernelGGL(k_evalh_expr, dim3(blocks), dim3(threads), 0, stream, p, d_inter, d_values, d_lk, d_sh);
    }

    unsigned eblocks = (unsigned)std::min<size_t>((rows + 255) / 256, 0x7fffffffu);
    if (d->n_perm_sets) {
        PermArgs a{};
        a.row_begin = row_begin;
        a.row_end = row_end;
        a.values = d_values;
        a.perm_z = d_perm_z;
        a.perm_cols = d_perm_cols;
        a.perm_sigma = d_perm_sigma;
        a.l0 = (const Fr*)d->l0;
        a.l_last = (const Fr*)d->l_last;
        a.l_active_row = (const Fr*)d->l_active_row;
        a.tw_lo = pl->tw_lo;
        a.tw_hi = pl->tw_hi;
        a.n_sets = d->n_perm_sets;
        a.n_columns = d->n_perm_columns;
        a.chunk_len = d->chunk_len;
        a.extended_k = d->extended_k;
        a.rot_scale = rot_scale;
        a.last_rotation = last_rotation;
        a.y = p.y;
        a.beta = p.beta;
        a.gamma = p.gamma;
        a.delta = fr_from_u64x4(d->delta);
        a.delta_start = fp_mul(p.beta, fr_from_u64x4(d->zeta));  // evaluation.rs:1012
        hipLaunchKernelGGL(k_evalh_perm, dim3(eblocks), dim3(256), 0, stream, a);
    }
    size_t zoff = 0, slot = 0;
    for (uint32_t lk = 0; lk < d->n_lookups; lk++) {
        LookupArgs a{};
        a.row_begin = row_begin;
        a.row_end = row_end;
        a.values = d_values;
        a.zs = d_lookup_z + zoff;
        a.m = (const Fr*)d->lookup_m[lk];
        a.table = d_lk + slot * size;
        a.l0 = (const Fr*)d->l0;
        a.l_last = (const Fr*)d->l_last;
        a.l_active_row = (const Fr*)d->l_active_row;
        a.sets_len = d->lookup_sets[lk];
        a.extended_k = d->extended_k;
        a.rot_scale = rot_scale;
        a.last_rotation = last_rotation;
        a.y = p.y;
        hipLaunchKernelGGL(k_evalh_lookup, dim3(eblocks), dim3(256), 0, stream, a);
        zoff += d->lookup_sets[lk];
        slot += 1 + 2 * (size_t)d->lookup_sets[lk];
    }
    for (uint32_t sh = 0; sh < d->n_shuffles; sh++) {
        ShuffleArgs a{};
        a.row_begin = row_begin;
        a.row_end = row_end;
        a.values = d_values;
        a.z = (const Fr*)d->shuffle_z[sh];
        a.input = d_sh + (size_t)(2 * sh) * size;
        a.shuffle = d_sh + (size_t)(2 * sh + 1) * size;
        a.l0 = (const Fr*)d->l0;
        a.l_last = (const Fr*)d->l_last;
        a.l_active_row = (const Fr*)d->l_active_row;
        a.extended_k = d->extended_k;
        a.rot_scale = rot_scale;
        a.y = p.y;
        hipLaunchKernelGGL(k_evalh_shuffle, dim3(eblocks), dim3(256), 0, stream, a);
    }
    H2_HIP(hipGetLastError());
    return H2_OK;
}

// Host-buffer variant: upload every coset the descriptor references (each distinct pointer once),
// run, read the values back.
int evalh_host(DeviceCtx* ctx, const h2_evalh_desc* d, uint64_t* values) {
    if (!d || !values) {
        set_last_error("h2_evaluate_h: null argument");
        return H2_ERR_INVALID;
    }
    if (d->row_count) {
        set_last_error("h2_evaluate_h: a row range is for the device entry point (h2_dev_evaluate_h)");
        return H2_ERR_INVALID;
    }
    const size_t size = (size_t)1 << d->extended_k, bytes = size * sizeof(Fr);
    std::map<const uint64_t*, const uint64_t*> up;  // host -> device
    std::vector<void*> owned;
    auto cleanup = [&] {
        for (void* q : owned) (void)hipFree(q);
    };
    try {
        auto dev_of = [&](const uint64_t* h) -> const uint64_t* {
            if (!h) return nullptr;
            auto it = up.find(h);
            if (it != up.end()) return it->second;
            void* q = nullptr;
            H2_HIP(hipMalloc(&q, bytes));
            owned.push_back(q);
            host_upload(q, h, bytes, ctx->stream);
            up[h] = (const uint64_t*)q;
            return (const uint64_t*)q;
        };
        h2_evalh_desc dd = *d;
        size_t n_lookup_z = 0;
        for (uint32_t t = 0; t < d->n_lookups; t++) n_lookup_z += d->lookup_sets[t];
        auto map_table = [&](const uint64_t* const* tab, size_t n, std::vector<const uint64_t*>& out) {
            out.resize(n);
            for (size_t i = 0; i < n; i++) out[i] = dev_of(tab[i]);
            return out.data();
        };
        std::vector<const uint64_t*> t_fixed, t_adv, t_inst, t_pz, t_ps, t_lz, t_lm, t_sz;
        dd.fixed = map_table(d->fixed, d->n_fixed, t_fixed);
        dd.advice = map_table(d->advice, d->n_advice, t_adv);
        dd.instance = map_table(d->instance, d->n_instance, t_inst);
        dd.perm_z = map_table(d->perm_z, d->n_perm_sets, t_pz);
        dd.perm_sigma = map_table(d->perm_sigma, d->n_perm_columns, t_ps);
        dd.lookup_z = map_table(d->lookup_z, n_lookup_z, t_lz);
        dd.lookup_m = map_table(d->lookup_m, d->n_lookups, t_lm);
        dd.shuffle_z = map_table(d->shuffle_z, d->n_shuffles, t_sz);
        dd.l0 = dev_of(d->l0);
        dd.l_last = dev_of(d->l_last);
        dd.l_active_row = dev_of(d->l_active_row);
        void* d_values = nullptr;
        H2_HIP(hipMalloc(&d_values, bytes));
        owned.push_back(d_values);
        int rc = evalh_device(ctx, &dd, (Fr*)d_values, ctx->stream, true);
        if (rc == H2_OK) {
            host_download(values, d_values, bytes, ctx->stream);
            H2_HIP(hipStreamSynchronize(ctx->stream));
        }
        cleanup();
        return rc;
    } catch (...) {
        cleanup();
        throw;
    }
}

// ---------------------------------------------------------------- coefficient forms in, extended values out
// out[i] = in[c * i + j]  /  out[c * i + j] = in[i]: one coset of the n-th roots of unity inside the extended domain
__global__ void __launch_bounds__(256) k_coset_gather(const Fr* in, Fr* out, size_t n, uint32_t log_c, uint32_t j) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) fp_store(out + i, fp_load(in + ((i << log_c) | j)));
}
__global__ void __launch_bounds__(256) k_coset_scatter(const Fr* in, Fr* out, size_t n, uint32_t log_c, uint32_t j) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) fp_store(out + ((i << log_c) | j), fp_load(in + i));
}

// The shape of the reference's cuda `Evaluator::evaluate_h` (plonk/evaluation.rs:1229-1985): every column arrives as a
// COEFFICIENT vector of 2^k elements in host memory (the `cuda` proving key keeps no extended cosets, plonk.rs:226-240),
// `l_active_row` as extended values (2^extended_k, host), and the numerator comes back as 2^extended_k host values.
// The reference re-derives extended cosets through a 5-entry cache while it walks its expression trees; here the
// extended domain is visited one coset g_j H of the n-th roots of unity at a time (g_j = zeta * extended_omega^j,
// extended index c i + j = point i of coset j, c = 2^(extended_k - k)): every distinct column is uploaded ONCE, taken to
// coset j by a[t] *= g_j^t and an n-point NTT, and the fused evaluator runs with extended_k := k, zeta := g_j,
// extended_omega := omega -- on one coset a rotation is an index shift and nothing else changes.  Device memory is
// (distinct columns) x 2 x 2^k x 32 B + two extended vectors: bounded by the circuit's width, not by width x 2^extended_k.
//
// One worker = one leased device running the cosets first, first + step, ...  With `whole` (the only worker) the values
// are scattered to their stride-c positions on the device and come back in one transfer; otherwise each coset's n values
// come back through the slot's pinned buffer and this thread writes them to values[c i + j] (the workers of the other
// devices do the same for their cosets at the same time: disjoint 32-byte cells of the caller's vector).
static int evalh_coeffs_worker(DeviceCtx* ctx, const h2_evalh_desc* d, uint64_t* values, uint32_t first, uint32_t step, bool whole,
                               const EvalhFinish* finish = nullptr) {
    const uint32_t log_c = d->extended_k - d->k, c = 1u << log_c;
    const size_t n = (size_t)1 << d->k, size = (size_t)1 << d->extended_k;
    const size_t nbytes = n * sizeof(Fr), ebytes = size * sizeof(Fr);
    hipStream_t stream = ctx->stream;
    std::vector<void*> owned;
    auto cleanup = [&] {
        for (void* q : owned) (void)hipFree(q);
    };
    // ONE device block for the call, handed out piece by piece (a call used to make ~2 x columns + 5 hipMalloc / hipFree pairs,
    // every hipFree a device synchronisation); sized from the distinct vectors of the descriptor, individual allocations if the
    // block cannot be had or turns out short.  The block belongs to the slot (ctx->coeff_arena, grow-only, given back by
    // h2_release_plans): a hipMalloc of 6.4 GiB per call (k = 22, 16 columns) took 0.3 ms when the runtime still had the range
    // of the previous call and 160 ms when it did not -- one proof in three of the literal drop-in's measured sequence.
    char* arena = nullptr;
    size_t arena_bytes = 0, arena_used = 0;
    auto dmalloc = [&](size_t bytes) -> void* {
        const size_t take = (bytes + 255) & ~(size_t)255;
        if (arena && arena_used + take <= arena_bytes) {
            void* q = arena + arena_used;
            arena_used += take;
            return q;
        }
        void* q = nullptr;
        H2_HIP(hipMalloc(&q, bytes));
        owned.push_back(q);
        return q;
    };
    try {
        {
            std::set<const uint64_t*> distinct;
            size_t n_lz = 0;
            for (uint32_t t = 0; t < d->n_lookups; t++) n_lz += d->lookup_sets[t];
            auto count = [&](const uint64_t* const* tab, size_t cnt) {
                for (size_t i = 0; i < cnt; i++)
                    if (tab[i]) distinct.insert(tab[i]);
            };
            count(d->fixed, d->n_fixed); count(d->advice, d->n_advice); count(d->instance, d->n_instance);
            count(d->perm_z, d->n_perm_sets); count(d->perm_sigma, d->n_perm_columns); count(d->lookup_z, n_lz);
            count(d->lookup_m, d->n_lookups); count(d->shuffle_z, d->n_shuffles);
            if (d->l0) distinct.insert(d->l0);
            if (d->l_last) distinct.insert(d->l_last);
            const size_t width = std::max<size_t>(1, std::min<size_t>(16, ((size_t)1 << 30) / nbytes));
            arena_bytes = (2 * distinct.size() + 2 + width) * (nbytes + 256) + (whole ? 2 * (ebytes + 256) : 0);
            try {
                arena = (char*)ctx->coeff_arena.get(arena_bytes);
            } catch (const HipError&) {
                (void)hipGetLastError();
                arena = nullptr;
                arena_bytes = 0;
            }
        }
        // every distinct coefficient vector once: (device coefficients, device values on the current coset)
        std::map<const uint64_t*, std::pair<Fr*, Fr*>> cols;
        auto add = [&](const uint64_t* h) {
            if (!h || cols.count(h)) return;
            // a vector inside a range registered with h2_poly_register (the proving key's fixed / sigma / l_0 / l_last forms,
            // plonk.rs:226-240) already has its device copy: only read here, never written
            Fr* dc = const_cast<Fr*>(poly_resident(ctx, h, n));
            if (!dc) {
                dc = (Fr*)dmalloc(nbytes);
                host_upload(dc, h, nbytes, stream);
            }
            Fr* dv = (Fr*)dmalloc(nbytes);
            cols[h] = {dc, dv};
        };
        size_t n_lookup_z = 0;
        for (uint32_t t = 0; t < d->n_lookups; t++) n_lookup_z += d->lookup_sets[t];
        auto add_table = [&](const uint64_t* const* tab, size_t cnt) {
            for (size_t i = 0; i < cnt; i++) add(tab[i]);
        };
        add_table(d->fixed, d->n_fixed);
        add_table(d->advice, d->n_advice);
        add_table(d->instance, d->n_instance);
        add_table(d->perm_z, d->n_perm_sets);
        add_table(d->perm_sigma, d->n_perm_columns);
        add_table(d->lookup_z, n_lookup_z);
        add_table(d->lookup_m, d->n_lookups);
        add_table(d->shuffle_z, d->n_shuffles);
        add(d->l0);
        add(d->l_last);
        Fr* d_active = nullptr;
        Fr* d_active_j = nullptr;
        Fr* d_values = nullptr;
        Fr* stage = nullptr;  // pinned, n elements: one coset of l_active_row on its way in, one coset of values on its way out
        if (d->l_active_row) d_active_j = (Fr*)dmalloc(nbytes);
        if (whole) {
            if (d->l_active_row) {
                // (the proving key's l_active_row, extended values: registered with the key's other vectors, plonk.rs:224)
                d_active = const_cast<Fr*>(poly_resident(ctx, d->l_active_row, size));
                if (!d_active) {
                    d_active = (Fr*)dmalloc(ebytes);
                    host_upload(d_active, d->l_active_row, ebytes, stream);
                }
            }
            d_values = (Fr*)dmalloc(ebytes);
        } else {
            stage = (Fr*)ctx->pinned.get(nbytes);
        }
        Fr* d_values_j = (Fr*)dmalloc(nbytes);
        // scratch of a batch of transforms: up to 16 vectors per launch, fewer when that would pass 1 GiB
        const size_t batch_width = std::max<size_t>(1, std::min<size_t>(16, ((size_t)1 << 30) / nbytes));
        Fr* d_tmp = (Fr*)dmalloc(batch_width * nbytes);
        // omega = extended_omega^c generates the n-th roots of unity
        const Fr w_ext = fr_from_u64x4(d->extended_omega), zeta = fr_from_u64x4(d->zeta);
        Fr omega = w_ext;
        for (uint32_t t = 0; t < log_c; t++) omega = fp_sqr(omega);
        uint64_t omega_u[4], g_u[4];
        for (int i = 0; i < 4; i++) omega_u[i] = (uint64_t)omega.l[2 * i] | ((uint64_t)omega.l[2 * i + 1] << 32);
        PlanRef pl = ntt_get_plan(ctx, d->k, omega_u, stream);  // the caller holds ctx->mu (DeviceLease)
        const unsigned nblocks = (unsigned)((n + 255) / 256);
        const Fr w_step = fp_pow_u32(w_ext, step);
        Fr g = fp_mul(zeta, fp_pow_u32(w_ext, first));  // g_first
        const Fr* active_host = (const Fr*)d->l_active_row;
        for (uint32_t j = first; j < c; j += step) {
            for (int i = 0; i < 4; i++) g_u[i] = (uint64_t)g.l[2 * i] | ((uint64_t)g.l[2 * i + 1] << 32);
            {
                // every column to coset j in batched, fused coset transforms (ntt.hip: g_j^t applied in the first pass's load,
                // sixteen vectors per launch): no copy, no separate scaling pass
                std::vector<const Fr*> srcs;
                std::vector<Fr*> dsts, tmps;
                for (auto& kv : cols) {
                    srcs.push_back(kv.second.first);
                    dsts.push_back(kv.second.second);
                    tmps.push_back(d_tmp + (tmps.size() % batch_width) * n);
                }
                ScaleTabRef tab = ntt_scale_table(pl.get(), g, nullptr, stream);
                for (size_t at = 0; at < srcs.size(); at += batch_width)   // (a chunk's scratch slots are distinct)
                    ntt_run_many(ctx, pl.get(), srcs.data() + at, dsts.data() + at, tmps.data() + at,
                                 std::min(batch_width, srcs.size() - at), (uint32_t)n, nullptr, nullptr, stream, tab.get(), 1u);
            }
            if (d_active) {
                hipLaunchKernelGGL(k_coset_gather, dim3(nblocks), dim3(256), 0, stream, d_active, d_active_j, n, log_c, j);
            } else if (active_host) {
                for (size_t i = 0; i < n; i++) stage[i] = active_host[(i << log_c) | j];
                H2_HIP(hipMemcpyAsync(d_active_j, stage, nbytes, hipMemcpyHostToDevice, stream));
            }
            h2_evalh_desc dd = *d;
            dd.extended_k = d->k;
            for (int i = 0; i < 4; i++) {
                dd.zeta[i] = g_u[i];
                dd.extended_omega[i] = omega_u[i];
            }
            auto on_coset = [&](const uint64_t* h) -> const uint64_t* { return h ? (const uint64_t*)cols[h].second : nullptr; };
            auto map_table = [&](const uint64_t* const* tab, size_t cnt, std::vector<const uint64_t*>& out) {
                out.resize(cnt);
                for (size_t i = 0; i < cnt; i++) out[i] = on_coset(tab[i]);
                return out.data();
            };
            std::vector<const uint64_t*> t_fixed, t_adv, t_inst, t_pz, t_ps, t_lz, t_lm, t_sz;
            dd.fixed = map_table(d->fixed, d->n_fixed, t_fixed);
            dd.advice = map_table(d->advice, d->n_advice, t_adv);
            dd.instance = map_table(d->instance, d->n_instance, t_inst);
            dd.perm_z = map_table(d->perm_z, d->n_perm_sets, t_pz);
            dd.perm_sigma = map_table(d->perm_sigma, d->n_perm_columns, t_ps);
            dd.lookup_z = map_table(d->lookup_z, n_lookup_z, t_lz);
            dd.lookup_m = map_table(d->lookup_m, d->n_lookups, t_lm);
            dd.shuffle_z = map_table(d->shuffle_z, d->n_shuffles, t_sz);
            dd.l0 = on_coset(d->l0);
            dd.l_last = on_coset(d->l_last);
            dd.l_active_row = (const uint64_t*)d_active_j;
            int rc = evalh_device(ctx, &dd, d_values_j, stream, true);
            if (rc != H2_OK) {
                cleanup();
                return rc;
            }
            if (whole) {
                hipLaunchKernelGGL(k_coset_scatter, dim3(nblocks), dim3(256), 0, stream, d_values_j, d_values, n, log_c, j);
            } else {
                H2_HIP(hipMemcpyAsync(stage, d_values_j, nbytes, hipMemcpyDeviceToHost, stream));
                H2_HIP(hipStreamSynchronize(stream));
                Fr* out = (Fr*)values;
                for (size_t i = 0; i < n; i++) out[(i << log_c) | j] = stage[i];
            }
            g = fp_mul(g, w_step);
        }
        H2_HIP(hipGetLastError());
        int frc = H2_OK;
        if (whole && finish)
            frc = (*finish)(ctx, d_values, stream);          // (what follows the evaluation, on the device: capi.hip)
        else if (whole)
            host_download(values, d_values, ebytes, stream);
        H2_HIP(hipStreamSynchronize(stream));
        cleanup();
        return frc;
    } catch (...) {
        (void)hipStreamSynchronize(stream);
        cleanup();
        throw;
    }
}

// The reference's cuda evaluate_h deals its gates / lookups / shuffles over the N_GPU devices of the process inside the one
// call (plonk/evaluation.rs:326-333,1262-1275,1513-1520,1830-1837: `group_expr_len`, one thread per GPU).  Here the unit is a
// coset of the extended domain -- the work per coset is the same and nothing is exchanged between cosets: P = min(pool
// size, cosets) threads lease a device each (acquire_gpu, arithmetic.rs:314-321), upload the coefficient vectors once per
// device and take the cosets p, p + P, ...; with one device (or one coset) the caller's thread does it all.
uint32_t evalh_host_workers(const h2_evalh_desc* d) {
    if (!d || d->extended_k < d->k || d->extended_k > 28) return 1;
    const uint32_t c = 1u << (d->extended_k - d->k);
    return (uint32_t)std::max(1, std::min<int>(device_count(), (int)c));
}

int evalh_host_coeffs(const h2_evalh_desc* d, uint64_t* values) { return evalh_host_coeffs(d, values, nullptr, nullptr); }

int evalh_host_coeffs(const h2_evalh_desc* d, uint64_t* values, const EvalhFinish* finish, bool* finished) {
    if (finished) *finished = false;
    if (!d || (!values && !(finish && evalh_host_workers(d) <= 1))) {
        set_last_error("h2_evaluate_h_coeff: null argument");
        return H2_ERR_INVALID;
    }
    if (d->extended_k < d->k || d->extended_k > 28) {
        set_last_error("h2_evaluate_h_coeff: bad k / extended_k");
        return H2_ERR_INVALID;
    }
    if (d->row_count) {
        set_last_error("h2_evaluate_h_coeff: a row range is for the device entry point (h2_dev_evaluate_h)");
        return H2_ERR_INVALID;
    }
    const uint32_t c = 1u << (d->extended_k - d->k);
    const uint32_t workers = (uint32_t)std::max(1, std::min<int>(device_count(), (int)c));
    if (workers <= 1) {
        DeviceLease lease;
        if (finished) *finished = finish != nullptr;
        return evalh_coeffs_worker(lease.ctx, d, values, 0, 1, true, finish);
    }
    std::vector<int> rcs(workers, H2_OK);
    std::vector<std::string> errs(workers);
    std::vector<std::thread> th;
    for (uint32_t p = 0; p < workers; p++) {
        th.emplace_back([&, p] {
            rcs[p] = guarded([&] {
                DeviceLease lease;
                return evalh_coeffs_worker(lease.ctx, d, values, p, workers, false);
            });
            if (rcs[p] != H2_OK) errs[p] = get_last_error();
        });
    }
    for (auto& t : th) t.join();
    for (uint32_t p = 0; p < workers; p++)
        if (rcs[p] != H2_OK) {
            set_last_error(errs[p]);
            return rcs[p];
        }
    return H2_OK;
}

}  // namespace h2
