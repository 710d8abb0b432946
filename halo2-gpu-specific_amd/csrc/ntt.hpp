// ntt.hpp -- NTT plan (cached twiddle tables per (log_n, omega)) and the pass driver.
#pragma once
#include <atomic>
#include <vector>

#include "common.hpp"

namespace h2 {

struct NttPlan {
    uint32_t log_n = 0;
    Fr w;                              // the root of unity the plan was built for (Montgomery form)
    std::vector<uint32_t> bits;        // B_p per pass
    Fr* tables = nullptr;              // one allocation: lo | hi | per-pass butterfly tables
    const Fr* tw_lo = nullptr;         // w^i,        i < min(n, 4096)
    const Fr* tw_hi = nullptr;         // w^(i<<12),  i < n >> 12
    std::vector<const Fr*> tw_bfly;    // per pass: (w^(n/R))^e, e < R/2
    std::vector<const Fr*> tw_direct;  // per pass: full inter-pass twiddle table or nullptr
    // per pass: 1 = its tw_bfly / tw_direct / last-pass tables hold (plain value, floor(value 2^256 / r)) PAIRS, the operands of
    // the constant-operand product fp_mul_const (the radix-4 lazy passes of transforms >= 2^18; H2_NTT_CONSTW=0: none)
    std::vector<uint8_t> cw;
    size_t table_bytes = 0;            // of `tables` and the per-pass direct tables
    std::mutex mu;                     // guards scaled_hi, last_direct
    std::map<std::string, Fr*> scaled_hi;  // divisor -> tw_hi * divisor (iNTT: 1/n folded into the last pass)
    // (generator, divisor) -> two-level table of g^i (coset transforms): 4096 + n/4096 entries.  At most SCALE_TABS_MAX per
    // plan: the public coset entry points take any generator, so the least recently used table nobody holds is dropped
    struct ScaleTab {
        Fr* ptr = nullptr;
        size_t bytes = 0;
        uint64_t last_use = 0;
        int users = 0;  // ScaleTabRef holders: between looking the table up and having launched the pass that reads it
    };
    static constexpr size_t SCALE_TABS_MAX = 32;
    std::map<std::string, ScaleTab> scale_tabs;
    uint64_t scale_clock = 0;
    // the last pass's complete inter-pass twiddle set, w^(rho * K) at [(K << B_last) | rho] (2^log_n entries, streamed in
    // the order the pass loads its elements), keyed by the divisor folded into it ("" = none).  These are the large
    // optional tables (32 B x n each): they count against the per-device budget (ntt_table_budget) and the least
    // recently used idle one is evicted when a new one would exceed it; a transform that finds none composes its
    // twiddles from the two-level tables (one more product per element, same values).
    struct LastTable {
        Fr* ptr = nullptr;
        size_t bytes = 0;
        uint64_t last_use = 0;
        int users = 0;  // transforms between looking the table up and having launched the pass that reads it
    };
    std::map<std::string, LastTable> last_direct;
    std::map<std::string, int> last_misses;  // lookups of a key that found no table (a key evicts others from its 2nd miss on)
    std::atomic<int> users{0};         // callers holding the plan (PlanRef): a plan in use is not released
    uint64_t last_use = 0;
};

// a plan handed out by ntt_get_plan stays alive until its PlanRef goes (h2_release_plans skips plans in use)
struct PlanRef {
    NttPlan* pl = nullptr;
    PlanRef() = default;
    explicit PlanRef(NttPlan* p) : pl(p) {}
    PlanRef(PlanRef&& o) noexcept : pl(o.pl) { o.pl = nullptr; }
    PlanRef& operator=(PlanRef&& o) noexcept {
        if (this != &o) {
            if (pl) pl->users.fetch_sub(1);
            pl = o.pl;
            o.pl = nullptr;
        }
        return *this;
    }
    PlanRef(const PlanRef&) = delete;
    PlanRef& operator=(const PlanRef&) = delete;
    ~PlanRef() {
        if (pl) pl->users.fetch_sub(1);
    }
    NttPlan* operator->() const { return pl; }
    NttPlan* get() const { return pl; }
};

void ntt_split(uint32_t log_n, std::vector<uint32_t>& bits);
// call with ctx->mu held (the plan map is per device); the returned reference pins the plan
PlanRef ntt_get_plan(DeviceCtx* ctx, uint32_t log_n, const uint64_t omega[4], hipStream_t stream);
// frees every plan of `ctx` that no caller holds, with all its tables (synchronises the device first); call with
// ctx->mu held.  Returns the bytes released.
size_t ntt_release_plans(DeviceCtx* ctx);
size_t ntt_detach_idle_plans(DeviceCtx* ctx, std::vector<NttPlan*>& gone);  // under ctx->mu
void ntt_free_plans(std::vector<NttPlan*>& gone);                           // no lock held: synchronises, frees
// bytes of device memory the plans of `ctx` hold (twiddle tables + last-pass tables); call with ctx->mu held
size_t ntt_plan_bytes(DeviceCtx* ctx);
// per-device budget of the optional last-pass tables: H2_NTT_TABLE_BUDGET (bytes; K / M / G suffixes) or
// h2_set_table_budget; default 1/32 of the device's memory (9 GiB on an MI355X: a k = 24 proof's four tables take 3)
size_t ntt_table_budget(DeviceCtx* ctx);
void ntt_set_table_budget(size_t bytes);
// src (in_len valid elements, zero-extended to 2^log_n) -> dst; tmp = 2^log_n scratch (needed when
// the plan has >= 2 passes).  pre3 / post3: nullable HOST pointers to 3 Fr each (passed by value
// in the kernel arguments): x[i] *= pre3[i % 3] (i % 3 != 0) on load, y[i] *= post3[i % 3] on store.
// scale_tab (ntt_scale_table) with scale_mode 1: x[i] *= g^i on the first pass's load; 2: y[i] *= g^i (* divisor) on the
// final store -- the transforms between coefficients and ONE coset g H of a larger domain, without a separate scaling pass.
void ntt_run(DeviceCtx* ctx, NttPlan* pl, const Fr* src, Fr* dst, Fr* tmp, uint32_t in_len, const Fr* pre3,
             const Fr* post3, hipStream_t stream, const Fr* scale_tab = nullptr, uint32_t scale_mode = 0);
// the returned reference pins the table until the caller has launched what reads it (an evicted table is released with
// hipFree, which waits for the work already launched)
struct ScaleTabRef {
    NttPlan* pl = nullptr;
    NttPlan::ScaleTab* tab = nullptr;
    ScaleTabRef() = default;
    ScaleTabRef(NttPlan* p, NttPlan::ScaleTab* t) : pl(p), tab(t) {}
    ScaleTabRef(ScaleTabRef&& o) noexcept : pl(o.pl), tab(o.tab) { o.pl = nullptr; o.tab = nullptr; }
    ScaleTabRef& operator=(ScaleTabRef&& o) noexcept {
        if (this != &o) {
            drop();
            pl = o.pl;
            tab = o.tab;
            o.pl = nullptr;
            o.tab = nullptr;
        }
        return *this;
    }
    ScaleTabRef(const ScaleTabRef&) = delete;
    ScaleTabRef& operator=(const ScaleTabRef&) = delete;
    ~ScaleTabRef() { drop(); }
    const Fr* get() const { return tab ? tab->ptr : nullptr; }
    void drop() {
        if (tab) {
            std::lock_guard<std::mutex> g(pl->mu);
            tab->users--;
        }
        tab = nullptr;
    }
};
ScaleTabRef ntt_scale_table(NttPlan* pl, const Fr& g, const Fr* d, hipStream_t stream);
// `count` transforms of one plan with the same scales, several vectors per launch; tmps[i] = scratch of vector i
void ntt_run_many(DeviceCtx* ctx, NttPlan* pl, const Fr* const* srcs, Fr* const* dsts, Fr* const* tmps, size_t count,
                  uint32_t in_len, const Fr* pre3, const Fr* post3, hipStream_t stream, const Fr* scale_tab = nullptr,
                  uint32_t scale_mode = 0);
Fr fr_from_u64x4(const uint64_t v[4]);

}  // namespace h2
