// evalh.hpp -- evaluate_h drivers (evalh.hip)
#pragma once
#include <functional>

#include "common.hpp"
#include "evalh_gen.hpp"

namespace h2 {
// column pointers inside `d` are device pointers; the descriptor and its program arrays are host memory
int evalh_device(DeviceCtx* ctx, const h2_evalh_desc* d, Fr* d_values, hipStream_t stream, bool have_lock);
// everything in host memory
int evalh_host(DeviceCtx* ctx, const h2_evalh_desc* d, uint64_t* values);
// host memory, columns as COEFFICIENT vectors of 2^k elements (l_active_row extended): the cuda evaluate_h's shape; leases
// its devices itself -- the cosets of the extended domain are dealt over the pool (HALO2_PROOFS_N_GPU)
int evalh_host_coeffs(const h2_evalh_desc* d, uint64_t* values);
// the same, with what follows the evaluation kept on the device: when ONE device does all the cosets, `finish` is handed the
// 2^extended_k values where they lie (under the slot's lease, on its stream) instead of their being copied to `values`
// (which may then be null) and *finished is set; with several devices the values go to `values` as above
typedef std::function<int(DeviceCtx*, Fr*, hipStream_t)> EvalhFinish;
int evalh_host_coeffs(const h2_evalh_desc* d, uint64_t* values, const EvalhFinish* finish, bool* finished);
// the devices evalh_host_coeffs would deal the cosets of this descriptor over
uint32_t evalh_host_workers(const h2_evalh_desc* d);

// ---- the generated form (evalh_gen.cpp builds it, evalh.hip loads and launches it)
struct EvalhPlan;  // the loaded kernels of one program on one device
// The plan for this descriptor's program on the CURRENT device: from the cache, or generated + compiled + loaded now
// (*cached: 1 memory, 2 disk, 0 built now).  nullptr when generation is switched off (H2_EVALH_JIT=0) or unavailable (no
// hipRTC, a compile failure: reported once on stderr) -- the interpreter kernels then run the program.
//
// The cache is bounded (H2_EVALH_PLANS_MAX, default 128 programs x devices): past it the least recently used plan that
// nobody holds is unloaded (after its device has drained: launches are asynchronous) -- a prover service that sees new
// circuits for months does not accumulate code objects; an evicted program comes back from the disk cache.  A caller holds
// its plan through the reference below from the lookup until its launches are queued.
void evalh_plan_release(const EvalhPlan* plan);
class EvalhPlanRef {
    const EvalhPlan* p_ = nullptr;

public:
    EvalhPlanRef() = default;
    explicit EvalhPlanRef(const EvalhPlan* p) : p_(p) {}
    EvalhPlanRef(EvalhPlanRef&& o) noexcept : p_(o.p_) { o.p_ = nullptr; }
    EvalhPlanRef& operator=(EvalhPlanRef&& o) noexcept {
        if (this != &o) {
            if (p_) evalh_plan_release(p_);
            p_ = o.p_;
            o.p_ = nullptr;
        }
        return *this;
    }
    EvalhPlanRef(const EvalhPlanRef&) = delete;
    EvalhPlanRef& operator=(const EvalhPlanRef&) = delete;
    ~EvalhPlanRef() {
        if (p_) evalh_plan_release(p_);
    }
    explicit operator bool() const { return p_ != nullptr; }
    const EvalhPlan* get() const { return p_; }
};
EvalhPlanRef evalh_plan_get(const h2_evalh_desc* d, int* cached);
void evalh_plan_info(const EvalhPlan* plan, h2_evalh_info* info);
uint64_t evalh_plan_evictions();
void evalh_gen_info(const evgen::Generated& g, h2_evalh_info* info);
size_t evalh_fill_stage_args(const evgen::Stage& st, const h2_evalh_desc* d, Fr* values, const Fr* tw_lo, const Fr* tw_hi,
                             size_t row_begin, size_t row_end, unsigned char* buf, size_t cap);
uint64_t evalh_generated_launches();
}  // namespace h2
