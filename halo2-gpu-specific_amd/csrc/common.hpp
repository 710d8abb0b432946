// common.hpp -- device context, error plumbing and grow-only device arenas shared by the
// NTT / MSM / elementwise translation units of libhalo2_hip.so.
//
// Mirrors the reference's process-global device pool (`N_GPU`, `GPU_LOCK`, `GPU_COND_VAR`,
// /root/reference/halo2_proofs/src/plonk/prover.rs:56-74; `acquire_gpu`/`release_gpu`,
// arithmetic.rs:314-331): one in-flight host-API operation per device, callers on any thread.
// Unlike the reference nothing is re-created per call: streams, scratch and twiddle plans are
// cached per device for the life of the process.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <condition_variable>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/halo2_hip.h"
#include "field.hpp"

namespace h2 {

void set_last_error(const std::string& msg);
const char* get_last_error();

struct HipError {
    hipError_t code;
    const char* expr;
    const char* file;
    int line;
};

#define H2_HIP(expr)                                                        \
    do {                                                                    \
        hipError_t h2_e_ = (expr);                                          \
        if (h2_e_ != hipSuccess) throw h2::HipError{h2_e_, #expr, __FILE__, __LINE__}; \
    } while (0)

// Grow-only device buffer (never shrinks; reused across calls).
struct DevBuf {
    void* ptr = nullptr;
    size_t cap = 0;
    void* get(size_t bytes);
    void release();
};

// Grow-only pinned host buffer (async device-to-host staging).
struct PinnedBuf {
    void* ptr = nullptr;
    size_t cap = 0;
    void* get(size_t bytes);
};

// Bulk copies between the CALLER's host vectors and device memory (hostcopy.hip): hipMemcpyAsync, with the long copies from /
// into ordinary (pageable) memory taken one at a time per process -- concurrent ones obstruct each other in the runtime.
// As hipMemcpyAsync on pageable memory: an upload returns when the source has been read, a download when `dst` holds the data.
bool host_pinned(const void* p);
void host_upload(void* d_dst, const void* src, size_t bytes, hipStream_t s);
void host_download(void* dst, const void* d_src, size_t bytes, hipStream_t s);

struct NttPlan;  // ntt.hip

// device copy of a registered host range of SRS points (msm.hip): `gen` is the registration it was uploaded for
struct ResidentCopy {
    void* ptr = nullptr;
    size_t len = 0;
    uint64_t gen = 0;
};

// What the host-API slots of ONE physical device share: the transform plans (twiddle tables), the device copies of
// registered SRS ranges (1 GiB + a 12 GiB table at k = 24: never per slot) and their accounting.
struct DeviceShared {
    std::mutex mu;                     // guards `resident` (a lookup may upload an SRS) and plan creation across the slots
    std::map<const void*, ResidentCopy> resident;  // registered host base ranges -> device copies (msm.hip)
    std::vector<ResidentCopy> retired;  // copies whose registration was replaced: freed by the next h2_bases_unregister, under every slot's lock
    std::map<std::string, NttPlan*> plans;
    size_t ntt_last_table_bytes = 0;   // of the optional last-pass tables (ntt.hip; guarded by ntt.hip's table mutex)
};

// One host-API SLOT of a device: its own streams, staging buffers and scratch, so that two host-slice calls on one device
// -- the reference's entry points are invoked concurrently from rayon workers (plonk/prover.rs:293-299, 731-737) --
// overlap one call's transfers with the other's kernels (H2_HOST_SLOTS, default 2; the reference runs one operation per
// device at a time, arithmetic.rs:314-331).  The h2_dev_* entry points use slot 0's context of the current device for
// plans only.
struct DeviceCtx {
    int device = -1;
    int slot = 0;
    hipStream_t stream = nullptr;      // compute stream for host-API calls
    hipStream_t copy_stream = nullptr; // H2D/D2H overlap
    hipStream_t aux_stream[2] = {nullptr, nullptr};  // third / fourth lane of the batched-MSM pipeline
    std::mutex mu;                     // one in-flight host-API op per slot
    DevBuf buf_a, buf_b, buf_c, buf_d; // staging / ping-pong scratch
    DevBuf msm_scratch;
    DevBuf evalh_scratch;
    DevBuf coeff_arena;                // h2_evaluate_h_coeff / h2_quotient_poly_coeff: the columns' coefficient and coset vectors of a call
    PinnedBuf pinned;
    DeviceShared* shared;
    std::map<const void*, ResidentCopy>& resident;
    std::map<std::string, NttPlan*>& plans;
    size_t& ntt_last_table_bytes;
    hipDeviceProp_t prop;
    explicit DeviceCtx(DeviceShared* s)
        : shared(s), resident(s->resident), plans(s->plans), ntt_last_table_bytes(s->ntt_last_table_bytes) {}
};

// Device pool (HALO2_PROOFS_N_GPU honoured, prover.rs:57-70).
int device_count();
DeviceCtx* ctx_for(int device);        // slot 0 of a physical device (the h2_dev_* entry points); creates on first use; throws HipError
DeviceCtx* ctx_for_entry(int entry);   // pool entry (acquire_device) -> its device's slot context
std::vector<DeviceCtx*> existing_contexts();  // the contexts created so far
int acquire_device();            // blocking free-list, arithmetic.rs:314-321
void release_device(int idx);    // arithmetic.rs:324-331

struct DeviceLease {
    int idx;
    DeviceCtx* ctx;
    DeviceLease() : idx(acquire_device()), ctx(nullptr) {
        try {
            ctx = ctx_for_entry(idx);
        } catch (...) {
            release_device(idx);
            throw;
        }
        ctx->mu.lock();  // two pool slots may map onto one physical device (idx % devices.len())
        (void)hipSetDevice(ctx->device);
    }
    ~DeviceLease() {
        ctx->mu.unlock();
        release_device(idx);
    }
};

template <class F>
int guarded(F&& f) {
    try {
        return f();
    } catch (const HipError& e) {
        char buf[512];
        snprintf(buf, sizeof buf, "HIP error %d (%s) in `%s` at %s:%d", (int)e.code, hipGetErrorString(e.code), e.expr,
                 e.file, e.line);
        set_last_error(buf);
        return e.code == hipErrorOutOfMemory ? H2_ERR_OOM : H2_ERR_HIP;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return H2_ERR_INVALID;
    }
}

}  // namespace h2
