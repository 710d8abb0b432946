// hostcopy.hip -- bulk copies between the CALLER's host vectors and device memory for the host-slice entry points.
//
// The reference hands `&[Fr]` slices of ordinary (pageable) memory to its GPU crate from several rayon workers at once
// (plonk/prover.rs:293-299, 643-646, 731-737; arithmetic.rs:351-352, 391-394, 507-508), and nearly every one of those vectors is
// NEW: a `Vec` the previous step has just filled.  Through hipMemcpyAsync a pageable copy is a blocking operation of the
// runtime that registers the caller's pages with the driver piecewise and feeds the DMA engine from the calling thread.  That
// is fast for a buffer the runtime has seen and whose pages are huge (numpy's allocations: 128 MiB in 2.4 ms, as page-locked
// memory) and slower for what a drop-in actually passes -- fresh 4 KiB-paged allocations (malloc / torch / a Rust `Vec`): an
// h2_intt of a new 32 MiB vector 2.3-3.0 ms against 1.34 ms for the same vector again (tools/experiments/pageable_intt_probe.py).
//
// Long copies from / into ordinary memory therefore go through the library's own page-locked staging: a few threads per
// transfer (H2_HOST_COPY_THREADS, default 2), each with a lane of two 8 MiB page-locked blocks, memcpy into / out of a block
// and queue ASYNCHRONOUS DMA from / into it on the caller's stream.  The link only ever sees page-locked memory -- no
// registration of the caller's pages, transfers of different calls interleave -- at 2.2-2.4 ms per 32 MiB round trip whatever
// the buffer's history.  Measured on the two drop-in proofs, alternating on one box (profiles/r6_host_copy_threads_ab.txt,
// r6_host_copy_modes_ab.txt): mini-PLONK k = 22 from ordinary memory 0.46-0.54 -> 0.40-0.47 s, the 64-column circuit at k = 20
// 1.77-1.90 -> 1.54-1.70 s.  On hugepage-backed buffers that the runtime already knows the staged path is the slower one (5.3 ->
// 6.6 ms per call: profiles/r6_pageable_calls_probe.txt), which is why it was measured and set aside twice before the
// fresh-vector case was measured.  H2_HOST_COPY_THREADS=0 selects the runtime's pageable path.  (Taking the runtime's long
// pageable copies one at a time per process was tried too: nothing on mini-PLONK, 1.8 -> 3.4 s on the 64-column proof.)
// Page-locked ranges (asynchronous DMA) and copies below 4 MiB always go straight to hipMemcpyAsync.
//
// Semantics are those of hipMemcpyAsync on pageable memory: an upload returns when the source has been read (the transfers
// themselves may still be in flight on the stream), a download returns when the destination holds the data.
#include <cstdlib>
#include <cstring>
#include <exception>
#include <thread>

#include "common.hpp"

namespace h2 {

namespace {
// ---- the staged path (H2_HOST_COPY_THREADS > 0) ----
constexpr size_t STAGE_CHUNK = (size_t)8 << 20;   // bytes per staging block (2 per lane)
constexpr size_t STAGE_MIN = (size_t)4 << 20;     // shorter copies: the runtime's own path

struct Lane {
    int dev = 0;
    char* slot[2] = {nullptr, nullptr};
    hipEvent_t ev[2];
    bool pending[2] = {false, false};   // a transfer queued from / into the slot whose completion nobody has waited for yet
};

std::mutex g_lane_mu;
std::vector<Lane*> g_idle_lanes;

int copy_threads() {
    static const int t = [] {
        const char* e = getenv("H2_HOST_COPY_THREADS");
        const int v = e ? atoi(e) : 2;
        return v < 0 ? 0 : (v > 16 ? 16 : v);
    }();
    return t;
}

Lane* lane_get(int dev) {
    {
        std::lock_guard<std::mutex> g(g_lane_mu);
        for (size_t i = 0; i < g_idle_lanes.size(); i++)
            if (g_idle_lanes[i]->dev == dev) {
                Lane* l = g_idle_lanes[i];
                g_idle_lanes.erase(g_idle_lanes.begin() + i);
                return l;
            }
    }
    Lane* l = new Lane;
    l->dev = dev;
    for (int i = 0; i < 2; i++) {
        H2_HIP(hipHostMalloc((void**)&l->slot[i], STAGE_CHUNK, hipHostMallocPortable));
        H2_HIP(hipEventCreateWithFlags(&l->ev[i], hipEventDisableTiming));
    }
    return l;
}

void lane_put(Lane* l) {
    std::lock_guard<std::mutex> g(g_lane_mu);
    g_idle_lanes.push_back(l);
}

void slot_wait(Lane* l, int sl) {
    if (l->pending[sl]) {
        H2_HIP(hipEventSynchronize(l->ev[sl]));
        l->pending[sl] = false;
    }
}

// the chunks t, t + T, t + 2T, ... of one transfer, through one lane
void lane_work(bool up, char* dev_ptr, char* host, size_t bytes, int t, int T, int dev, hipStream_t s) {
    H2_HIP(hipSetDevice(dev));
    Lane* l = lane_get(dev);
    struct Put {
        Lane* l;
        ~Put() { lane_put(l); }
    } put{l};
    const size_t chunks = (bytes + STAGE_CHUNK - 1) / STAGE_CHUNK;
    auto span = [&](size_t c, size_t* off, size_t* len) {
        *off = c * STAGE_CHUNK;
        *len = std::min(STAGE_CHUNK, bytes - *off);
    };
    size_t off, len;
    if (up) {
        size_t j = 0;
        for (size_t c = (size_t)t; c < chunks; c += (size_t)T, j++) {
            const int sl = (int)(j & 1);
            slot_wait(l, sl);
            span(c, &off, &len);
            memcpy(l->slot[sl], host + off, len);
            H2_HIP(hipMemcpyAsync(dev_ptr + off, l->slot[sl], len, hipMemcpyHostToDevice, s));
            H2_HIP(hipEventRecord(l->ev[sl], s));
            l->pending[sl] = true;       // (the lane's next user waits for it before it overwrites the block)
        }
        return;
    }
    slot_wait(l, 0);
    slot_wait(l, 1);
    auto issue = [&](size_t c, int sl) {
        span(c, &off, &len);
        H2_HIP(hipMemcpyAsync(l->slot[sl], dev_ptr + off, len, hipMemcpyDeviceToHost, s));
        H2_HIP(hipEventRecord(l->ev[sl], s));
        l->pending[sl] = true;
    };
    size_t j = 0;
    if ((size_t)t < chunks) issue((size_t)t, 0);
    for (size_t c = (size_t)t; c < chunks; c += (size_t)T, j++) {
        const int sl = (int)(j & 1);
        if (c + (size_t)T < chunks) issue(c + (size_t)T, sl ^ 1);
        slot_wait(l, sl);
        span(c, &off, &len);
        memcpy(host + off, l->slot[sl], len);
    }
}

void staged_copy(bool up, void* dev_ptr, void* host, size_t bytes, hipStream_t s) {
    int dev = 0;
    H2_HIP(hipGetDevice(&dev));
    const size_t chunks = (bytes + STAGE_CHUNK - 1) / STAGE_CHUNK;
    const int T = (int)std::min<size_t>((size_t)copy_threads(), chunks);
    std::vector<std::exception_ptr> failed((size_t)T);
    std::vector<std::thread> helpers;
    auto run = [&](int t) {
        try {
            lane_work(up, (char*)dev_ptr, (char*)host, bytes, t, T, dev, s);
        } catch (...) {
            failed[(size_t)t] = std::current_exception();
        }
    };
    for (int t = 1; t < T; t++) helpers.emplace_back(run, t);
    run(0);
    for (auto& h : helpers) h.join();
    for (auto& f : failed)
        if (f) std::rethrow_exception(f);
}

}  // namespace

// page-locked host memory (hipHostMalloc / h2_host_alloc_pinned / hipHostRegister)?
bool host_pinned(const void* p) {
    if (!p) return true;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

void host_upload(void* d_dst, const void* src, size_t bytes, hipStream_t s) {
    if (!bytes) return;
    if (bytes >= STAGE_MIN && copy_threads() && !host_pinned(src)) {
        staged_copy(true, d_dst, const_cast<void*>(src), bytes, s);
        return;
    }
    H2_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, s));
}

void host_download(void* dst, const void* d_src, size_t bytes, hipStream_t s) {
    if (!bytes) return;
    if (bytes >= STAGE_MIN && copy_threads() && !host_pinned(dst)) {
        staged_copy(false, const_cast<void*>(d_src), dst, bytes, s);
        return;
    }
    H2_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, s));
}

}  // namespace h2
