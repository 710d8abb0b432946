// context.hip -- device pool, per-device context, error text.
// Reference counterpart: N_GPU / GPU_LOCK / GPU_COND_VAR (plonk/prover.rs:56-74) and
// acquire_gpu / release_gpu (arithmetic.rs:314-331).
#include <cstdlib>
#include <cstring>

#include "common.hpp"

// HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue
// serialise: the library alone runs a second MSM stream and a stream per host-API slot next to its caller's compute and
// copy streams.  Eight queues (measured: the wide circuit at k = 22 515 -> 452 ms while its columns went through the two-stream
// pipeline, 382 -> 373 ms from a compact witness on the final build; nothing else moves: profiles/r5_hw_queues_ab.txt).  The runtime reads
// the variable when it initialises -- at the first HIP call of the process -- so setting it when this library is LOADED is
// early enough for a host that uses HIP only through this library; a value the caller has set is kept.
// This changes HIP's behaviour for EVERY user of the runtime in the process, and setenv is not safe against other threads
// reading the environment: load the library before the host starts threads, or opt out with H2_NO_RUNTIME_DEFAULTS=1 and set
// GPU_MAX_HW_QUEUES (or not) yourself.
__attribute__((constructor)) static void h2_runtime_defaults() {
    const char* off = getenv("H2_NO_RUNTIME_DEFAULTS");
    if (off && *off && strcmp(off, "0") != 0) return;
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
}

namespace h2 {

static thread_local std::string g_last_error;
void set_last_error(const std::string& msg) { g_last_error = msg; }
const char* get_last_error() { return g_last_error.c_str(); }

void* DevBuf::get(size_t bytes) {
    if (bytes <= cap) return ptr;
    if (ptr) H2_HIP(hipFree(ptr));
    ptr = nullptr;
    cap = 0;
    size_t want = bytes + (bytes >> 3);  // 12.5 % headroom so nearby sizes reuse the block
    hipError_t e = hipMalloc(&ptr, want);
    if (e != hipSuccess) {
        want = bytes;
        H2_HIP(hipMalloc(&ptr, want));
    }
    cap = want;
    return ptr;
}
void* PinnedBuf::get(size_t bytes) {
    if (bytes <= cap) return ptr;
    if (ptr) H2_HIP(hipHostFree(ptr));
    ptr = nullptr;
    cap = 0;
    H2_HIP(hipHostMalloc(&ptr, bytes, hipHostMallocDefault));
    cap = bytes;
    return ptr;
}
void DevBuf::release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
}

namespace {
struct Pool {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<int> free_list;  // GPU_LOCK: Vec of free pool entries: device index + n_gpu * slot
    std::vector<DeviceCtx*> ctxs;        // [physical device * slots + slot]
    std::vector<DeviceShared*> shared;   // per physical device
    int n_gpu = -1;  // pool size (N_GPU)
    int n_visible = 0;
    int slots = 2;   // host-API slots per pool entry (H2_HOST_SLOTS)
    bool inited = false;

    void init() {
        if (inited) return;
        int cnt = 0;
        if (hipGetDeviceCount(&cnt) != hipSuccess) cnt = 0;
        n_visible = cnt;
        n_gpu = cnt;
        // HALO2_PROOFS_N_GPU overrides the pool size (prover.rs:57-70); indices wrap modulo the
        // visible devices like `devices[gpu_idx % devices.len()]` (arithmetic.rs:355).
        if (const char* env = std::getenv("HALO2_PROOFS_N_GPU")) {
            int v = std::atoi(env);
            if (v > 0) n_gpu = v;
        }
        if (cnt == 0) n_gpu = 0;
        if (const char* env = std::getenv("H2_HOST_SLOTS")) {
            int v = std::atoi(env);
            if (v >= 1 && v <= 4) slots = v;
        }
        // handed out from the back: every device's first slot before any second one
        for (int sub = slots - 1; sub >= 0; sub--)
            for (int i = n_gpu - 1; i >= 0; i--) free_list.push_back(i + n_gpu * sub);
        ctxs.assign(cnt > 0 ? (size_t)cnt * slots : 0, nullptr);
        shared.assign(cnt > 0 ? cnt : 0, nullptr);
        inited = true;
    }
};
Pool& pool() {
    static Pool p;
    return p;
}
}  // namespace

int device_count() {
    Pool& p = pool();
    std::lock_guard<std::mutex> g(p.mu);
    p.init();
    return p.n_gpu;
}

static DeviceCtx* ctx_at(int dev, int sub);

DeviceCtx* ctx_for(int device) {
    Pool& p = pool();
    {
        std::lock_guard<std::mutex> g(p.mu);
        p.init();
        if (p.n_visible == 0) throw HipError{hipErrorNoDevice, "no HIP device visible", __FILE__, __LINE__};
    }
    return ctx_at(device % p.n_visible, 0);
}

DeviceCtx* ctx_for_entry(int entry) {
    Pool& p = pool();
    {
        std::lock_guard<std::mutex> g(p.mu);
        p.init();
        if (p.n_visible == 0) throw HipError{hipErrorNoDevice, "no HIP device visible", __FILE__, __LINE__};
    }
    // pool entry = device index + n_gpu * slot; device indices wrap modulo the visible devices (arithmetic.rs:355)
    const int per = p.n_gpu > 0 ? p.n_gpu : 1;
    return ctx_at((entry % per) % p.n_visible, (entry / per) % p.slots);
}

static DeviceCtx* ctx_at(int dev, int sub) {
    Pool& p = pool();
    std::lock_guard<std::mutex> g(p.mu);
    const size_t at = (size_t)dev * p.slots + sub;
    if (!p.ctxs[at]) {
        if (!p.shared[dev]) p.shared[dev] = new DeviceShared();
        DeviceCtx* c = new DeviceCtx(p.shared[dev]);
        c->device = dev;
        c->slot = sub;
        H2_HIP(hipSetDevice(dev));
        H2_HIP(hipGetDeviceProperties(&c->prop, dev));
        H2_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        H2_HIP(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        for (int k = 0; k < 2; k++) H2_HIP(hipStreamCreateWithFlags(&c->aux_stream[k], hipStreamNonBlocking));
        p.ctxs[at] = c;
    }
    return p.ctxs[at];
}

std::vector<DeviceCtx*> existing_contexts() {
    Pool& p = pool();
    std::lock_guard<std::mutex> g(p.mu);
    std::vector<DeviceCtx*> out;
    for (DeviceCtx* c : p.ctxs)
        if (c) out.push_back(c);
    return out;
}

int acquire_device() {
    Pool& p = pool();
    std::unique_lock<std::mutex> lk(p.mu);
    p.init();
    if (p.n_gpu == 0) throw HipError{hipErrorNoDevice, "no HIP device visible", __FILE__, __LINE__};
    p.cv.wait(lk, [&] { return !p.free_list.empty(); });
    int idx = p.free_list.back();
    p.free_list.pop_back();
    return idx;
}

void release_device(int idx) {
    Pool& p = pool();
    {
        std::lock_guard<std::mutex> g(p.mu);
        p.free_list.push_back(idx);
    }
    p.cv.notify_one();
}

}  // namespace h2
