// capi.hip -- the extern "C" boundary declared in include/halo2_hip.h.
// Host-buffer entry points = the reference's gpu_* functions (arithmetic.rs:134-534) with the
// per-call program creation removed: a pooled device context keeps streams, scratch and
// twiddle plans alive across calls.  h2_dev_* = the same ops on device-resident data.
#include <cstring>
#include <sys/mman.h>
#include <thread>

#include "common.hpp"
#include "evalh.hpp"
#include "msm.hpp"
#include "ntt.hpp"
#include "poly.hpp"
#include "logup.hpp"
#include "scan.hpp"

using namespace h2;

namespace {

hipStream_t pick_stream(DeviceCtx* ctx, void* stream) { return stream ? (hipStream_t)stream : ctx->stream; }

DeviceCtx* current_ctx() {
    int dev = 0;
    H2_HIP(hipGetDevice(&dev));
    return ctx_for(dev);
}

int bad(const char* msg) {
    set_last_error(msg);
    return H2_ERR_INVALID;
}

// plan lookup is the only shared mutable state touched by the h2_dev_* paths
PlanRef plan_locked(DeviceCtx* ctx, uint32_t log_n, const uint64_t omega[4], hipStream_t s, bool have_lock) {
    if (have_lock) return ntt_get_plan(ctx, log_n, omega, s);
    std::lock_guard<std::mutex> g(ctx->mu);
    return ntt_get_plan(ctx, log_n, omega, s);
}

int dev_ntt_impl(DeviceCtx* ctx, const Fr* src, Fr* dst, Fr* tmp, uint32_t in_len, const uint64_t omega[4],
                 uint32_t log_n, const Fr* pre3, const Fr* post3, hipStream_t s, bool have_lock) {
    if (log_n > 28) return bad("log_n exceeds the 2-adicity of Fr (S = 28)");
    std::vector<uint32_t> bits;
    ntt_split(log_n, bits);
    if (bits.size() >= 2 && tmp == nullptr) return bad("NTT of this size needs a scratch buffer (d_tmp)");
    PlanRef pl = plan_locked(ctx, log_n, omega, s, have_lock);  // pinned until the passes are launched
    ntt_run(ctx, pl.get(), src, dst, tmp, in_len, pre3, post3, s);
    return H2_OK;
}

int dev_extended_to_coeff_impl(DeviceCtx* ctx, Fr* d_a, Fr* d_tmp, uint32_t extended_k, const uint64_t g_coset[4],
                               const uint64_t g_coset_inv[4], const uint64_t extended_omega_inv[4],
                               const uint64_t extended_ifft_divisor[4], hipStream_t s, bool have_lock) {
    // into_coset = false: coset_powers = [g_coset_inv, g_coset] (domain.rs:385-387), fused with the
    // iFFT divisor: y[i] *= divisor * {1, g_coset_inv, g_coset}[i % 3].
    Fr d = fr_from_u64x4(extended_ifft_divisor);
    Fr gi = fr_from_u64x4(g_coset_inv), g = fr_from_u64x4(g_coset);
    Fr post3[3] = {d, fp_mul(d, gi), fp_mul(d, g)};  // host-side Montgomery products
    return dev_ntt_impl(ctx, d_a, d_a, d_tmp, 1u << extended_k, extended_omega_inv, extended_k, nullptr, post3, s,
                        have_lock);
}

// A host vector that a call only READS: the device copy of a range registered with h2_poly_register (uploaded once per device:
// the proving key's coefficient forms, a proof's final polynomials), or nullptr -- the caller then uploads as the reference does.
const Fr* resident_operand(DeviceCtx* ctx, const uint64_t* host, size_t n) { return host ? poly_resident(ctx, host, n) : nullptr; }

// Elementwise host-slice operations over long vectors run as a pipeline of chunks on three streams of the slot: chunk c + 1
// crosses PCIe on the copy stream while the kernel of chunk c runs on the compute stream and the result of chunk c - 1 leaves
// on the third -- the two directions of the link at once, instead of upload-all -> compute -> download-all on one stream.
// Two staging slots; `up` / `run` / `down` enqueue on the stream they are handed.  From page-locked host memory the copies are
// asynchronous DMA and the overlap is real; from ordinary memory hipMemcpyAsync stages and returns when the source has been
// read, so the chunks follow one another as before (no worse: the same bytes in smaller pieces).
constexpr size_t PIPE_CHUNK = (size_t)1 << 19;      // elements per chunk: 16 MiB
constexpr size_t PIPE_MIN = (size_t)1 << 21;        // shorter vectors keep the single-shot form
template <class Up, class Run, class Down>
void pipeline_chunks(DeviceCtx* ctx, size_t total, Up up, Run run, Down down) {
    hipEvent_t in_done[2], k_done[2], out_done[2];
    for (int i = 0; i < 2; i++) {
        H2_HIP(hipEventCreateWithFlags(&in_done[i], hipEventDisableTiming));
        H2_HIP(hipEventCreateWithFlags(&k_done[i], hipEventDisableTiming));
        H2_HIP(hipEventCreateWithFlags(&out_done[i], hipEventDisableTiming));
    }
    struct Guard {
        hipEvent_t* a;
        hipEvent_t* b;
        hipEvent_t* c;
        DeviceCtx* ctx;
        ~Guard() {
            // (also on the way out of a throw: nothing may still be reading the staging slots when the caller's frame goes)
            (void)hipStreamSynchronize(ctx->copy_stream);
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipStreamSynchronize(ctx->aux_stream[0]);
            for (int i = 0; i < 2; i++) {
                (void)hipEventDestroy(a[i]);
                (void)hipEventDestroy(b[i]);
                (void)hipEventDestroy(c[i]);
            }
        }
    } guard{in_done, k_done, out_done, ctx};
    size_t c = 0;
    for (size_t off = 0; off < total; off += PIPE_CHUNK, c++) {
        const size_t len = std::min(PIPE_CHUNK, total - off);
        const int slot = (int)(c & 1);
        if (c >= 2) H2_HIP(hipStreamWaitEvent(ctx->copy_stream, out_done[slot], 0));   // the slot's previous result has left
        up(off, len, slot, ctx->copy_stream);
        H2_HIP(hipEventRecord(in_done[slot], ctx->copy_stream));
        H2_HIP(hipStreamWaitEvent(ctx->stream, in_done[slot], 0));
        run(off, len, slot, ctx->stream);
        H2_HIP(hipEventRecord(k_done[slot], ctx->stream));
        H2_HIP(hipStreamWaitEvent(ctx->aux_stream[0], k_done[slot], 0));
        down(off, len, slot, ctx->aux_stream[0]);
        H2_HIP(hipEventRecord(out_done[slot], ctx->aux_stream[0]));
    }
    H2_HIP(hipStreamSynchronize(ctx->aux_stream[0]));
    H2_HIP(hipStreamSynchronize(ctx->stream));
}
bool pipeline_enabled() {
    static const bool on = !(getenv("H2_HOST_PIPELINE") && atoi(getenv("H2_HOST_PIPELINE")) == 0);
    return on;
}
// page-locked host memory (hipHostMalloc / h2_host_alloc_pinned / hipHostRegister)?  Only then are the chunk copies asynchronous
// DMA; from ordinary memory every hipMemcpyAsync stages and blocks, and sixteen small blocking copies are slower than one large
// one (measured: the k = 22 drop-in proof from ordinary memory 1.25 -> 1.52 s with the pipeline forced on): single shot there.
bool use_pipeline(size_t size, std::initializer_list<const void*> host) {
    if (!pipeline_enabled() || size < PIPE_MIN) return false;
    for (const void* p : host)
        if (!host_pinned(p)) return false;
    return true;
}
// A result vector in ORDINARY host memory that nobody has touched yet -- what every operation of the reference's data flow
// returns: a fresh `Vec` -- faults its pages in one by one under the device-to-host copy, on the runtime's single staging thread:
// 21 ms for the 128 MiB of a k = 22 vector against 2.5 ms on the link (tools/experiments/hostreg_probe.py).  The pages of the
// destination are populated HERE instead, by a few threads at once -- huge pages advised, then one read-write touch per 4 KiB
// page of the slice -- started when the call begins, under its uploads and kernels, and joined before the copy back is issued.
// Touching beats MADV_POPULATE_WRITE on the same advised range (a 128 MiB result: 5.6 ms per call against 8.4, 5.3 ms when the
// pages already exist; four calls at once 19.6-20 ms against 27-31: tools/experiments/prefault_mode_ab.sh);
// H2_HOST_PREFAULT_POPULATE=1 selects the madvise form.  Page-locked destinations and short ones are left alone;
// H2_HOST_PREFAULT=<threads> (default 4; 0 switches it off).
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
struct Prefault {
    std::vector<std::thread> workers;
    Prefault() = default;
    Prefault(void* dst, size_t bytes) { start(dst, bytes); }
    void start(void* dst, size_t bytes) {
        static const int threads = [] {
            const char* e = getenv("H2_HOST_PREFAULT");
            const int v = e ? atoi(e) : 4;
            return v < 0 ? 0 : (v > 32 ? 32 : v);
        }();
        if (!threads || !dst || bytes < ((size_t)4 << 20) || host_pinned(dst)) return;
        const uintptr_t page = 4096, lo = ((uintptr_t)dst + page - 1) & ~(page - 1), hi = ((uintptr_t)dst + bytes) & ~(page - 1);
        if (hi <= lo) return;
        // huge pages where the kernel grants them (transparent_hugepage = madvise or always): 512 times fewer faults to take
        (void)madvise((void*)lo, hi - lo, MADV_HUGEPAGE);
        const size_t pages = (hi - lo) / page, per = (((pages + threads - 1) / threads) + 511) & ~(size_t)511;   // slices of whole 2 MiB
        for (int t = 0; t < threads; t++) {
            const size_t first = (size_t)t * per, count = first < pages ? std::min(per, pages - first) : 0;
            if (!count) break;
            char* at = (char*)lo + first * page;
            workers.emplace_back([at, count] {
                static const bool populate = getenv("H2_HOST_PREFAULT_POPULATE") && atoi(getenv("H2_HOST_PREFAULT_POPULATE")) != 0;
                if (populate && madvise(at, count * page, MADV_POPULATE_WRITE) == 0) return;
                for (size_t i = 0; i < count; i++) {   // (the value written is the one read)
                    volatile char* q = at + i * page;
                    *q = *q;
                }
            });
        }
    }
    void join() {
        for (std::thread& w : workers) w.join();
        workers.clear();
    }
    ~Prefault() { join(); }
};

void fr_to_u64x4(const Fr& v, uint64_t out[4]) {
    for (int i = 0; i < 4; i++) out[i] = (uint64_t)v.l[2 * i] | ((uint64_t)v.l[2 * i + 1] << 32);
}

}  // namespace

extern "C" {

int h2_version(void) { return 1; }

int h2_device_count(void) { return device_count(); }

const char* h2_last_error(void) { return get_last_error(); }

int h2_synchronize(void) {
    return guarded([&] {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
        for (int d = 0; d < n; d++) {
            H2_HIP(hipSetDevice(d));
            H2_HIP(hipDeviceSynchronize());
        }
        return (int)H2_OK;
    });
}

// ------------------------------------------------------------------ device memory / streams for hosts without a HIP binding
int h2_set_device(int device) {
    return guarded([&] {
        H2_HIP(hipSetDevice(device));
        return (int)H2_OK;
    });
}

int h2_dev_alloc(size_t bytes, void** d_out) {
    if (!d_out) return bad("h2_dev_alloc: null argument");
    return guarded([&] {
        *d_out = nullptr;
        if (bytes) H2_HIP(hipMalloc(d_out, bytes));
        return (int)H2_OK;
    });
}

int h2_dev_free(void* d_ptr) {
    return guarded([&] {
        if (d_ptr) H2_HIP(hipFree(d_ptr));
        return (int)H2_OK;
    });
}

int h2_host_alloc_pinned(size_t bytes, void** out) {
    if (!out) return bad("h2_host_alloc_pinned: null argument");
    return guarded([&] {
        *out = nullptr;
        if (bytes) H2_HIP(hipHostMalloc(out, bytes, hipHostMallocDefault));
        return (int)H2_OK;
    });
}

int h2_host_free_pinned(void* ptr) {
    return guarded([&] {
        if (ptr) H2_HIP(hipHostFree(ptr));
        return (int)H2_OK;
    });
}

int h2_stream_create(void** stream_out) {
    if (!stream_out) return bad("h2_stream_create: null argument");
    return guarded([&] {
        hipStream_t s = nullptr;
        H2_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        *stream_out = (void*)s;
        return (int)H2_OK;
    });
}

int h2_stream_destroy(void* stream) {
    return guarded([&] {
        if (stream) H2_HIP(hipStreamDestroy((hipStream_t)stream));
        return (int)H2_OK;
    });
}

int h2_stream_synchronize(void* stream) {
    return guarded([&] {
        H2_HIP(hipStreamSynchronize(pick_stream(current_ctx(), stream)));
        return (int)H2_OK;
    });
}

int h2_dev_upload(void* d_dst, const void* src, size_t bytes, void* stream) {
    if (bytes && (!d_dst || !src)) return bad("h2_dev_upload: null argument");
    return guarded([&] {
        if (bytes) host_upload(d_dst, src, bytes, pick_stream(current_ctx(), stream));
        return (int)H2_OK;
    });
}

int h2_dev_download(void* dst, const void* d_src, size_t bytes, void* stream) {
    if (bytes && (!dst || !d_src)) return bad("h2_dev_download: null argument");
    return guarded([&] {
        hipStream_t s = pick_stream(current_ctx(), stream);
        if (bytes) host_download(dst, d_src, bytes, s);
        H2_HIP(hipStreamSynchronize(s));
        return (int)H2_OK;
    });
}

// ------------------------------------------------------------------ library-held device memory
int h2_release_plans(void) {
    return guarded([&] {
        // the calling thread's current device is restored on every way out (a throw from H2_HIP included): later h2_dev_*
        // calls of this thread must not find themselves on another device
        struct RestoreDevice {
            int prev = 0;
            RestoreDevice() { (void)hipGetDevice(&prev); }
            ~RestoreDevice() { (void)hipSetDevice(prev); }
        } restore;
        for (DeviceCtx* ctx : existing_contexts()) {
            std::vector<NttPlan*> gone;
            H2_HIP(hipSetDevice(ctx->device));
            {
                std::lock_guard<std::mutex> g(ctx->mu);      // no transform of this context is between lookup and launch
                ntt_detach_idle_plans(ctx, gone);
                ctx->coeff_arena.release();                   // (no call of the slot is running: its block of column vectors goes too)
            }
            ntt_free_plans(gone);                             // synchronises and frees with no lock held
        }
        return (int)H2_OK;
    });
}

int h2_set_table_budget(size_t bytes) {
    ntt_set_table_budget(bytes);
    return H2_OK;
}

size_t h2_library_memory_bytes(void) {
    size_t total = 0;
    int rc = guarded([&] {
        DeviceCtx* ctx = current_ctx();
        {
            std::lock_guard<std::mutex> g(ctx->mu);
            total = ntt_plan_bytes(ctx) + msm_library_bytes(ctx);          // shared by the host-API slots of the device
        }
        for (DeviceCtx* c : existing_contexts()) {                        // ... plus every slot's own buffers
            if (c->device != ctx->device) continue;
            std::lock_guard<std::mutex> g(c->mu);
            total += c->buf_a.cap + c->buf_b.cap + c->buf_c.cap + c->buf_d.cap + c->msm_scratch.cap + c->evalh_scratch.cap + c->coeff_arena.cap;
        }
        return (int)H2_OK;
    });
    return rc == H2_OK ? total : 0;
}

// ------------------------------------------------------------------ NTT, host buffers
int h2_ntt(uint64_t* a, const uint64_t omega[4], uint32_t log_n) {
    if (!a || !omega) return bad("h2_ntt: null argument");
    return guarded([&] {
        DeviceLease lease;
        DeviceCtx* ctx = lease.ctx;
        size_t bytes = sizeof(Fr) << log_n;
        Fr* d_a = (Fr*)ctx->buf_a.get(bytes);
        Fr* d_t = (Fr*)ctx->buf_b.get(bytes);
        host_upload(d_a, a, bytes, ctx->stream);
        int rc = dev_ntt_impl(ctx, d_a, d_a, d_t, 1u << log_n, omega, log_n, nullptr, nullptr, ctx->stream, true);
        if (rc != H2_OK) return rc;
        host_download(a, d_a, bytes, ctx->stream);
        H2_HIP(hipStreamSynchronize(ctx->stream));
        return (int)H2_OK;
    });
}

int h2_intt(uint64_t* a, const uint64_t omega_inv[4], const uint64_t divisor[4], uint32_t log_n) {
    if (!a || !omega_inv || !divisor) return bad("h2_intt: null argument");
    return guarded([&] {
        DeviceLease lease;
        DeviceCtx* ctx = lease.ctx;
        size_t bytes = sizeof(Fr) << log_n;
        Fr* d_a = (Fr*)ctx->buf_a.get(bytes);
        Fr* d_t = (Fr*)ctx->buf_b.get(bytes);
        Fr d = fr_from_u64x4(divisor);
        Fr post3[3] = {d, d, d};
        host_upload(d_a, a, bytes, ctx->stream);
        int rc = dev_ntt_impl(ctx, d_a, d_a, d_t, 1u << log_n, omega_inv, log_n, nullptr, post3, ctx->stream, true);
        if (rc != H2_OK) return rc;
        host_download(a, d_a, bytes, ctx->stream);
        H2_HIP(hipStreamSynchronize(ctx->stream));
        return (int)H2_OK;
    });
}

// lagrange_to_coeff of a vector the caller KEEPS (plonk/prover.rs:643-646: `domain.lagrange_to_coeff(advice_values.clone())` per
// column): the values are read where they are, the coefficients written to `out` -- no host copy of the column first
int h2_intt_to(const uint64_t* a, uint64_t* out, const uint64_t omega_inv[4], const uint64_t divisor[4], uint32_t log_n) {
    if (!a || !out || !omega_inv || !divisor) return bad("h2_intt_to: null argument");
    return guarded([&] {
        DeviceLease lease;
        DeviceCtx* ctx = lease.ctx;
        size_t bytes = sizeof(Fr) << log_n;
        Prefault pf((const void*)out == (const void*)a ? nullptr : out, bytes);
        Fr* d_a = (Fr*)ctx->buf_a.get(bytes);
        Fr* d_t = (Fr*)ctx->buf_b.get(bytes);
        Fr d = fr_from_u64x4(divisor);
        Fr post3[3] = {d, d, d};
        const Fr* src = resident_operand(ctx, a, (size_t)1 << log_n);
        if (!src) {
            host_upload(d_a, a, bytes, ctx->stream);
            src = d_a;
        }
        int rc = dev_ntt_impl(ctx, src, d_a, d_t, 1u << log_n, omega_inv, log_n, nullptr, post3, ctx->stream, true);
        if (rc != H2_OK) return rc;
        pf.join();
        host_download(out, d_a, bytes, ctx->stream);
        H2_HIP(hipStreamSynchronize(ctx->stream));
        return (int)H2_OK;
    });
}

int h2_coeff_to_extended(const uint64_t* coeffs, uint64_t* out, uint32_t k, uint32_t extended_k,
                         const uint64_t g_coset[4], const uint64_t g_coset_inv[4], const uint64_t extended_omega[4]) {
    if (!coeffs || !out || !g_coset || !g_coset_inv || !extended_omega) return bad("h2_coeff_to_extended: null argument");
    if (extended_k < k) return bad("h2_coeff_to_extended: extended_k < k");
    return guarded([&] {
        DeviceLease lease;
        DeviceCtx* ctx = lease.ctx;
        size_t in_bytes = sizeof(Fr) << k, ext_bytes = sizeof(Fr) << extended_k;
        Prefault pf(out, ext_bytes);
        Fr* d_in = (Fr*)ctx->buf_a.get(ext_bytes);
        Fr* d_t = (Fr*)ctx->buf_b.get(ext_bytes);
        // into_coset = true: coset_powers = [g_coset, g_coset_inv] (domain.rs:383-385)
        Fr pre3[3] = {fr_from_u64x4(g_coset), fr_from_u64x4(g_coset), fr_from_u64x4(g_coset_inv)};
        const Fr* src = resident_operand(ctx, coeffs, (size_t)1 << k);
        if (!src) {
            host_upload(d_in, coeffs, in_bytes, ctx->stream);
            src = d_in;
        }
        int rc = dev_ntt_impl(ctx, src, d_in, d_t, 1u << k, extended_omega, extended_k, pre3, nullptr, ctx->stream, true);
        if (rc != H2_OK) return rc;
        pf.join();
        host_download(out, d_in, ext_bytes, ctx->stream);
        H2_HIP(hipStreamSynchronize(ctx->stream));
        return (int)H2_OK;
    });
}

int h2_extended_to_coeff(const uint64_t* a, uint64_t* out, size_t out_len, uint32_t extended_k,
                         const uint64_t g_coset[4], const uint64_t g_coset_inv[4],
                         const uint64_t extended_omega_inv[4], const uint64_t extended_ifft_divisor[4]) {
    if (!a || !out || !g_coset || !g_coset_inv || !extended_omega_inv || !extended_ifft_divisor)
        return bad("h2_extended_to_coeff: null argument");
    if (out_len > ((size_t)1 << extended_k)) return bad("h2_extended_to_coeff: out_len exceeds the extended domain");
    return guarded([&] {
        DeviceLease lease;
        DeviceCtx* ctx = lease.ctx;
        size_t ext_bytes = sizeof(Fr) << extended_k;
        Prefault pf(out == a ? nullptr : out, out_len * sizeof(Fr));
        Fr* d_a = (Fr*)ctx->buf_a.get(ext_bytes);
        Fr* d_t = (Fr*)ctx->buf_b.get(ext_bytes);
        host_upload(d_a, a, ext_bytes, ctx->stream);
        int rc = dev_extended_to_coeff_impl(ctx, d_a, d_t, extended_k, g_coset, g_coset_inv, extended_omega_inv,
                                            extended_ifft_divisor, ctx->stream, true);
        if (rc != H2_OK) return rc;
        pf.join();
        host_download(out, d_a, out_len * sizeof(Fr), ctx->stream);
        H2_HIP(hipStreamSynchronize(ctx->stream));
        return (int)H2_OK;
    });
}

// ------------------------------------------------------------------ Montgomery conversion
static int host_batch_mont(uint64_t* a, size_t n, bool to_mont) {
    if (!a && n) return bad("h2_batch_mont: null argument");
    return guarded([&] {
        DeviceLease lease;
        DeviceCtx* ctx = lease.ctx;
        size_t bytes = n * sizeof(Fr);
        if (n == 0) return (int)H2_OK;
        if (use_pipeline(n, {a})) {
            Fr* slots = (Fr*)ctx->buf_a.get(2 * PIPE_CHUNK * sizeof(Fr));
            int rc = H2_OK;
            pipeline_chunks(ctx, n,
                [&](size_t off, size_t len, int slot, hipStream_t st) {
                    host_upload(slots + slot * PIPE_CHUNK, a + 4 * off, len * sizeof(Fr), st);
                },
                [&](size_t, size_t len, int slot, hipStream_t st) {
                    int r = batch_mont_launch(slots + slot * PIPE_CHUNK, len, to_mont, st);
                    if (r != H2_OK) rc = r;
                },
                [&](size_t off, size_t len, int slot, hipStream_t st) {
                    host_download(a + 4 * off, slots + slot * PIPE_CHUNK, len * sizeof(Fr), st);
                });
            return rc;
        }
        Fr* d_a = (Fr*)ctx->buf_a.get(bytes);
        host_upload(d_a, a, bytes, ctx->stream);
        int rc = batch_mont_launch(d_a, n, to_mont, ctx->stream);
        if (rc != H2_OK) return rc;
        host_download(a, d_a, bytes, ctx->stream);
        H2_HIP(hipStreamSynchronize(ctx->stream));
        return (int)H2_OK;
    });
}
int h2_batch_mont(uint64_t* a, size_t n) { return host_batch_mont(a, n, true); }
int h2_batch_unmont(uint64_t* a, size_t n) { return host_batch_mont(a, n, false); }

// ------------------------------------------------------------------ elementwise, host buffers
int h2_eval_op(int op, uint64_t* res, const uint64_t* l, const uint64_t* r, int32_t l_rot, int32_t r_rot, size_t size,
               const uint64_t c[4]) {
    if (!res) return bad("h2_eval_op: null result");
    return guarded([&] {
        DeviceLease lease;
        DeviceCtx* ctx = lease.ctx;
        size_t bytes = size * sizeof(Fr);
        if (size == 0) return (int)H2_OK;
        const Fr* res_l = resident_operand(ctx, l, size);
        const Fr* res_r = resident_operand(ctx, r, size);
        if (l_rot == 0 && r_rot == 0 && use_pipeline(size, {res, res_l ? nullptr : (const void*)l, res_r ? nullptr : (const void*)r})) {
            // no rotation: element i depends on element i of the operands only -- chunk by chunk, both directions of PCIe at once
            Fr* sl = (l && !res_l) ? (Fr*)ctx->buf_b.get(2 * PIPE_CHUNK * sizeof(Fr)) : nullptr;
            Fr* sr = (r && !res_r) ? (Fr*)ctx->buf_c.get(2 * PIPE_CHUNK * sizeof(Fr)) : nullptr;
            Fr* so = (Fr*)ctx->buf_a.get(2 * PIPE_CHUNK * sizeof(Fr));
            int rc = H2_OK;
            pipeline_chunks(ctx, size,
                [&](size_t off, size_t len, int slot, hipStream_t st) {
                    if (sl) host_upload(sl + slot * PIPE_CHUNK, l + 4 * off, len * sizeof(Fr), st);
                    if (sr) host_upload(sr + slot * PIPE_CHUNK, r + 4 * off, len * sizeof(Fr), st);
                },
                [&](size_t off, size_t len, int slot, hipStream_t st) {
                    const Fr* pl = !l ? nullptr : (res_l ? res_l + off : sl + slot * PIPE_CHUNK);
                    const Fr* pr = !r ? nullptr : (res_r ? res_r + off : sr + slot * PIPE_CHUNK);
                    int rr = eval_op_launch(op, so + slot * PIPE_CHUNK, pl, pr, 0, 0, len, c, st);
                    if (rr != H2_OK) rc = rr;
                },
                [&](size_t off, size_t len, int slot, hipStream_t st) {
                    host_download(res + 4 * off, so + slot * PIPE_CHUNK, len * sizeof(Fr), st);
                });
            return rc;
        }
        Prefault pf((const void*)res == (const void*)l || (const void*)res == (const void*)r ? nullptr : res, bytes);
        Fr* d_res = (Fr*)ctx->buf_a.get(bytes);
        const Fr* d_l = res_l;
        const Fr* d_r = res_r;
        if (l && !d_l) {
            Fr* up = (Fr*)ctx->buf_b.get(bytes);
            host_upload(up, l, bytes, ctx->stream);
            d_l = up;
        }
        if (r && !d_r) {
            Fr* up = (Fr*)ctx->buf_c.get(bytes);
            host_upload(up, r, bytes, ctx->stream);
            d_r = up;
        }
        int rc = eval_op_launch(op, d_res, d_l, d_r, l_rot, r_rot, size, c, ctx->stream);
        if (rc != H2_OK) return rc;
        pf.join();
        host_download(res, d_res, bytes, ctx->stream);
        H2_HIP(hipStreamSynchronize(ctx->stream));
        return (int)H2_OK;
    });
}

int h2_divide_by_vanishing_poly(uint64_t* a, size_t size, const uint64_t* t_evaluations, size_t t_len) {
    if (!a || !t_evaluations) return bad("h2_divide_by_vanishing_poly: null argument");
    return guarded([&] {
        DeviceLease lease;
        DeviceCtx* ctx = lease.ctx;
        if (size == 0) return (int)H2_OK;
        if (t_len && PIPE_CHUNK % t_len == 0 && use_pipeline(size, {a})) {
            // a[i] *= t[i % t_len]: a chunk that starts at a multiple of t_len sees the table from its first entry
            Fr* slots = (Fr*)ctx->buf_a.get(2 * PIPE_CHUNK * sizeof(Fr));
            Fr* d_t = (Fr*)ctx->buf_b.get(t_len * sizeof(Fr));
            host_upload(d_t, t_evaluations, t_len * sizeof(Fr), ctx->stream);
            H2_HIP(hipStreamSynchronize(ctx->stream));
            int rc = H2_OK;
            pipeline_chunks(ctx, size,
                [&](size_t off, size_t len, int slot, hipStream_t st) {
                    host_upload(slots + slot * PIPE_CHUNK, a + 4 * off, len * sizeof(Fr), st);
                },
                [&](size_t, size_t len, int slot, hipStream_t st) {
                    int r = divide_by_vanishing_launch(slots + slot * PIPE_CHUNK, len, d_t, t_len, st);
                    if (r != H2_OK) rc = r;
                },
                [&](size_t off, size_t len, int slot, hipStream_t st) {
                    host_download(a + 4 * off, slots + slot * PIPE_CHUNK, len * sizeof(Fr), st);
                });
            return rc;
        }
        Fr* d_a = (Fr*)ctx->buf_a.get(size * sizeof(Fr));
        Fr* d_t = (Fr*)ctx->buf_b.get(t_len * sizeof(Fr));
        host_upload(d_a, a, size * sizeof(Fr), ctx->stream);
        host_upload(d_t, t_evaluations, t_len * sizeof(Fr), ctx->stream);
        int rc = divide_by_vanishing_launch(d_a, size, d_t, t_len, ctx->stream);
        if (rc != H2_OK) return rc;
        host_download(a, d_a, size * sizeof(Fr), ctx->stream);
        H2_HIP(hipStreamSynchronize(ctx->stream));
        return (int)H2_OK;
    });
}

// ------------------------------------------------------------------ MSM, host buffers
int h2_msm(const uint64_t* scalars, const uint64_t* bases, size_t n, uint32_t max_bits, uint64_t out_xyz[12]) {
    if (!out_xyz || (n && (!scalars || !bases))) return bad("h2_msm: null argument");
    return guarded([&] {
        if (max_bits == 0 || n == 0) {  // arithmetic.rs:346, :421
            msm_identity(out_xyz);
            return (int)H2_OK;
        }
        DeviceLease lease;
        return msm_host(lease.ctx, scalars, bases, n, max_bits, out_xyz);
    });
}

int h2_poly_register(const uint64_t* values, size_t n) {
    if (!values || n == 0) return bad("h2_poly_register: null / empty range");
    return poly_register(values, n);
}

int h2_poly_unregister(const uint64_t* values) {
    if (!values) return bad("h2_poly_unregister: null argument");
    return guarded([&] { return bases_unregister(values); });   // one registry: the device copies are freed the same way
}

int h2_bases_register(const uint64_t* bases, size_t n) {
    if (!bases || n == 0) return bad("h2_bases_register: null / empty range");
    return bases_register(bases, n);
}
int h2_bases_unregister(const uint64_t* bases) {
    if (!bases) return bad("h2_bases_unregister: null");
    return bases_unregister(bases);
}

int h2_g1_sum(const uint64_t* points_xyz, size_t count, uint64_t out_xyz[12]) {
    if (!out_xyz || (count && !points_xyz)) return bad("h2_g1_sum: null argument");
    g1_sum_host(points_xyz, count, out_xyz);
    return H2_OK;
}

int h2_dev_g1_fold(const void* d_points_xyz, uint32_t world, uint32_t count, void* d_out_xyz, void* stream) {
    if (count && (!d_points_xyz || !d_out_xyz)) return bad("h2_dev_g1_fold: null argument");
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        return g1_fold_launch((const uint64_t*)d_points_xyz, world, count, (uint64_t*)d_out_xyz, pick_stream(ctx, stream));
    });
}

int h2_msm_multi(const uint64_t* scalars, const uint64_t* bases, size_t n, uint32_t max_bits, uint64_t out_xyz[12]) {
    if (!out_xyz || (n && (!scalars || !bases))) return bad("h2_msm_multi: null argument");
    return guarded([&] {
        if (max_bits == 0 || n == 0) {
            msm_identity(out_xyz);
            return (int)H2_OK;
        }
        return msm_host_multi(scalars, bases, n, max_bits, out_xyz);
    });
}

int h2_msm_intt(uint64_t* scalars, const uint64_t* bases, size_t n, uint32_t max_bits, const uint64_t omega_inv[4],
                const uint64_t divisor[4], uint32_t log_n, uint64_t out_xyz[12]) {
    if (!out_xyz || !scalars || !bases || !omega_inv || !divisor) return bad("h2_msm_intt: null argument");
    if (n != ((size_t)1 << log_n)) return bad("h2_msm_intt: n != 2^log_n");
    return guarded([&] {
        DeviceLease lease;
        DeviceCtx* ctx = lease.ctx;
        size_t sbytes = n * sizeof(Fr);
        // one upload of the scalars feeds both the MSM and the iNTT (arithmetic.rs:402-404)
        Fr* d_s = (Fr*)ctx->buf_a.get(sbytes);
        Fr* d_t = (Fr*)ctx->buf_b.get(sbytes);
        host_upload(d_s, scalars, sbytes, ctx->stream);
        int rc = H2_OK;
        if (max_bits == 0)
            msm_identity(out_xyz);
        else
            rc = msm_host_resident_scalars(ctx, d_s, bases, n, max_bits, out_xyz);
        if (rc != H2_OK) return rc;
        Fr d = fr_from_u64x4(divisor);
        Fr post3[3] = {d, d, d};
        rc = dev_ntt_impl(ctx, d_s, d_s, d_t, (uint32_t)n, omega_inv, log_n, nullptr, post3, ctx->stream, true);
        if (rc != H2_OK) return rc;
        host_download(scalars, d_s, sbytes, ctx->stream);
        H2_HIP(hipStreamSynchronize(ctx->stream));
        return (int)H2_OK;
    });
}

// ------------------------------------------------------------------ device-resident entry points
int h2_dev_ntt(void* d_a, void* d_tmp, const uint64_t omega[4], uint32_t log_n, void* stream) {
    if (!d_a || !omega) return bad("h2_dev_ntt: null argument");
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        return dev_ntt_impl(ctx, (Fr*)d_a, (Fr*)d_a, (Fr*)d_tmp, 1u << log_n, omega, log_n, nullptr, nullptr,
                            pick_stream(ctx, stream), false);
    });
}

int h2_dev_intt(void* d_a, void* d_tmp, const uint64_t omega_inv[4], const uint64_t divisor[4], uint32_t log_n,
                void* stream) {
    if (!d_a || !omega_inv || !divisor) return bad("h2_dev_intt: null argument");
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        Fr d = fr_from_u64x4(divisor);
        Fr post3[3] = {d, d, d};
        return dev_ntt_impl(ctx, (Fr*)d_a, (Fr*)d_a, (Fr*)d_tmp, 1u << log_n, omega_inv, log_n, nullptr, post3,
                            pick_stream(ctx, stream), false);
    });
}

int h2_dev_coeff_to_extended(const void* d_coeffs, void* d_out, void* d_tmp, uint32_t k, uint32_t extended_k,
                             const uint64_t g_coset[4], const uint64_t g_coset_inv[4],
                             const uint64_t extended_omega[4], void* stream) {
    if (!d_coeffs || !d_out || !g_coset || !g_coset_inv || !extended_omega)
        return bad("h2_dev_coeff_to_extended: null argument");
    if (extended_k < k) return bad("h2_dev_coeff_to_extended: extended_k < k");
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        Fr pre3[3] = {fr_from_u64x4(g_coset), fr_from_u64x4(g_coset), fr_from_u64x4(g_coset_inv)};
        if (d_coeffs == d_out && extended_k > 8 && d_tmp == nullptr) return bad("in-place extension needs d_tmp");
        return dev_ntt_impl(ctx, (const Fr*)d_coeffs, (Fr*)d_out, (Fr*)d_tmp, 1u << k, extended_omega, extended_k,
                            pre3, nullptr, pick_stream(ctx, stream), false);
    });
}

int h2_dev_coset_ntt(const void* d_coeffs, void* d_out, void* d_tmp, uint32_t log_n, const uint64_t g[4],
                     const uint64_t omega[4], void* stream) {
    if (!d_coeffs || !d_out || !g || !omega) return bad("h2_dev_coset_ntt: null argument");
    if (log_n > 28) return bad("log_n exceeds the 2-adicity of Fr (S = 28)");
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        hipStream_t s = pick_stream(ctx, stream);
        std::vector<uint32_t> bits;
        ntt_split(log_n, bits);
        if (bits.size() >= 2 && d_tmp == nullptr) return bad("NTT of this size needs a scratch buffer (d_tmp)");
        PlanRef pl = plan_locked(ctx, log_n, omega, s, false);
        ScaleTabRef tab = ntt_scale_table(pl.get(), fr_from_u64x4(g), nullptr, s);
        ntt_run(ctx, pl.get(), (const Fr*)d_coeffs, (Fr*)d_out, (Fr*)d_tmp, 1u << log_n, nullptr, nullptr, s, tab.get(), 1u);
        return (int)H2_OK;
    });
}

// Several vectors through one plan, up to 16 per launch: d_tmp = min(count, 16) x 2^log_n Fr, shared by the chunks.
static int dev_ntt_batch_impl(const void* const* srcs, void* const* dsts, size_t count, void* d_tmp, uint32_t log_n,
                              const uint64_t omega[4], const uint64_t* g, const uint64_t* divisor, uint32_t scale_mode,
                              void* stream, const char* what, uint32_t in_log = 0xffffffffu, const Fr* pre3 = nullptr) {
    if (count == 0) return H2_OK;
    if (!srcs || !dsts || !omega) return bad(what);
    if (log_n > 28) return bad("log_n exceeds the 2-adicity of Fr (S = 28)");
    for (size_t i = 0; i < count; i++)
        if (!srcs[i] || !dsts[i]) return bad(what);
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        hipStream_t s = pick_stream(ctx, stream);
        std::vector<uint32_t> bits;
        ntt_split(log_n, bits);
        if (bits.size() >= 2 && d_tmp == nullptr) return bad("NTT of this size needs a scratch buffer (d_tmp)");
        PlanRef pl = plan_locked(ctx, log_n, omega, s, false);
        ScaleTabRef tab_ref;
        Fr d{};
        Fr post3[3];
        const Fr* post = nullptr;
        if (divisor) d = fr_from_u64x4(divisor);
        if (g) {
            tab_ref = ntt_scale_table(pl.get(), fr_from_u64x4(g), divisor ? &d : nullptr, s);
        } else if (divisor) {
            post3[0] = post3[1] = post3[2] = d;
            post = post3;
        }
        const size_t n = (size_t)1 << log_n;
        std::vector<const Fr*> in(count);
        std::vector<Fr*> out(count), tmp(count);
        for (size_t i = 0; i < count; i++) {
            in[i] = (const Fr*)srcs[i];
            out[i] = (Fr*)dsts[i];
            tmp[i] = d_tmp ? (Fr*)d_tmp + (i % 16) * n : nullptr;
        }
        const uint32_t in_len = in_log == 0xffffffffu ? (uint32_t)n : (1u << in_log);   // shorter: zero-extended (coeff_to_extended)
        const Fr* tab = tab_ref.get();
        ntt_run_many(ctx, pl.get(), in.data(), out.data(), tmp.data(), count, in_len, pre3, post, s, tab,
                     tab ? scale_mode : 0u);
        return (int)H2_OK;
    });
}

int h2_dev_ntt_batch(void* const* d_a, size_t count, void* d_tmp, const uint64_t omega[4], uint32_t log_n, void* stream) {
    return dev_ntt_batch_impl((const void* const*)d_a, d_a, count, d_tmp, log_n, omega, nullptr, nullptr, 0, stream,
                              "h2_dev_ntt_batch: null argument");
}

int h2_dev_intt_batch(void* const* d_a, size_t count, void* d_tmp, const uint64_t omega_inv[4], const uint64_t divisor[4],
                      uint32_t log_n, void* stream) {
    if (!divisor) return bad("h2_dev_intt_batch: null argument");
    return dev_ntt_batch_impl((const void* const*)d_a, d_a, count, d_tmp, log_n, omega_inv, nullptr, divisor, 0, stream,
                              "h2_dev_intt_batch: null argument");
}

int h2_dev_coeff_to_extended_batch(const void* const* d_coeffs, void* const* d_out, size_t count, void* d_tmp, uint32_t k,
                                   uint32_t extended_k, const uint64_t g_coset[4], const uint64_t g_coset_inv[4],
                                   const uint64_t extended_omega[4], void* stream) {
    if (!g_coset || !g_coset_inv) return bad("h2_dev_coeff_to_extended_batch: null argument");
    if (extended_k < k) return bad("h2_dev_coeff_to_extended_batch: extended_k < k");
    const Fr pre3[3] = {fr_from_u64x4(g_coset), fr_from_u64x4(g_coset), fr_from_u64x4(g_coset_inv)};
    return dev_ntt_batch_impl(d_coeffs, d_out, count, d_tmp, extended_k, extended_omega, nullptr, nullptr, 0, stream,
                              "h2_dev_coeff_to_extended_batch: null argument", k, pre3);
}

int h2_dev_coset_ntt_batch(const void* const* d_coeffs, void* const* d_out, size_t count, void* d_tmp, uint32_t log_n,
                           const uint64_t g[4], const uint64_t omega[4], void* stream) {
    if (!g) return bad("h2_dev_coset_ntt_batch: null argument");
    return dev_ntt_batch_impl(d_coeffs, d_out, count, d_tmp, log_n, omega, g, nullptr, 1u, stream,
                              "h2_dev_coset_ntt_batch: null argument");
}

int h2_dev_coset_intt(void* d_a, void* d_tmp, uint32_t log_n, const uint64_t g_inv[4], const uint64_t omega_inv[4],
                      const uint64_t divisor[4], void* stream) {
    if (!d_a || !g_inv || !omega_inv || !divisor) return bad("h2_dev_coset_intt: null argument");
    if (log_n > 28) return bad("log_n exceeds the 2-adicity of Fr (S = 28)");
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        hipStream_t s = pick_stream(ctx, stream);
        std::vector<uint32_t> bits;
        ntt_split(log_n, bits);
        if (bits.size() >= 2 && d_tmp == nullptr) return bad("NTT of this size needs a scratch buffer (d_tmp)");
        PlanRef pl = plan_locked(ctx, log_n, omega_inv, s, false);
        const Fr d = fr_from_u64x4(divisor);
        ScaleTabRef tab = ntt_scale_table(pl.get(), fr_from_u64x4(g_inv), &d, s);
        ntt_run(ctx, pl.get(), (const Fr*)d_a, (Fr*)d_a, (Fr*)d_tmp, 1u << log_n, nullptr, nullptr, s, tab.get(), 2u);
        return (int)H2_OK;
    });
}

int h2_dev_extended_to_coeff(void* d_a, void* d_tmp, uint32_t extended_k, const uint64_t g_coset[4],
                             const uint64_t g_coset_inv[4], const uint64_t extended_omega_inv[4],
                             const uint64_t extended_ifft_divisor[4], void* stream) {
    if (!d_a || !g_coset || !g_coset_inv || !extended_omega_inv || !extended_ifft_divisor)
        return bad("h2_dev_extended_to_coeff: null argument");
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        return dev_extended_to_coeff_impl(ctx, (Fr*)d_a, (Fr*)d_tmp, extended_k, g_coset, g_coset_inv,
                                          extended_omega_inv, extended_ifft_divisor, pick_stream(ctx, stream), false);
    });
}

size_t h2_msm_scratch_bytes(size_t n, uint32_t max_bits) { return msm_scratch_bytes(n, max_bits); }
size_t h2_msm_batch_scratch_bytes(size_t n, uint32_t max_bits, size_t count) { return msm_batch_scratch_bytes(n, max_bits, count); }
int h2_msm_shape(size_t n, uint32_t max_bits, uint32_t* c, uint32_t* windows, uint32_t* buckets_per_window) {
    msm_shape_query(n, max_bits, c, windows, buckets_per_window);
    return H2_OK;
}

int h2_dev_msm(const void* d_scalars, const void* d_bases, size_t n, uint32_t max_bits, void* d_scratch,
               size_t scratch_bytes, uint64_t out_xyz[12], void* stream) {
    if (!out_xyz || (n && (!d_scalars || !d_bases))) return bad("h2_dev_msm: null argument");
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        return msm_device(ctx, (const Fr*)d_scalars, (const uint64_t*)d_bases, n, max_bits, d_scratch, scratch_bytes,
                          out_xyz, pick_stream(ctx, stream));
    });
}

int h2_dev_msm_batch(const void* const* d_scalars, size_t count, const void* d_bases, size_t n, uint32_t max_bits,
                     void* d_scratch, size_t scratch_bytes, uint64_t* out_xyz, void* stream) {
    if (count && (!d_scalars || !out_xyz || (n && !d_bases))) return bad("h2_dev_msm_batch: null argument");
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        std::lock_guard<std::mutex> g(ctx->mu);  // uses the context's two internal streams and pinned staging
        return msm_device_batch(ctx, (const Fr* const*)d_scalars, count, (const uint64_t*)d_bases, n, max_bits, d_scratch,
                                scratch_bytes, out_xyz, pick_stream(ctx, stream));
    });
}

int h2_dev_bases_precompute(const void* d_bases, size_t n, uint32_t digits, void* stream) {
    if (n && !d_bases) return bad("h2_dev_bases_precompute: null argument");
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        return bases_precompute((const uint64_t*)d_bases, n, digits, pick_stream(ctx, stream));
    });
}
int h2_dev_bases_forget(const void* d_bases) {
    return guarded([&] { return bases_forget((const uint64_t*)d_bases); });
}
size_t h2_dev_bases_precompute_bytes(size_t n, uint32_t digits) { return bases_precompute_bytes(n, digits); }

int h2_dev_random_points(uint64_t seed, size_t n, void* d_out, void* stream) {
    if (!d_out && n) return bad("h2_dev_random_points: null argument");
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        return random_points_launch(seed, n, (uint64_t*)d_out, pick_stream(ctx, stream));
    });
}

int h2_dev_msm_batch_ex(const void* const* d_scalars, const void* const* d_bases_each, const uint32_t* max_bits_each,
                        size_t count, size_t n, void* d_scratch, size_t scratch_bytes, uint64_t* out_xyz, void* stream) {
    if (count && (!d_scalars || !d_bases_each || !max_bits_each || !out_xyz)) return bad("h2_dev_msm_batch_ex: null argument");
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        std::lock_guard<std::mutex> g(ctx->mu);  // uses the context's two internal streams and pinned staging
        return msm_device_batch_ex(ctx, (const Fr* const*)d_scalars, (const uint64_t* const*)d_bases_each, max_bits_each,
                                   count, nullptr, n, 0, d_scratch, scratch_bytes, out_xyz, pick_stream(ctx, stream));
    });
}

int h2_dev_fixed_base_mul(const void* d_scalars, const void* d_table, size_t n, void* d_points, void* stream) {
    if (n && (!d_scalars || !d_table || !d_points)) return bad("h2_dev_fixed_base_mul: null argument");
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        return fixed_base_mul_launch((const Fr*)d_scalars, (const uint64_t*)d_table, n, (uint64_t*)d_points,
                                     pick_stream(ctx, stream));
    });
}

int h2_dev_points_decompress(const void* d_bytes, size_t n, void* d_points, void* stream) {
    if (n && (!d_bytes || !d_points)) return bad("h2_dev_points_decompress: null argument");
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        std::lock_guard<std::mutex> g(ctx->mu);
        uint32_t* d_bad = (uint32_t*)ctx->buf_d.get(256);
        return points_decompress_launch(d_bytes, n, (uint64_t*)d_points, d_bad, pick_stream(ctx, stream));
    });
}

int h2_dev_points_compress(const void* d_points, size_t n, void* d_bytes, void* stream) {
    if (n && (!d_bytes || !d_points)) return bad("h2_dev_points_compress: null argument");
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        return points_compress_launch((const uint64_t*)d_points, n, d_bytes, pick_stream(ctx, stream));
    });
}

int h2_dev_random_fr(const uint8_t key[32], size_t n, void* d_out, void* stream) {
    if ((!d_out && n) || !key) return bad("h2_dev_random_fr: null argument");
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        return random_fr_launch(key, n, (uint64_t*)d_out, pick_stream(ctx, stream));
    });
}

// the same into a host vector (the reference fills its `random_poly` on the host, plonk/vanishing/prover.rs:47-61: a host that
// wants the device's keyed stream in a `Vec` gets it with one copy down)
int h2_random_fr(const uint8_t key[32], size_t n, uint64_t* out) {
    if ((!out && n) || !key) return bad("h2_random_fr: null argument");
    return guarded([&] {
        if (n == 0) return (int)H2_OK;
        DeviceLease lease;
        DeviceCtx* ctx = lease.ctx;
        Prefault pf(out, n * sizeof(Fr));
        uint64_t* d = (uint64_t*)ctx->buf_a.get(n * sizeof(Fr));
        int rc = random_fr_launch(key, n, d, ctx->stream);
        if (rc != H2_OK) return rc;
        pf.join();
        host_download(out, d, n * sizeof(Fr), ctx->stream);
        H2_HIP(hipStreamSynchronize(ctx->stream));
        return (int)H2_OK;
    });
}

int h2_dev_eval_op(int op, void* d_res, const void* d_l, const void* d_r, int32_t l_rot, int32_t r_rot, size_t size,
                   const uint64_t c[4], void* stream) {
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        return eval_op_launch(op, (Fr*)d_res, (const Fr*)d_l, (const Fr*)d_r, l_rot, r_rot, size, c,
                              pick_stream(ctx, stream));
    });
}

int h2_dev_divide_by_vanishing_poly(void* d_a, size_t size, const void* d_t, size_t t_len, void* stream) {
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        return divide_by_vanishing_launch((Fr*)d_a, size, (const Fr*)d_t, t_len, pick_stream(ctx, stream));
    });
}

int h2_dev_batch_mont(void* d_a, size_t n, void* stream) {
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        return batch_mont_launch((Fr*)d_a, n, true, pick_stream(ctx, stream));
    });
}
int h2_dev_max_scalar_bits(const void* const* d_cols, size_t count, size_t n, void* d_words, uint32_t* out_bits,
                           void* stream) {
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        if (count && (!d_cols || !d_words || !out_bits)) {
            set_last_error("h2_dev_max_scalar_bits: null argument");
            return (int)H2_ERR_INVALID;
        }
        return max_scalar_bits_launch((const Fr* const*)d_cols, count, n, (uint32_t*)d_words, out_bits,
                                      pick_stream(ctx, stream));
    });
}
int h2_dev_batch_unmont(void* d_a, size_t n, void* stream) {
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        return batch_mont_launch((Fr*)d_a, n, false, pick_stream(ctx, stream));
    });
}
int h2_dev_widen_u64(const void* d_src, size_t n, void* d_dst, void* stream) {
    if (n && (!d_src || !d_dst)) return bad("h2_dev_widen_u64: null argument");
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        return widen_u64_launch((const uint64_t*)d_src, n, (Fr*)d_dst, pick_stream(ctx, stream));
    });
}

// ------------------------------------------------------------------ adjacent numerics (a24)
int h2_dev_eval_polynomial(const void* d_poly, size_t n, const uint64_t point[4], uint64_t out[4], void* stream) {
    if (!point || !out || (n && !d_poly)) return bad("h2_dev_eval_polynomial: null argument");
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        std::lock_guard<std::mutex> g(ctx->mu);
        Fr* tmp = (Fr*)ctx->buf_d.get(eval_polynomial_tmp_elems(n) * sizeof(Fr));
        return eval_polynomial_launch((const Fr*)d_poly, n, point, tmp, out, pick_stream(ctx, stream));
    });
}

int h2_dev_eval_polynomial_batch(const void* const* d_polys, size_t count, size_t n, const uint64_t* points, uint64_t* out,
                                 void* stream) {
    if (count && (!d_polys || !points || !out)) return bad("h2_dev_eval_polynomial_batch: null argument");
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        std::lock_guard<std::mutex> g(ctx->mu);
        Fr* tmp = (Fr*)ctx->buf_d.get(eval_polynomial_batch_tmp_bytes(count, n));
        return eval_polynomial_batch_launch((const Fr* const*)d_polys, count, n, points, tmp, out, pick_stream(ctx, stream));
    });
}

int h2_eval_polynomial(const uint64_t* poly, size_t n, const uint64_t point[4], uint64_t out[4]) {
    if (!point || !out || (n && !poly)) return bad("h2_eval_polynomial: null argument");
    return guarded([&] {
        DeviceLease lease;
        DeviceCtx* ctx = lease.ctx;
        Fr* tmp = (Fr*)ctx->buf_d.get(eval_polynomial_tmp_elems(n) * sizeof(Fr));
        const Fr* d = n ? resident_operand(ctx, poly, n) : nullptr;
        if (!d) {
            Fr* up = (Fr*)ctx->buf_a.get((n ? n : 1) * sizeof(Fr));
            if (n) host_upload(up, poly, n * sizeof(Fr), ctx->stream);
            d = up;
        }
        return eval_polynomial_launch(d, n, point, tmp, out, ctx->stream);
    });
}

// `count` evaluations, polynomial j at point j, in one call (plonk/prover.rs:700-790 evaluates every committed polynomial at x,
// omega x, ... in a rayon par_iter over eval_polynomial_st): every level of the folds is ONE launch over all of them and the
// values come back in one copy, instead of `count` latency-bound calls.  A polynomial inside a registered range is read on
// the device; one that is not goes up once however many points it is evaluated at.
int h2_eval_polynomial_batch(const uint64_t* const* polys, size_t count, size_t n, const uint64_t* points, uint64_t* out) {
    if (count && (!polys || !points || !out)) return bad("h2_eval_polynomial_batch: null argument");
    for (size_t j = 0; n && j < count; j++)
        if (!polys[j]) return bad("h2_eval_polynomial_batch: null polynomial");
    return guarded([&] {
        if (count == 0) return (int)H2_OK;
        if (n == 0) {
            memset(out, 0, 32 * count);
            return (int)H2_OK;
        }
        DeviceLease lease;
        DeviceCtx* ctx = lease.ctx;
        std::vector<const Fr*> d(count);
        std::map<const uint64_t*, size_t> staged;        // host vector -> its slot in the upload block
        for (size_t j = 0; j < count; j++) {
            d[j] = resident_operand(ctx, polys[j], n);
            if (!d[j] && !staged.count(polys[j])) {
                const size_t slot = staged.size();
                staged[polys[j]] = slot;
            }
        }
        if (!staged.empty()) {
            Fr* up = (Fr*)ctx->buf_a.get(staged.size() * n * sizeof(Fr));
            for (auto& kv : staged) host_upload(up + kv.second * n, kv.first, n * sizeof(Fr), ctx->stream);
            for (size_t j = 0; j < count; j++)
                if (!d[j]) d[j] = up + staged[polys[j]] * n;
        }
        Fr* tmp = (Fr*)ctx->buf_d.get(eval_polynomial_batch_tmp_bytes(count, n));
        return eval_polynomial_batch_launch(d.data(), count, n, points, tmp, out, ctx->stream);
    });
}

int h2_dev_batch_invert(void* d_a, void* d_tmp, size_t n, void* stream) {
    if (n && (!d_a || !d_tmp)) return bad("h2_dev_batch_invert: null argument");
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        return batch_invert_launch((Fr*)d_a, (Fr*)d_tmp, n, pick_stream(ctx, stream));
    });
}

int h2_batch_invert(uint64_t* a, size_t n) {
    if (n && !a) return bad("h2_batch_invert: null argument");
    return guarded([&] {
        if (n == 0) return (int)H2_OK;
        DeviceLease lease;
        DeviceCtx* ctx = lease.ctx;
        Fr* d = (Fr*)ctx->buf_a.get(n * sizeof(Fr));
        Fr* t = (Fr*)ctx->buf_b.get(n * sizeof(Fr));
        host_upload(d, a, n * sizeof(Fr), ctx->stream);
        int rc = batch_invert_launch(d, t, n, ctx->stream);
        if (rc != H2_OK) return rc;
        host_download(a, d, n * sizeof(Fr), ctx->stream);
        H2_HIP(hipStreamSynchronize(ctx->stream));
        return (int)H2_OK;
    });
}

int h2_dev_kate_division(const void* d_a, size_t n, const uint64_t b[4], void* d_q, void* stream) {
    if (!b || (n >= 2 && (!d_a || !d_q))) return bad("h2_dev_kate_division: null argument");
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        std::lock_guard<std::mutex> g(ctx->mu);
        Fr* tmp = (Fr*)ctx->buf_d.get(scan_tmp_elems(n) * sizeof(Fr));
        return kate_division_launch((const Fr*)d_a, n, b, (Fr*)d_q, tmp, pick_stream(ctx, stream));
    });
}

int h2_kate_division(const uint64_t* a, size_t n, const uint64_t b[4], uint64_t* q) {
    if (!b || (n >= 2 && (!a || !q))) return bad("h2_kate_division: null argument");
    return guarded([&] {
        if (n < 2) return (int)H2_OK;
        DeviceLease lease;
        DeviceCtx* ctx = lease.ctx;
        Prefault pf(q == a ? nullptr : q, (n - 1) * sizeof(Fr));
        Fr* d_q = (Fr*)ctx->buf_b.get(n * sizeof(Fr));
        Fr* tmp = (Fr*)ctx->buf_d.get(scan_tmp_elems(n) * sizeof(Fr));
        const Fr* d_a = resident_operand(ctx, a, n);
        if (!d_a) {
            Fr* up = (Fr*)ctx->buf_a.get(n * sizeof(Fr));
            host_upload(up, a, n * sizeof(Fr), ctx->stream);
            d_a = up;
        }
        int rc = kate_division_launch(d_a, n, b, d_q, tmp, ctx->stream);
        if (rc != H2_OK) return rc;
        pf.join();
        host_download(q, d_q, (n - 1) * sizeof(Fr), ctx->stream);
        H2_HIP(hipStreamSynchronize(ctx->stream));
        return (int)H2_OK;
    });
}

int h2_dev_prefix_product(const void* d_f, size_t n, const uint64_t init[4], void* d_z, void* stream) {
    if (!init || (n && !d_z) || (n > 1 && !d_f)) return bad("h2_dev_prefix_product: null argument");
    if (d_f == d_z && n > 1) return bad("h2_dev_prefix_product: in-place is not supported");
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        std::lock_guard<std::mutex> g(ctx->mu);
        Fr* tmp = (Fr*)ctx->buf_d.get(scan_tmp_elems(n) * sizeof(Fr));
        return prefix_product_launch((const Fr*)d_f, n, init, (Fr*)d_z, tmp, pick_stream(ctx, stream));
    });
}

int h2_prefix_product(const uint64_t* f, size_t n, const uint64_t init[4], uint64_t* z) {
    if (!init || (n && !z) || (n > 1 && !f)) return bad("h2_prefix_product: null argument");
    return guarded([&] {
        if (n == 0) return (int)H2_OK;
        DeviceLease lease;
        DeviceCtx* ctx = lease.ctx;
        Prefault pf(z, n * sizeof(Fr));
        Fr* d_f = (Fr*)ctx->buf_a.get(n * sizeof(Fr));
        Fr* d_z = (Fr*)ctx->buf_b.get(n * sizeof(Fr));
        Fr* tmp = (Fr*)ctx->buf_d.get(scan_tmp_elems(n) * sizeof(Fr));
        if (n > 1) host_upload(d_f, f, (n - 1) * sizeof(Fr), ctx->stream);
        int rc = prefix_product_launch(d_f, n, init, d_z, tmp, ctx->stream);
        if (rc != H2_OK) return rc;
        pf.join();
        host_download(z, d_z, n * sizeof(Fr), ctx->stream);
        H2_HIP(hipStreamSynchronize(ctx->stream));
        return (int)H2_OK;
    });
}

int h2_dev_lincomb(void* d_res, const void* const* d_polys, const uint64_t* coeffs, size_t count, size_t size,
                   void* stream) {
    if (!d_res || (count && (!d_polys || !coeffs))) return bad("h2_dev_lincomb: null argument");
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        return lincomb_launch((Fr*)d_res, (const Fr* const*)d_polys, coeffs, count, size, pick_stream(ctx, stream));
    });
}

// host buffers: every operand crosses PCIe once (count x size x 32 B in, size x 32 B out)
int h2_lincomb(uint64_t* res, const uint64_t* const* polys, const uint64_t* coeffs, size_t count, size_t size) {
    if (!res || (count && (!polys || !coeffs))) return bad("h2_lincomb: null argument");
    for (size_t i = 0; i < count; i++)
        if (!polys[i]) return bad("h2_lincomb: null operand");
    return guarded([&] {
        DeviceLease lease;
        DeviceCtx* ctx = lease.ctx;
        const size_t bytes = size * sizeof(Fr);
        if (count == 0 || size == 0) {
            memset(res, 0, bytes);
            return (int)H2_OK;
        }
        // operands inside a range registered with h2_poly_register are read where they lie on the device; the others cross PCIe
        std::vector<const Fr*> ptrs(count);
        std::vector<size_t> staged;                     // operands that have to be uploaded
        for (size_t i = 0; i < count; i++) {
            ptrs[i] = resident_operand(ctx, polys[i], size);
            if (!ptrs[i]) staged.push_back(i);
        }
        bool pinned_all = host_pinned(res);
        for (size_t j : staged) pinned_all = pinned_all && host_pinned(polys[j]);
        if (pinned_all && use_pipeline(size, {})) {
            // chunk by chunk: the operands' chunk c + 1 goes up while chunk c is combined and the result of chunk c - 1 comes down
            Fr* sin = staged.empty() ? nullptr : (Fr*)ctx->buf_a.get(2 * staged.size() * PIPE_CHUNK * sizeof(Fr));
            Fr* sout = (Fr*)ctx->buf_b.get(2 * PIPE_CHUNK * sizeof(Fr));
            int rc = H2_OK;
            pipeline_chunks(ctx, size,
                [&](size_t off, size_t len, int slot, hipStream_t st) {
                    for (size_t j = 0; j < staged.size(); j++)
                        H2_HIP(hipMemcpyAsync(sin + (slot * staged.size() + j) * PIPE_CHUNK, polys[staged[j]] + 4 * off, len * sizeof(Fr),
                                              hipMemcpyHostToDevice, st));
                },
                [&](size_t off, size_t len, int slot, hipStream_t st) {
                    std::vector<const Fr*> at(count);
                    for (size_t i = 0; i < count; i++) at[i] = ptrs[i] ? ptrs[i] + off : nullptr;
                    for (size_t j = 0; j < staged.size(); j++) at[staged[j]] = sin + (slot * staged.size() + j) * PIPE_CHUNK;
                    int r = lincomb_launch(sout + slot * PIPE_CHUNK, at.data(), coeffs, count, len, st);
                    if (r != H2_OK) rc = r;
                },
                [&](size_t off, size_t len, int slot, hipStream_t st) {
                    host_download(res + 4 * off, sout + slot * PIPE_CHUNK, len * sizeof(Fr), st);
                });
            return rc;
        }
        Prefault pf(count && (const void*)res == (const void*)polys[0] ? nullptr : res, bytes);
        Fr* d_all = staged.empty() ? nullptr : (Fr*)ctx->buf_a.get(bytes * staged.size());
        Fr* d_res = (Fr*)ctx->buf_b.get(bytes);
        for (size_t j = 0; j < staged.size(); j++) {
            ptrs[staged[j]] = d_all + j * size;
            host_upload(d_all + j * size, polys[staged[j]], bytes, ctx->stream);
        }
        int rc = lincomb_launch(d_res, ptrs.data(), coeffs, count, size, ctx->stream);
        if (rc != H2_OK) return rc;
        pf.join();
        host_download(res, d_res, bytes, ctx->stream);
        H2_HIP(hipStreamSynchronize(ctx->stream));
        return (int)H2_OK;
    });
}

// The quotient contributions of a multi-point opening in ONE call (poly/multiopen/shplonk/prover.rs:95-153: every rotation set's
// linear combination, minus its low-degree remainder polynomial, divided by the set's vanishing polynomial, the sets folded by
// powers of v; and :205-219 with n_sets = 1: the final quotient l(X) / (X - u)):
//   out = sum_s  (sum_i coeffs[s][i] * polys[s][i](X)  -  low_s(X)) / prod_j (X - points[s][j])
// The caller multiplies v^(R-1-s) into set s's coeffs and low (division is linear: the same field elements).  Operands inside
// a range registered with h2_poly_register are read where they lie on the device, the others go up once; every combination,
// subtraction and synthetic division stays on the device; `out` (n coefficients) comes down once.  `remainders` (optional):
// what each division left, in order -- the value of its dividend at the point (the reference's must_be_zero check, :213-214).
int h2_quotient_sum(uint64_t* out, size_t n, size_t n_sets, const size_t* counts, const uint64_t* const* polys, const uint64_t* coeffs,
                    const size_t* low_counts, const uint64_t* low, const size_t* point_counts, const uint64_t* points,
                    uint64_t* remainders) {
    if (!out || (n_sets && (!counts || !polys || !coeffs || !low_counts || !point_counts))) return bad("h2_quotient_sum: null argument");
    size_t total = 0, total_low = 0, total_points = 0;
    for (size_t s = 0; s < n_sets; s++) {
        total += counts[s];
        total_low += low_counts[s];
        total_points += point_counts[s];
        if (low_counts[s] > n) return bad("h2_quotient_sum: more low coefficients than coefficients");
    }
    if ((total_low && !low) || (total_points && !points)) return bad("h2_quotient_sum: null argument");
    for (size_t i = 0; i < total; i++)
        if (!polys[i]) return bad("h2_quotient_sum: null operand");
    return guarded([&] {
        const size_t bytes = n * sizeof(Fr);
        if (n == 0) return (int)H2_OK;
        if (n_sets == 0) {
            memset(out, 0, bytes);
            return (int)H2_OK;
        }
        DeviceLease lease;
        DeviceCtx* ctx = lease.ctx;
        Prefault pf(out, bytes);
        size_t max_staged = 0, max_low = 0;
        {
            size_t at = 0;
            for (size_t s = 0; s < n_sets; s++) {
                size_t st = 0;
                for (size_t i = 0; i < counts[s]; i++)
                    if (!resident_operand(ctx, polys[at + i], n)) st++;
                max_staged = std::max(max_staged, st);
                max_low = std::max(max_low, low_counts[s]);
                at += counts[s];
            }
        }
        Fr* d_up = max_staged ? (Fr*)ctx->buf_a.get(bytes * max_staged) : nullptr;
        Fr* cur = (Fr*)ctx->buf_b.get(bytes);
        Fr* nxt = (Fr*)ctx->buf_c.get(bytes);
        // one block: the running sum, the synthetic divisions' scratch, a set's low coefficients
        const size_t tmp_elems = std::max(scan_tmp_elems(n), eval_polynomial_tmp_elems(n));
        Fr* acc = (Fr*)ctx->buf_d.get((n + tmp_elems + max_low) * sizeof(Fr));
        Fr* tmp = acc + n;
        Fr* d_low = tmp + tmp_elems;
        H2_HIP(hipMemsetAsync(acc, 0, bytes, ctx->stream));
        size_t at = 0, at_low = 0, at_pt = 0;
        for (size_t s = 0; s < n_sets; s++) {
            const size_t count = counts[s];
            std::vector<const Fr*> ptrs(count);
            size_t st = 0;
            for (size_t i = 0; i < count; i++) {
                ptrs[i] = resident_operand(ctx, polys[at + i], n);
                if (!ptrs[i]) {
                    host_upload(d_up + st * n, polys[at + i], bytes, ctx->stream);
                    ptrs[i] = d_up + st * n;
                    st++;
                }
            }
            int rc = H2_OK;
            if (count)
                rc = lincomb_launch(cur, ptrs.data(), coeffs + 4 * at, count, n, ctx->stream);
            else
                H2_HIP(hipMemsetAsync(cur, 0, bytes, ctx->stream));
            if (rc != H2_OK) return rc;
            if (low_counts[s]) {
                host_upload(d_low, low + 4 * at_low, low_counts[s] * sizeof(Fr), ctx->stream);
                rc = eval_op_launch(H2_OP_SUB, cur, cur, d_low, 0, 0, low_counts[s], nullptr, ctx->stream);
                if (rc != H2_OK) return rc;
            }
            for (size_t j = 0; j < point_counts[s]; j++) {
                const uint64_t* pt = points + 4 * (at_pt + j);
                if (remainders) {
                    rc = eval_polynomial_launch(cur, n, pt, tmp, remainders + 4 * (at_pt + j), ctx->stream);
                    if (rc != H2_OK) return rc;
                }
                if (n >= 2) {
                    rc = kate_division_launch(cur, n, pt, nxt, tmp, ctx->stream);      // n - 1 coefficients
                    if (rc != H2_OK) return rc;
                }
                H2_HIP(hipMemsetAsync(nxt + (n - 1), 0, sizeof(Fr), ctx->stream));    // (resized as shplonk/prover.rs:118)
                std::swap(cur, nxt);
            }
            rc = eval_op_launch(H2_OP_SUM, acc, acc, cur, 0, 0, n, nullptr, ctx->stream);
            if (rc != H2_OK) return rc;
            at += count;
            at_low += low_counts[s];
            at_pt += point_counts[s];
        }
        pf.join();
        host_download(out, acc, bytes, ctx->stream);
        H2_HIP(hipStreamSynchronize(ctx->stream));
        return (int)H2_OK;
    });
}

int h2_dev_permutation_sigma(void* d_out, const void* d_map_col, const void* d_map_row, size_t n,
                             const uint64_t delta[4], const uint64_t omega[4], void* stream) {
    if (n && (!d_out || !d_map_col || !d_map_row || !delta || !omega)) return bad("h2_dev_permutation_sigma: null argument");
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        return perm_sigma_launch((Fr*)d_out, (const uint32_t*)d_map_col, (const uint32_t*)d_map_row, n, delta, omega,
                                 pick_stream(ctx, stream));
    });
}

// host buffers: the reference computes these products in a rayon loop between its GPU calls (permutation/prover.rs:89-128); a
// host that wants them on the device calls this -- value / sigma (sigma: a proving-key column, found on the device when
// registered with h2_poly_register) in, num / den out (first == 0: read, multiplied into, written back), chunk-pipelined
int h2_permutation_terms(uint64_t* num, uint64_t* den, const uint64_t* value, const uint64_t* sigma, size_t n,
                         const uint64_t beta[4], const uint64_t gamma[4], const uint64_t delta_pow[4], const uint64_t omega[4], int first) {
    if (n && (!num || !den || !value || !sigma || !beta || !gamma || !delta_pow || !omega)) return bad("h2_permutation_terms: null argument");
    return guarded([&] {
        if (n == 0) return (int)H2_OK;
        DeviceLease lease;
        DeviceCtx* ctx = lease.ctx;
        const Fr* res_sigma = resident_operand(ctx, sigma, n);
        const Fr* res_value = resident_operand(ctx, value, n);
        Prefault pf_num(first ? num : nullptr, n * sizeof(Fr)), pf_den(first ? den : nullptr, n * sizeof(Fr));
        const bool pipe = use_pipeline(n, {num, den, res_value ? nullptr : (const void*)value, res_sigma ? nullptr : (const void*)sigma});
        const size_t chunk = pipe ? PIPE_CHUNK : n;
        Fr* s_num = (Fr*)ctx->buf_a.get(2 * chunk * sizeof(Fr));
        Fr* s_den = (Fr*)ctx->buf_b.get(2 * chunk * sizeof(Fr));
        Fr* s_val = res_value ? nullptr : (Fr*)ctx->buf_c.get(2 * chunk * sizeof(Fr));
        Fr* s_sig = res_sigma ? nullptr : (Fr*)ctx->buf_d.get(2 * chunk * sizeof(Fr));
        int rc = H2_OK;
        auto up = [&](size_t off, size_t len, int slot, hipStream_t st) {
            if (s_val) host_upload(s_val + slot * chunk, value + 4 * off, len * sizeof(Fr), st);
            if (s_sig) host_upload(s_sig + slot * chunk, sigma + 4 * off, len * sizeof(Fr), st);
            if (!first) {
                host_upload(s_num + slot * chunk, num + 4 * off, len * sizeof(Fr), st);
                host_upload(s_den + slot * chunk, den + 4 * off, len * sizeof(Fr), st);
            }
        };
        auto run = [&](size_t off, size_t len, int slot, hipStream_t st) {
            // rows [off, off + len): the numerator's delta^c omega^i starts at delta_pow * omega^off
            uint64_t dp[4];
            fr_to_u64x4(fp_mul(fr_from_u64x4(delta_pow), fp_pow_u32(fr_from_u64x4(omega), (uint32_t)off)), dp);
            int r = perm_terms_launch(s_num + slot * chunk, s_den + slot * chunk, res_value ? res_value + off : s_val + slot * chunk,
                                      res_sigma ? res_sigma + off : s_sig + slot * chunk, len, beta, gamma, dp, omega, first, st);
            if (r != H2_OK) rc = r;
        };
        auto down = [&](size_t off, size_t len, int slot, hipStream_t st) {
            host_download(num + 4 * off, s_num + slot * chunk, len * sizeof(Fr), st);
            host_download(den + 4 * off, s_den + slot * chunk, len * sizeof(Fr), st);
        };
        if (pipe) {
            pipeline_chunks(ctx, n, up, run, down);
        } else {
            up(0, n, 0, ctx->stream);
            run(0, n, 0, ctx->stream);
            pf_num.join();
            pf_den.join();
            down(0, n, 0, ctx->stream);
            H2_HIP(hipStreamSynchronize(ctx->stream));
        }
        return rc;
    });
}

// One grand-product column of the permutation argument in ONE call (permutation/prover.rs:72-165, for one set of columns):
//   z[0] = init;   z[i + 1] = z[i] * prod_j (value_j[i] + beta * delta_pow * delta^j * omega^i + gamma)
//                                   / prod_j (value_j[i] + beta * sigma_j[i] + gamma)
// -- the per-column products, the batch inversion of the denominators, the product with the numerators and the prefix scan
// stay on the device: the columns go up once (those registered with h2_poly_register not at all), z comes down once, instead
// of num / den crossing PCIe in both directions around every step.  The caller writes its blinding rows into z and reads
// z[n - (blinding_factors + 1)] as the next set's init, as the reference does on its vectors.
int h2_permutation_product(uint64_t* z, const uint64_t* const* values, const uint64_t* const* sigmas, size_t count, size_t n,
                           const uint64_t beta[4], const uint64_t gamma[4], const uint64_t delta_pow[4], const uint64_t delta[4],
                           const uint64_t omega[4], const uint64_t init[4]) {
    if (n && (!z || !count || !values || !sigmas || !beta || !gamma || !delta_pow || !delta || !omega || !init))
        return bad("h2_permutation_product: null argument");
    for (size_t j = 0; n && j < count; j++)
        if (!values[j] || !sigmas[j]) return bad("h2_permutation_product: null column");
    return guarded([&] {
        if (n == 0) return (int)H2_OK;
        DeviceLease lease;
        DeviceCtx* ctx = lease.ctx;
        Prefault pf(z, n * sizeof(Fr));
        const size_t bytes = n * sizeof(Fr);
        Fr* num = (Fr*)ctx->buf_a.get(bytes);
        Fr* den = (Fr*)ctx->buf_b.get(bytes);
        Fr* s_val = (Fr*)ctx->buf_c.get(std::max(bytes, scan_tmp_elems(n) * sizeof(Fr)));
        Fr* s_sig = (Fr*)ctx->buf_d.get(std::max(bytes, scan_tmp_elems(n) * sizeof(Fr)));
        Fr dp = fr_from_u64x4(delta_pow);
        const Fr d = fr_from_u64x4(delta);
        for (size_t j = 0; j < count; j++) {
            const Fr* value = resident_operand(ctx, values[j], n);
            const Fr* sigma = resident_operand(ctx, sigmas[j], n);
            if (!value) {
                host_upload(s_val, values[j], bytes, ctx->stream);
                value = s_val;
            }
            if (!sigma) {
                host_upload(s_sig, sigmas[j], bytes, ctx->stream);
                sigma = s_sig;
            }
            uint64_t dpj[4];
            fr_to_u64x4(dp, dpj);
            int rc = perm_terms_launch(num, den, value, sigma, n, beta, gamma, dpj, omega, j == 0, ctx->stream);
            if (rc != H2_OK) return rc;
            dp = fp_mul(dp, d);
        }
        int rc = batch_invert_launch(den, s_val, n, ctx->stream);
        if (rc != H2_OK) return rc;
        rc = eval_op_launch(H2_OP_MUL, num, num, den, 0, 0, n, nullptr, ctx->stream);
        if (rc != H2_OK) return rc;
        rc = prefix_product_launch(num, n, init, den, s_sig, ctx->stream);      // z over the inverted denominators' block
        if (rc != H2_OK) return rc;
        pf.join();
        host_download(z, den, bytes, ctx->stream);
        H2_HIP(hipStreamSynchronize(ctx->stream));
        return (int)H2_OK;
    });
}

int h2_dev_permutation_terms(void* d_num, void* d_den, const void* d_value, const void* d_sigma, size_t n,
                             const uint64_t beta[4], const uint64_t gamma[4], const uint64_t delta_pow[4],
                             const uint64_t omega[4], int first, void* stream) {
    if (n && (!d_num || !d_den || !d_value || !d_sigma || !beta || !gamma || !delta_pow || !omega))
        return bad("h2_dev_permutation_terms: null argument");
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        return perm_terms_launch((Fr*)d_num, (Fr*)d_den, (const Fr*)d_value, (const Fr*)d_sigma, n, beta, gamma,
                                 delta_pow, omega, first, pick_stream(ctx, stream));
    });
}

int h2_dev_distribute_powers(void* d_a, size_t n, const uint64_t g[4], void* stream) {
    if ((n && !d_a) || !g) return bad("h2_dev_distribute_powers: null argument");
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        return distribute_powers_launch((Fr*)d_a, n, g, pick_stream(ctx, stream));
    });
}

int h2_dev_prefix_sum(const void* d_f, size_t n, const uint64_t init[4], void* d_z, void* stream) {
    if (!init || (n && !d_z) || (n > 1 && !d_f)) return bad("h2_dev_prefix_sum: null argument");
    if (d_f == d_z && n > 1) return bad("h2_dev_prefix_sum: in-place is not supported");
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        std::lock_guard<std::mutex> g(ctx->mu);
        Fr* tmp = (Fr*)ctx->buf_d.get(scan_tmp_elems(n) * sizeof(Fr));
        return prefix_sum_launch((const Fr*)d_f, n, init, (Fr*)d_z, tmp, pick_stream(ctx, stream));
    });
}

size_t h2_logup_scratch_bytes(size_t n) { return logup_scratch_bytes(n); }

// ---- host-vector twins of the remaining steps of a proof with lookups (the reference runs them as host loops: scan
// plonk/logup/prover.rs:353-367, multiplicities :104-180, sigma columns plonk/permutation/keygen.rs:197-238,
// distribute_powers_zeta poly/domain.rs:382-398): a host that keeps its vectors in memory needs no device pointer for any of them
int h2_prefix_sum(const uint64_t* f, size_t n, const uint64_t init[4], uint64_t* z) {
    if (!init || (n && !z) || (n > 1 && !f)) return bad("h2_prefix_sum: null argument");
    return guarded([&] {
        if (n == 0) return (int)H2_OK;
        DeviceLease lease;
        DeviceCtx* ctx = lease.ctx;
        Prefault pf((const void*)z == (const void*)f ? nullptr : z, n * sizeof(Fr));
        Fr* d_f = (Fr*)ctx->buf_a.get(n * sizeof(Fr));
        Fr* d_z = (Fr*)ctx->buf_b.get(n * sizeof(Fr));
        Fr* tmp = (Fr*)ctx->buf_d.get(scan_tmp_elems(n) * sizeof(Fr));
        if (n > 1) host_upload(d_f, f, (n - 1) * sizeof(Fr), ctx->stream);
        int rc = prefix_sum_launch(d_f, n, init, d_z, tmp, ctx->stream);
        if (rc != H2_OK) return rc;
        pf.join();
        host_download(z, d_z, n * sizeof(Fr), ctx->stream);
        H2_HIP(hipStreamSynchronize(ctx->stream));
        return (int)H2_OK;
    });
}

// One grand-sum column of a logup lookup in ONE call (plonk/logup/prover.rs:243-347, `commit_z`, for one set of inputs):
//   z[0] = init;   z[i + 1] = z[i] + sum_j 1 / (beta + inputs[j][i])  -  m[i] / (beta + table[i])
// the last term only for the set that carries the table (table != NULL: the first set, :283-290).  The reference makes a
// vector of beta + f per input, batch-inverts it, adds it in, and scans on the host; here every one of those vectors lives on the
// device: the inputs (and table, m) go up once -- registered ones not at all -- and z comes down once.
int h2_logup_grand_sum(uint64_t* z, const uint64_t* const* inputs, size_t count, const uint64_t* table, const uint64_t* m, size_t n,
                       const uint64_t beta[4], const uint64_t init[4]) {
    if (n && (!z || !beta || !init || (count && !inputs) || ((table == nullptr) != (m == nullptr))))
        return bad("h2_logup_grand_sum: null argument");
    for (size_t j = 0; n && j < count; j++)
        if (!inputs[j]) return bad("h2_logup_grand_sum: null input");
    return guarded([&] {
        if (n == 0) return (int)H2_OK;
        DeviceLease lease;
        DeviceCtx* ctx = lease.ctx;
        Prefault pf(z, n * sizeof(Fr));
        const size_t bytes = n * sizeof(Fr);
        hipStream_t s = ctx->stream;
        Fr* up = (Fr*)ctx->buf_a.get(bytes);
        Fr* term = (Fr*)ctx->buf_b.get(bytes);
        Fr* tmp = (Fr*)ctx->buf_c.get(bytes);
        Fr* acc = (Fr*)ctx->buf_d.get((2 * n + scan_tmp_elems(n)) * sizeof(Fr));
        Fr* d_z = acc + n;
        Fr* scan_tmp = d_z + n;
        H2_HIP(hipMemsetAsync(acc, 0, bytes, s));
        auto operand = [&](const uint64_t* host) -> const Fr* {
            const Fr* d = resident_operand(ctx, host, n);
            if (d) return d;
            host_upload(up, host, bytes, s);
            return up;
        };
        int rc = H2_OK;
        for (size_t j = 0; j < count; j++) {
            rc = eval_op_launch(H2_OP_SUM_C, term, operand(inputs[j]), nullptr, 0, 0, n, beta, s);     // beta + f_j
            if (rc != H2_OK) return rc;
            rc = batch_invert_launch(term, tmp, n, s);
            if (rc != H2_OK) return rc;
            rc = eval_op_launch(H2_OP_SUM, acc, acc, term, 0, 0, n, nullptr, s);
            if (rc != H2_OK) return rc;
        }
        if (table) {
            rc = eval_op_launch(H2_OP_SUM_C, term, operand(table), nullptr, 0, 0, n, beta, s);         // beta + t
            if (rc != H2_OK) return rc;
            rc = batch_invert_launch(term, tmp, n, s);
            if (rc != H2_OK) return rc;
            rc = eval_op_launch(H2_OP_MUL, term, term, operand(m), 0, 0, n, nullptr, s);               // m / (beta + t)
            if (rc != H2_OK) return rc;
            rc = eval_op_launch(H2_OP_SUB, acc, acc, term, 0, 0, n, nullptr, s);
            if (rc != H2_OK) return rc;
        }
        rc = prefix_sum_launch(acc, n, init, d_z, scan_tmp, s);
        if (rc != H2_OK) return rc;
        pf.join();
        host_download(z, d_z, bytes, s);
        H2_HIP(hipStreamSynchronize(s));
        return (int)H2_OK;
    });
}

int h2_distribute_powers(uint64_t* a, size_t n, const uint64_t g[4]) {
    if ((n && !a) || !g) return bad("h2_distribute_powers: null argument");
    return guarded([&] {
        if (n == 0) return (int)H2_OK;
        DeviceLease lease;
        DeviceCtx* ctx = lease.ctx;
        Fr* d = (Fr*)ctx->buf_a.get(n * sizeof(Fr));
        host_upload(d, a, n * sizeof(Fr), ctx->stream);
        int rc = distribute_powers_launch(d, n, g, ctx->stream);
        if (rc != H2_OK) return rc;
        host_download(a, d, n * sizeof(Fr), ctx->stream);
        H2_HIP(hipStreamSynchronize(ctx->stream));
        return (int)H2_OK;
    });
}

int h2_permutation_sigma(uint64_t* out, const uint32_t* map_col, const uint32_t* map_row, size_t n, const uint64_t delta[4],
                         const uint64_t omega[4]) {
    if (n && (!out || !map_col || !map_row || !delta || !omega)) return bad("h2_permutation_sigma: null argument");
    return guarded([&] {
        if (n == 0) return (int)H2_OK;
        DeviceLease lease;
        DeviceCtx* ctx = lease.ctx;
        Prefault pf(out, n * sizeof(Fr));
        Fr* d_out = (Fr*)ctx->buf_a.get(n * sizeof(Fr));
        uint32_t* d_col = (uint32_t*)ctx->buf_b.get(2 * n * sizeof(uint32_t));
        uint32_t* d_row = d_col + n;
        host_upload(d_col, map_col, n * sizeof(uint32_t), ctx->stream);
        host_upload(d_row, map_row, n * sizeof(uint32_t), ctx->stream);
        int rc = perm_sigma_launch(d_out, d_col, d_row, n, delta, omega, ctx->stream);
        if (rc != H2_OK) return rc;
        pf.join();
        host_download(out, d_out, n * sizeof(Fr), ctx->stream);
        H2_HIP(hipStreamSynchronize(ctx->stream));
        return (int)H2_OK;
    });
}

// table and inputs: host vectors of n elements (registered ones are read on the device); m: n elements out; *max_bits_out
// (optional): the bit length of the largest multiplicity
int h2_logup_multiplicity(const uint64_t* table, const uint64_t* const* inputs, size_t n_inputs, size_t usable_rows, size_t n,
                          uint64_t* m, uint32_t* max_bits_out) {
    if (n && (!table || !m || (n_inputs && !inputs))) return bad("h2_logup_multiplicity: null argument");
    for (size_t i = 0; n && i < n_inputs; i++)
        if (!inputs[i]) return bad("h2_logup_multiplicity: null input");
    if (usable_rows > n) return bad("h2_logup_multiplicity: usable_rows > n");
    return guarded([&] {
        if (max_bits_out) *max_bits_out = 0;
        if (n == 0) return (int)H2_OK;
        DeviceLease lease;
        DeviceCtx* ctx = lease.ctx;
        Prefault pf(m, n * sizeof(Fr));
        const size_t bytes = n * sizeof(Fr), sbytes = logup_scratch_bytes(n);
        std::vector<const Fr*> d_in(n_inputs);
        size_t staged = 0;
        const Fr* d_table = resident_operand(ctx, table, n);
        for (size_t i = 0; i < n_inputs; i++) {
            d_in[i] = resident_operand(ctx, inputs[i], n);
            if (!d_in[i]) staged++;
        }
        Fr* up = (Fr*)ctx->buf_a.get((staged + (d_table ? 0 : 1)) * bytes + 256);
        size_t at = 0;
        if (!d_table) {
            host_upload(up, table, bytes, ctx->stream);
            d_table = up;
            at = 1;
        }
        for (size_t i = 0; i < n_inputs; i++)
            if (!d_in[i]) {
                host_upload(up + at * n, inputs[i], bytes, ctx->stream);
                d_in[i] = up + at * n;
                at++;
            }
        Fr* d_m = (Fr*)ctx->buf_b.get(bytes);
        void* d_scratch = ctx->buf_c.get(sbytes);
        uint32_t max_count = 0;
        int rc = logup_multiplicity_launch(d_table, d_in.data(), n_inputs, usable_rows, n, d_m, d_scratch, sbytes, ctx->stream, &max_count);
        if (rc != H2_OK) return rc;
        if (max_bits_out) {
            uint32_t bits = 0;
            while (bits < 32 && (max_count >> bits)) bits++;
            *max_bits_out = bits;
        }
        pf.join();
        host_download(m, d_m, bytes, ctx->stream);
        H2_HIP(hipStreamSynchronize(ctx->stream));
        return (int)H2_OK;
    });
}


int h2_dev_logup_counts(const void* d_table, const void* const* d_inputs, size_t n_inputs, size_t usable_rows, size_t n,
                        size_t row_begin, size_t row_end, void* d_counts, void* d_scratch, size_t scratch_bytes, void* stream) {
    if (!d_table || !d_counts || !d_scratch || (n_inputs && !d_inputs)) return bad("h2_dev_logup_counts: null argument");
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        return logup_counts_launch((const Fr*)d_table, (const Fr* const*)d_inputs, n_inputs, usable_rows, n, row_begin, row_end,
                                   (uint32_t*)d_counts, d_scratch, scratch_bytes, pick_stream(ctx, stream));
    });
}

int h2_dev_logup_emit(const void* d_counts, size_t usable_rows, size_t n, void* d_m, void* stream) {
    if (!d_counts || !d_m) return bad("h2_dev_logup_emit: null argument");
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        return logup_emit_launch((const uint32_t*)d_counts, usable_rows, n, (Fr*)d_m, pick_stream(ctx, stream));
    });
}

int h2_dev_logup_multiplicity(const void* d_table, const void* const* d_inputs, size_t n_inputs, size_t usable_rows,
                              size_t n, void* d_m, void* d_scratch, size_t scratch_bytes, void* stream) {
    if (!d_table || !d_m || !d_scratch || (n_inputs && !d_inputs)) return bad("h2_dev_logup_multiplicity: null argument");
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        return logup_multiplicity_launch((const Fr*)d_table, (const Fr* const*)d_inputs, n_inputs, usable_rows, n, (Fr*)d_m,
                                         d_scratch, scratch_bytes, pick_stream(ctx, stream));
    });
}

int h2_dev_logup_multiplicity_bits(const void* d_table, const void* const* d_inputs, size_t n_inputs, size_t usable_rows,
                                   size_t n, void* d_m, void* d_scratch, size_t scratch_bytes, uint32_t* max_bits_out,
                                   void* stream) {
    if (!d_table || !d_m || !d_scratch || !max_bits_out || (n_inputs && !d_inputs)) return bad("h2_dev_logup_multiplicity_bits: null argument");
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        uint32_t max_count = 0;
        const int rc = logup_multiplicity_launch((const Fr*)d_table, (const Fr* const*)d_inputs, n_inputs, usable_rows, n, (Fr*)d_m,
                                                 d_scratch, scratch_bytes, pick_stream(ctx, stream), &max_count);
        uint32_t bits = 0;
        while (bits < 32 && (max_count >> bits)) bits++;
        *max_bits_out = bits;
        return rc;
    });
}

// ------------------------------------------------------------------ evaluate_h: the generated form
int h2_evalh_prepare(const h2_evalh_desc* desc, h2_evalh_info* info) {
    if (!desc) return bad("h2_evalh_prepare: null argument");
    return guarded([&] {
        current_ctx();  // the device of this thread is initialised
        int cached = 0;
        EvalhPlanRef held = evalh_plan_get(desc, &cached);
        const EvalhPlan* plan = held.get();
        if (!plan) return bad("h2_evalh_prepare: no generated kernels for this program (H2_EVALH_JIT=0, hipRTC unavailable or a rejected program: see stderr)");
        if (info) {
            evalh_plan_info(plan, info);
            info->from_cache = (uint32_t)cached;
        }
        return (int)H2_OK;
    });
}

int h2_evalh_compile(const h2_evalh_desc* desc, h2_evalh_info* info) {
    if (!desc) return bad("h2_evalh_compile: null argument");
    return guarded([&] {
        evgen::Generated g = evgen::compile(desc, evgen::Options::from_env());
        if (info) evalh_gen_info(g, info);
        return (int)H2_OK;
    });
}

int h2_evalh_source(const h2_evalh_desc* desc, uint32_t stage, char* buf, size_t cap, size_t* len) {
    if (!desc || !len || (cap && !buf)) return bad("h2_evalh_source: null argument");
    return guarded([&] {
        evgen::Generated g = evgen::generate(desc, evgen::Options::from_env());
        if (stage >= g.stages.size()) return bad("h2_evalh_source: no such stage");
        const std::string& src = g.stages[stage].source;
        *len = src.size();
        if (cap) {
            const size_t n = std::min(cap - 1, src.size());
            memcpy(buf, src.data(), n);
            buf[n] = 0;
        }
        return (int)H2_OK;
    });
}

int h2_evalh_stage_args(const h2_evalh_desc* desc, uint32_t stage, uint64_t* values, const uint64_t* tw_lo, const uint64_t* tw_hi,
                        uint64_t row_begin, uint64_t row_end, void* buf, size_t cap, size_t* len) {
    if (!desc || !len || (cap && !buf)) return bad("h2_evalh_stage_args: null argument");
    return guarded([&] {
        if (desc->extended_k < desc->k || desc->extended_k > 28) return bad("h2_evalh_stage_args: bad k / extended_k");
        evgen::Generated g = evgen::generate(desc, evgen::Options::from_env());
        if (stage >= g.stages.size()) return bad("h2_evalh_stage_args: no such stage");
        alignas(16) unsigned char tmp[4096];  // (pointers, u64 and Fr values are stored into it)
        *len = evalh_fill_stage_args(g.stages[stage], desc, (Fr*)values, (const Fr*)tw_lo, (const Fr*)tw_hi, (size_t)row_begin,
                                     (size_t)row_end, tmp, sizeof tmp);
        if (cap) {
            if (cap < *len) return bad("h2_evalh_stage_args: buffer too small");
            memcpy(buf, tmp, *len);
        }
        return (int)H2_OK;
    });
}

uint64_t h2_evalh_generated_launches(void) { return evalh_generated_launches(); }

// ------------------------------------------------------------------ evaluate_h
int h2_evaluate_h(const h2_evalh_desc* desc, uint64_t* values) {
    if (!desc || !values) return bad("h2_evaluate_h: null argument");
    return guarded([&] {
        Prefault pf(values, sizeof(Fr) << desc->extended_k);
        pf.join();   // (before anything is copied back: the touch fallback must not race with the copy)
        DeviceLease lease;
        return evalh_host(lease.ctx, desc, values);
    });
}

int h2_evaluate_h_coeff(const h2_evalh_desc* desc, uint64_t* values) {
    if (!desc || !values) return bad("h2_evaluate_h_coeff: null argument");
    return guarded([&] {
        Prefault pf(values, desc->extended_k <= 28 ? sizeof(Fr) << desc->extended_k : 0);
        pf.join();
        return evalh_host_coeffs(desc, values);
    });
}

// The vanishing argument's quotient, from coefficient forms to coefficient form, in ONE call: what the reference's cuda path does
// in three steps on host vectors of 2^extended_k elements -- Evaluator::evaluate_h (plonk/evaluation.rs:1229-1985), then
// divide_by_vanishing_poly (poly/domain.rs:354-373) and extended_to_coeff (:328-350) in vanishing::Argument::construct
// (plonk/vanishing/prover.rs:69-112).  The 2^extended_k values of the numerator stay on the device: divided there, taken back
// to coefficients there, and only the out_len = n * quotient_poly_degree coefficients of h(X) cross PCIe (k = 22, degree 4:
// 384 MiB down instead of 512 MiB down + up + down + up + 384 MiB down, and two 512 MiB host vectors that never exist).
// With several devices in the pool the cosets are dealt over them as h2_evaluate_h_coeff does and the two other steps follow
// through a host vector of the call's own.
int h2_quotient_poly_coeff(const h2_evalh_desc* desc, const uint64_t* t_evaluations, size_t t_len, const uint64_t g_coset[4],
                           const uint64_t g_coset_inv[4], const uint64_t extended_omega_inv[4],
                           const uint64_t extended_ifft_divisor[4], uint64_t* out, size_t out_len) {
    if (!desc || !t_evaluations || !t_len || !g_coset || !g_coset_inv || !extended_omega_inv || !extended_ifft_divisor || !out)
        return bad("h2_quotient_poly_coeff: null argument");
    if (desc->extended_k < desc->k || desc->extended_k > 28) return bad("h2_quotient_poly_coeff: bad k / extended_k");
    const size_t size = (size_t)1 << desc->extended_k;
    if (out_len > size) return bad("h2_quotient_poly_coeff: out_len exceeds the extended domain");
    if (size % t_len) return bad("h2_quotient_poly_coeff: t_len does not divide the extended domain");
    return guarded([&] {
        Prefault pf(out, out_len * sizeof(Fr));
        const uint32_t ek = desc->extended_k;
        EvalhFinish finish = [&](DeviceCtx* ctx, Fr* d_values, hipStream_t s) -> int {
            Fr* d_t = (Fr*)ctx->buf_c.get(t_len * sizeof(Fr));
            host_upload(d_t, t_evaluations, t_len * sizeof(Fr), s);
            int rc = divide_by_vanishing_launch(d_values, size, d_t, t_len, s);
            if (rc != H2_OK) return rc;
            Fr* d_tmp = (Fr*)ctx->buf_b.get(size * sizeof(Fr));
            rc = dev_extended_to_coeff_impl(ctx, d_values, d_tmp, ek, g_coset, g_coset_inv, extended_omega_inv, extended_ifft_divisor, s, true);
            if (rc != H2_OK) return rc;
            pf.join();
            host_download(out, d_values, out_len * sizeof(Fr), s);
            return (int)H2_OK;
        };
        if (evalh_host_workers(desc) <= 1) {
            bool finished = false;
            int rc = evalh_host_coeffs(desc, nullptr, &finish, &finished);
            if (rc == H2_OK && !finished) return bad("h2_quotient_poly_coeff: the evaluation did not hand its values over");
            return rc;
        }
        std::vector<uint64_t> values(4 * size);
        int rc = evalh_host_coeffs(desc, values.data());
        if (rc != H2_OK) return rc;
        rc = h2_divide_by_vanishing_poly(values.data(), size, t_evaluations, t_len);
        if (rc != H2_OK) return rc;
        return h2_extended_to_coeff(values.data(), out, out_len, ek, g_coset, g_coset_inv, extended_omega_inv, extended_ifft_divisor);
    });
}

int h2_dev_evaluate_h(const h2_evalh_desc* desc, void* d_values, void* stream) {
    if (!desc || !d_values) return bad("h2_dev_evaluate_h: null argument");
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        std::lock_guard<std::mutex> g(ctx->mu);  // the interpreter work space is per device
        return evalh_device(ctx, desc, (Fr*)d_values, pick_stream(ctx, stream), true);
    });
}

// ------------------------------------------------------------------ HIP-event timer for bench.py
namespace {
std::mutex g_timer_mu;
std::map<void*, std::pair<hipEvent_t, hipEvent_t>> g_timers;
}  // namespace

int h2_timer_start(void* stream) {
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        hipStream_t s = pick_stream(ctx, stream);
        std::lock_guard<std::mutex> g(g_timer_mu);
        auto& t = g_timers[(void*)s];
        if (!t.first) {
            H2_HIP(hipEventCreate(&t.first));
            H2_HIP(hipEventCreate(&t.second));
        }
        H2_HIP(hipEventRecord(t.first, s));
        return (int)H2_OK;
    });
}

int h2_timer_stop(void* stream, float* ms_out) {
    if (!ms_out) return bad("h2_timer_stop: null argument");
    return guarded([&] {
        DeviceCtx* ctx = current_ctx();
        hipStream_t s = pick_stream(ctx, stream);
        std::lock_guard<std::mutex> g(g_timer_mu);
        auto it = g_timers.find((void*)s);
        if (it == g_timers.end()) return bad("h2_timer_stop without h2_timer_start");
        H2_HIP(hipEventRecord(it->second.second, s));
        H2_HIP(hipEventSynchronize(it->second.second));
        H2_HIP(hipEventElapsedTime(ms_out, it->second.first, it->second.second));
        return (int)H2_OK;
    });
}

}  // extern "C"
