// How fast can a vector in ORDINARY host memory cross PCIe?  hipMemcpy from / to pageable memory (the runtime's own staging)
// against T host threads that each run a two-slot pipeline through their own page-locked staging blocks: memcpy into a slot,
// hipMemcpyAsync from it on the shared stream (and the reverse on the way back).  Prints GB/s per direction.
//   hipcc -O2 --offload-arch=gfx950 -o /tmp/pageable_probe tools/experiments/pageable_probe.hip -lpthread && /tmp/pageable_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                        \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Lane {
    char* slot[2] = {nullptr, nullptr};
    hipEvent_t ev[2];
};

static void staged(bool up, char* dev, char* host, size_t bytes, int T, size_t chunk, std::vector<Lane>& lanes, hipStream_t s) {
    const size_t chunks = (bytes + chunk - 1) / chunk;
    auto work = [&](int t) {
        Lane& L = lanes[t];
        if (up) {
            size_t j = 0;
            for (size_t c = t; c < chunks; c += T, j++) {
                const size_t off = c * chunk, len = std::min(chunk, bytes - off);
                const int sl = (int)(j & 1);
                if (j >= 2) CK(hipEventSynchronize(L.ev[sl]));
                memcpy(L.slot[sl], host + off, len);
                CK(hipMemcpyAsync(dev + off, L.slot[sl], len, hipMemcpyHostToDevice, s));
                CK(hipEventRecord(L.ev[sl], s));
            }
            if (j >= 1) CK(hipEventSynchronize(L.ev[(j - 1) & 1]));
            if (j >= 2) CK(hipEventSynchronize(L.ev[j & 1]));
        } else {
            std::vector<size_t> mine;
            for (size_t c = t; c < chunks; c += T) mine.push_back(c);
            auto issue = [&](size_t j) {
                const size_t off = mine[j] * chunk, len = std::min(chunk, bytes - off);
                CK(hipMemcpyAsync(L.slot[j & 1], dev + off, len, hipMemcpyDeviceToHost, s));
                CK(hipEventRecord(L.ev[j & 1], s));
            };
            if (!mine.empty()) issue(0);
            for (size_t j = 0; j < mine.size(); j++) {
                if (j + 1 < mine.size()) issue(j + 1);
                CK(hipEventSynchronize(L.ev[j & 1]));
                const size_t off = mine[j] * chunk, len = std::min(chunk, bytes - off);
                memcpy(host + off, L.slot[j & 1], len);
            }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
}

int main(int argc, char** argv) {
    const size_t bytes = (size_t)(argc > 1 ? atoi(argv[1]) : 128) << 20;
    char *dev, *pin;
    CK(hipMalloc(&dev, bytes));
    CK(hipHostMalloc(&pin, bytes, hipHostMallocPortable));
    memset(pin, 1, bytes);
    char* page = (char*)aligned_alloc(4096, bytes);
    memset(page, 2, bytes);
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto rate = [&](const char* what, auto fn) {
        fn();
        double best = 1e9;
        for (int r = 0; r < 5; r++) {
            const double t0 = now();
            fn();
            best = std::min(best, now() - t0);
        }
        printf("%-58s %8.2f ms  %6.1f GB/s\n", what, best * 1e3, bytes / best / 1e9);
        fflush(stdout);
    };
    printf("%zu MiB per transfer\n", bytes >> 20);
    rate("pinned   H2D hipMemcpyAsync + sync", [&] { CK(hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); });
    rate("pinned   D2H hipMemcpyAsync + sync", [&] { CK(hipMemcpyAsync(pin, dev, bytes, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); });
    rate("pageable H2D hipMemcpyAsync + sync", [&] { CK(hipMemcpyAsync(dev, page, bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); });
    rate("pageable D2H hipMemcpyAsync + sync", [&] { CK(hipMemcpyAsync(page, dev, bytes, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); });
    rate("host memcpy pageable -> pageable, 1 thread", [&] { memcpy(pin, page, bytes); });
    for (size_t chunk : {(size_t)2 << 20, (size_t)4 << 20, (size_t)8 << 20}) {
        for (int T : {1, 2, 4, 8, 12, 16}) {
            std::vector<Lane> lanes(T);
            for (auto& L : lanes)
                for (int i = 0; i < 2; i++) {
                    CK(hipHostMalloc(&L.slot[i], chunk, hipHostMallocPortable));
                    CK(hipEventCreateWithFlags(&L.ev[i], hipEventDisableTiming));
                }
            char what[128];
            snprintf(what, sizeof what, "staged   H2D %2d threads, %zu MiB chunks", T, chunk >> 20);
            rate(what, [&] { staged(true, dev, page, bytes, T, chunk, lanes, s); });
            snprintf(what, sizeof what, "staged   D2H %2d threads, %zu MiB chunks", T, chunk >> 20);
            rate(what, [&] { staged(false, dev, page, bytes, T, chunk, lanes, s); });
            for (auto& L : lanes)
                for (int i = 0; i < 2; i++) {
                    CK(hipHostFree(L.slot[i]));
                    CK(hipEventDestroy(L.ev[i]));
                }
        }
    }
    // a buffer whose pages exist but which the runtime has never seen (every vector of a proof is one): is there a per-buffer cost?
    for (int up = 0; up < 2; up++) {
        for (int T : {0, 4}) {
            const size_t chunk = (size_t)8 << 20;
            std::vector<Lane> lanes(T ? T : 1);
            for (auto& L : lanes)
                for (int i = 0; i < 2; i++) {
                    CK(hipHostMalloc(&L.slot[i], chunk, hipHostMallocPortable));
                    CK(hipEventCreateWithFlags(&L.ev[i], hipEventDisableTiming));
                }
            double best = 1e9, worst = 0;
            for (int r = 0; r < 4; r++) {
                char* once = (char*)aligned_alloc(4096, bytes);
                memset(once, 3, bytes);
                const double t0 = now();
                if (T)
                    staged(up, dev, once, bytes, T, chunk, lanes, s);
                else {
                    CK(up ? hipMemcpyAsync(dev, once, bytes, hipMemcpyHostToDevice, s) : hipMemcpyAsync(once, dev, bytes, hipMemcpyDeviceToHost, s));
                    CK(hipStreamSynchronize(s));
                }
                const double dt = now() - t0;
                best = std::min(best, dt);
                worst = std::max(worst, dt);
                free(once);
            }
            printf("%s %s, buffer touched but NEW to the runtime   best %6.2f ms  worst %6.2f ms\n", T ? "staged 4 threads  " : "hipMemcpyAsync    ", up ? "H2D" : "D2H",
                   best * 1e3, worst * 1e3);
        }
    }
    // a FRESH destination (pages never touched): what a new Vec is
    for (int T : {1, 8}) {
        const size_t chunk = (size_t)4 << 20;
        std::vector<Lane> lanes(T);
        for (auto& L : lanes)
            for (int i = 0; i < 2; i++) {
                CK(hipHostMalloc(&L.slot[i], chunk, hipHostMallocPortable));
                CK(hipEventCreateWithFlags(&L.ev[i], hipEventDisableTiming));
            }
        double best = 1e9;
        for (int r = 0; r < 3; r++) {
            char* fresh = (char*)aligned_alloc(4096, bytes);
            const double t0 = now();
            staged(false, dev, fresh, bytes, T, chunk, lanes, s);
            best = std::min(best, now() - t0);
            free(fresh);
        }
        printf("staged   D2H %2d threads into a FRESH allocation            %8.2f ms  %6.1f GB/s\n", T, best * 1e3, bytes / best / 1e9);
    }
    {
        double best = 1e9;
        for (int r = 0; r < 3; r++) {
            char* fresh = (char*)aligned_alloc(4096, bytes);
            const double t0 = now();
            CK(hipMemcpyAsync(fresh, dev, bytes, hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
            best = std::min(best, now() - t0);
            free(fresh);
        }
        printf("pageable D2H hipMemcpyAsync into a FRESH allocation         %8.2f ms  %6.1f GB/s\n", best * 1e3, bytes / best / 1e9);
    }
    return 0;
}
