// What does it cost to make the pages of a FRESH 128 MiB host vector exist?  T threads x {MADV_POPULATE_WRITE, touch} x {huge pages
// advised or not}, and what munmap of the populated range costs.   g++ -O2 -o /tmp/prefault_probe prefault_probe.cpp -lpthread
#include <sys/mman.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <thread>
#include <vector>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    for (const char* f : {"/sys/kernel/mm/transparent_hugepage/enabled", "/sys/kernel/mm/transparent_hugepage/defrag"}) {
        std::ifstream in(f);
        std::string s;
        std::getline(in, s);
        printf("%s: %s\n", f, s.c_str());
    }
    const size_t bytes = (size_t)128 << 20;
    for (int huge = 0; huge < 2; huge++)
        for (int mode = 0; mode < 2; mode++)
            for (int T : {1, 4, 8, 16, 32}) {
                double best = 1e9, unmap = 1e9;
                for (int r = 0; r < 3; r++) {
                    char* p = (char*)mmap(nullptr, bytes + (2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
                    char* a = (char*)(((uintptr_t)p + (2 << 20) - 1) & ~(uintptr_t)((2 << 20) - 1));
                    const double t0 = now();
                    if (huge) madvise(a, bytes, MADV_HUGEPAGE);
                    std::vector<std::thread> th;
                    const size_t per = bytes / T;
                    for (int t = 0; t < T; t++)
                        th.emplace_back([=] {
                            char* at = a + t * per;
                            if (mode == 0 && madvise(at, per, MADV_POPULATE_WRITE) == 0) return;
                            for (size_t i = 0; i < per; i += 4096) ((volatile char*)at)[i] = 0;
                        });
                    for (auto& x : th) x.join();
                    best = std::min(best, now() - t0);
                    const double t1 = now();
                    munmap(p, bytes + (2 << 20));
                    unmap = std::min(unmap, now() - t1);
                }
                printf("huge=%d %-14s T=%2d  populate %7.2f ms   munmap %6.2f ms\n", huge, mode ? "touch" : "POPULATE_WRITE", T, best * 1e3, unmap * 1e3);
            }
    return 0;
}
