// populate_bench.cpp -- what does the first touch of a fresh 600 MB result array cost on this host, by thread count,
// with and without MADV_HUGEPAGE, by MADV_POPULATE_WRITE and by plain stores?  (host path, fresh result arrays)
//   g++ -O2 -std=c++17 -pthread -o /tmp/populate_bench scripts/micro/populate_bench.cpp && /tmp/populate_bench
#include <sys/mman.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t bytes = (size_t)600 << 20;
    for (int huge = 1; huge >= 0; huge--)
        for (int mode = 0; mode < 2; mode++)
            for (int threads : {4, 8, 16, 32, 64}) {
                double best = 1e9;
                for (int rep = 0; rep < 3; rep++) {
                    char *p = (char *)mmap(nullptr, bytes + (2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
                    if (p == MAP_FAILED) return 1;
                    char *q = p + 4096 + 64;      // numpy-like: not huge-page aligned
                    if (huge) madvise((void *)(((uintptr_t)q + (2 << 20) - 1) & ~(uintptr_t)((2 << 20) - 1)), bytes - (2 << 20), MADV_HUGEPAGE);
                    const double t0 = now();
                    std::vector<std::thread> th;
                    for (int t = 0; t < threads; t++)
                        th.emplace_back([=] {
                            const size_t lo = bytes * t / threads, hi = bytes * (t + 1) / threads;
                            const uintptr_t b = ((uintptr_t)q + lo + 4095) & ~(uintptr_t)4095, e = ((uintptr_t)q + hi) & ~(uintptr_t)4095;
                            if (mode == 0) madvise((void *)b, e - b, MADV_POPULATE_WRITE);
                            else for (uintptr_t a = b; a < e; a += 4096) *(volatile char *)a = 1;
                        });
                    for (auto &t : th) t.join();
                    const double dt = now() - t0;
                    if (dt < best) best = dt;
                    munmap(p, bytes + (2 << 20));
                }
                printf("huge=%d %-22s threads=%2d  %.2f ms  %.1f GB/s\n", huge, mode == 0 ? "MADV_POPULATE_WRITE" : "one store per page", threads, best * 1e3, bytes / best / 1e9);
            }
    return 0;
}
