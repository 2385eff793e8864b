// What a fresh result buffer costs on this host, by strategy: g++ -O2 -pthread microbench/pagefault.cpp -o microbench/_ab/pagefault
// Fresh anonymous mappings of 72 MB (a noise picture's container) and 128 MiB (an int64 band), every page touched once.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <sys/mman.h>
#include <thread>
#include <vector>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void touch(uint8_t *p, size_t n, int threads, size_t grain)
{
    auto work = [&](size_t a, size_t b) { for (size_t o = a; o < b; o += 4096) ((volatile uint8_t *)p)[o] = 0; };
    if (threads <= 1) { work(0, n); return; }
    std::vector<std::thread> th;
    // interleaved grains: thread t takes grains t, t + threads, ...
    for (int t = 0; t < threads; ++t)
        th.emplace_back([&, t] { for (size_t g = (size_t)t * grain; g < n; g += grain * threads) work(g, g + grain < n ? g + grain : n); });
    for (auto &x : th) x.join();
}
int main()
{
    FILE *f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
    char line[256] = "?";
    if (f) { if (!fgets(line, sizeof line, f)) line[0] = 0; fclose(f); }
    printf("transparent_hugepage/enabled: %s", line);
    printf("hardware threads: %u\n", std::thread::hardware_concurrency());
    const size_t sizes[2] = {(size_t)72 << 20, (size_t)128 << 20};
    for (size_t n : sizes) {
        struct { const char *name; int huge, populate, threads; size_t grain; } cases[] = {
            {"touch, 1 thread", 0, 0, 1, 0}, {"touch, 8 threads in 8 contiguous parts", 0, 0, 8, 0}, {"touch, 8 threads, 2 MiB grains", 0, 0, 8, (size_t)2 << 20},
            {"touch, 16 threads, 2 MiB grains", 0, 0, 16, (size_t)2 << 20},
            {"MADV_POPULATE_WRITE", 0, 1, 1, 0}, {"MADV_HUGEPAGE + touch, 1 thread", 1, 0, 1, 0}, {"MADV_HUGEPAGE + touch, 8 threads, 2 MiB grains", 1, 0, 8, (size_t)2 << 20},
            {"MADV_HUGEPAGE + touch, 16 threads, 2 MiB grains", 1, 0, 16, (size_t)2 << 20}, {"MADV_HUGEPAGE + MADV_POPULATE_WRITE", 1, 1, 1, 0},
            {"MADV_POPULATE_WRITE by 8 threads, 2 MiB grains", 0, 2, 8, (size_t)2 << 20}, {"MADV_HUGEPAGE + MADV_POPULATE_WRITE by 8 threads", 1, 2, 8, (size_t)2 << 20},
        };
        for (auto &c : cases) {
            double best = 1e9, best_unmap = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                uint8_t *raw = (uint8_t *)mmap(nullptr, n + (2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
                if (raw == MAP_FAILED) return 1;
                uint8_t *p = (uint8_t *)(((uintptr_t)raw + (2 << 20) - 1) & ~(uintptr_t)((2 << 20) - 1));
                const double t0 = now();
                if (c.huge) madvise(p, n, MADV_HUGEPAGE);
                if (c.populate == 1) { if (madvise(p, n, MADV_POPULATE_WRITE) != 0) touch(p, n, 1, 0); }
                else if (c.populate == 2) {
                    std::vector<std::thread> th;
                    for (int t = 0; t < c.threads; ++t)
                        th.emplace_back([&, t] { for (size_t g = (size_t)t * c.grain; g < n; g += c.grain * c.threads) madvise(p + g, g + c.grain < n ? c.grain : n - g, MADV_POPULATE_WRITE); });
                    for (auto &x : th) x.join();
                } else touch(p, n, c.threads, c.grain ? c.grain : (n / c.threads + 4095) & ~(size_t)4095);
                const double t1 = now();
                munmap(raw, n + (2 << 20));
                const double t2 = now();
                if (t1 - t0 < best) best = t1 - t0;
                if (t2 - t1 < best_unmap) best_unmap = t2 - t1;
            }
            printf("%4zu MiB  %-52s %6.2f ms   (munmap %5.2f ms)\n", n >> 20, c.name, best, best_unmap);
        }
    }
    return 0;
}
