// How fast do host threads READ a buffer the GPU just filled by DMA? hipHostMalloc (default / non-coherent / NUMA-user) vs malloc + hipHostRegister
// vs plain malloc (no DMA). Output of the ingest pipeline (rendered tuples) is read this way: profiles/r05_pinned_read.txt
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double copy_out(const char* src, char* dst, size_t n, int T) {
    const double t0 = now();
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back([=] { memcpy(dst + n * t / T, src + n * t / T, n * (t + 1) / T - n * t / T); });
    for (auto& x : th) x.join();
    return now() - t0;
}
int main() {
    const size_t n = 128u << 20;
    const int T = 16;
    void* d = nullptr;
    hipMalloc(&d, n);
    hipMemset(d, 7, n);
    char* dst = (char*)malloc(n);
    memset(dst, 1, n);
    struct Kind { const char* name; unsigned flags; int mode; } kinds[] = {{"hipHostMalloc default", hipHostMallocDefault, 0}, {"hipHostMalloc non-coherent", hipHostMallocNonCoherent, 0},
                                                                         {"hipHostMalloc portable|mapped", hipHostMallocPortable | hipHostMallocMapped, 0},
                                                                         {"malloc + hipHostRegister", 0, 1}, {"plain malloc (no DMA, memset)", 0, 2}};
    for (const Kind& k : kinds) {
        char* h = nullptr;
        if (k.mode == 0) { if (hipHostMalloc((void**)&h, n, k.flags) != hipSuccess) { printf("%-32s alloc failed\n", k.name); continue; } }
        else { h = (char*)aligned_alloc(4096, n); memset(h, 3, n); if (k.mode == 1 && hipHostRegister(h, n, hipHostRegisterDefault) != hipSuccess) { printf("%-32s register failed\n", k.name); continue; } }
        double best = 1e9, d2h = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            if (k.mode != 2) { const double t0 = now(); hipMemcpy(h, d, n, hipMemcpyDeviceToHost); d2h = std::min(d2h, now() - t0); }
            else memset(h, rep, n);
            best = std::min(best, copy_out(h, dst, n, T));
        }
        printf("%-32s D2H %.1f GB/s   %d threads read it at %.1f GB/s\n", k.name, k.mode != 2 ? n / d2h / 1e9 : 0.0, T, n / best / 1e9);
        if (k.mode == 0) hipHostFree(h); else { if (k.mode == 1) hipHostUnregister(h); free(h); }
    }
    return 0;
}
