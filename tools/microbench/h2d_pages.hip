// Does the page size behind a pinned buffer move the copy rate? 4 x 64 MiB of anonymous memory with MADV_HUGEPAGE / MADV_NOHUGEPAGE / as hipHostMalloc
// hands it out, registered with HIP, copied to the GPU and back in 64 MiB pieces; AnonHugePages of the mapping from /proc/self/smaps.
// build: hipcc -O2 --offload-arch=gfx950 -o h2d_pages h2d_pages.hip     (measurement only; nothing in the product uses it)
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static long huge_kb_of(const void* p) {   // AnonHugePages of the mapping that holds p
    FILE* f = fopen("/proc/self/smaps", "r");
    if (!f) return -1;
    char line[512];
    bool in = false;
    long kb = -1;
    while (fgets(line, sizeof line, f)) {
        unsigned long a, b;
        if (sscanf(line, "%lx-%lx ", &a, &b) == 2 && strchr(line, '-') && strchr(line, ' ') && (line[0] != 'A')) in = (unsigned long)p >= a && (unsigned long)p < b;
        else if (in && strncmp(line, "AnonHugePages:", 14) == 0) { kb = atol(line + 14); break; }
    }
    fclose(f);
    return kb;
}

int main() {
    const size_t PIECE = 64ull << 20;
    const int NP = 4, REPS = 24;
    {
        FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
        char buf[128] = {0};
        if (f) { if (fgets(buf, sizeof buf, f)) printf("transparent_hugepage/enabled: %s", buf); fclose(f); }
        f = fopen("/sys/kernel/mm/transparent_hugepage/defrag", "r");
        if (f) { if (fgets(buf, sizeof buf, f)) printf("transparent_hugepage/defrag: %s", buf); fclose(f); }
    }
    uint8_t* d = nullptr;
    if (hipMalloc(&d, PIECE * NP) != hipSuccess) return 1;
    hipStream_t s;
    (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int mode = 0; mode < 3; ++mode) {   // 0: MADV_HUGEPAGE, 1: MADV_NOHUGEPAGE, 2: hipHostMalloc
        uint8_t* h = nullptr;
        if (mode < 2) {
            h = (uint8_t*)mmap(nullptr, PIECE * NP + (2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (h == MAP_FAILED) return 1;
            h = (uint8_t*)(((uintptr_t)h + (2 << 20) - 1) & ~(uintptr_t)((2 << 20) - 1));
            if (madvise(h, PIECE * NP, mode == 0 ? MADV_HUGEPAGE : MADV_NOHUGEPAGE) != 0) perror("madvise");
            memset(h, 1, PIECE * NP);
            if (hipHostRegister(h, PIECE * NP, hipHostRegisterDefault) != hipSuccess) { printf("register failed\n"); return 1; }
        } else {
            if (hipHostMalloc(&h, PIECE * NP, hipHostMallocDefault) != hipSuccess) return 1;
            memset(h, 1, PIECE * NP);
        }
        const long hk = huge_kb_of(h);
        for (int dir = 0; dir < 2; ++dir)
            for (int warm = 0; warm < 2; ++warm) {
                (void)hipStreamSynchronize(s);
                const double t0 = now();
                for (int r = 0; r < REPS; ++r) {
                    if (dir == 0) (void)hipMemcpyAsync(d + PIECE * (r % NP), h + PIECE * (r % NP), PIECE, hipMemcpyHostToDevice, s);
                    else (void)hipMemcpyAsync(h + PIECE * (r % NP), d + PIECE * (r % NP), PIECE, hipMemcpyDeviceToHost, s);
                }
                (void)hipStreamSynchronize(s);
                if (warm) printf("%-16s AnonHugePages %8ld kB of %zu  %s %6.1f GB/s\n", mode == 0 ? "MADV_HUGEPAGE" : mode == 1 ? "MADV_NOHUGEPAGE" : "hipHostMalloc", hk, PIECE * NP / 1024,
                                 dir == 0 ? "H2D" : "D2H", REPS * PIECE / (now() - t0) / 1e9);
            }
    }
    return 0;
}
