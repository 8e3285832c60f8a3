// Where the pinned host memory of a copy lives, and what that does to the link: for every NUMA node of the host, 4 x 64 MiB of anonymous
// memory bound to that node (mbind before the first touch), registered with HIP, then copied to the GPU and back in 64 MiB pieces on one stream;
// the same once more while 16 threads memcpy into OTHER buffers of the same node (what pa_process_reads' readers do while a window flies).
// build: hipcc -O2 --offload-arch=gfx950 -o h2d_numa h2d_numa.hip -lpthread     (measurement only; nothing in the product uses it)
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <sched.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void* node_alloc(size_t bytes, int node) {
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) return nullptr;
    if (node >= 0) {
        unsigned long mask[16] = {0};
        mask[node / 64] |= 1ul << (node % 64);
        if (syscall(SYS_mbind, p, bytes, 2 /* MPOL_BIND */, mask, 1024ul, 0ul) != 0) perror("mbind");
    }
    memset(p, 1, bytes);
    return p;
}

static int read_int(const std::string& path, int dflt) {
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return dflt;
    int v = dflt;
    if (fscanf(f, "%d", &v) != 1) v = dflt;
    fclose(f);
    return v;
}

int main(int argc, char** argv) {
    const size_t PIECE = 64ull << 20;
    const int NP = 4, REPS = 24;
    int nodes = 0;
    while (access(("/sys/devices/system/node/node" + std::to_string(nodes)).c_str(), F_OK) == 0) ++nodes;
    char bdf[64] = {0};
    hipDeviceGetPCIBusId(bdf, sizeof bdf, 0);
    std::string b(bdf);
    for (auto& ch : b) ch = (char)tolower(ch);
    const int gpu_node = read_int("/sys/bus/pci/devices/" + b + "/numa_node", -1);
    printf("host NUMA nodes %d, GPU 0 at %s on node %d, this thread on cpu %d\n", nodes, b.c_str(), gpu_node, sched_getcpu());
    uint8_t* d = nullptr;
    if (hipMalloc(&d, PIECE * NP) != hipSuccess) return 1;
    hipStream_t s, s2;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    for (int node = -1; node < nodes; ++node) {
        uint8_t* h = (uint8_t*)node_alloc(PIECE * NP, node);
        uint8_t* h2 = (uint8_t*)node_alloc(PIECE * NP, node);
        uint8_t* src = (uint8_t*)node_alloc(PIECE * NP, node);
        if (!h || !h2 || !src) return 1;
        if (hipHostRegister(h, PIECE * NP, hipHostRegisterDefault) != hipSuccess) { printf("register failed\n"); return 1; }
        if (hipHostRegister(h2, PIECE * NP, hipHostRegisterDefault) != hipSuccess) { printf("register failed\n"); return 1; }
        for (int mode = 0; mode < 4; ++mode) {   // 0: H2D alone, 1: D2H alone, 2: both, 3: H2D while 16 threads memcpy into the other registered buffer
            std::atomic<bool> stop{false};
            std::vector<std::thread> th;
            std::atomic<uint64_t> copied{0};
            if (mode == 3)
                for (int t = 0; t < 16; ++t)
                    th.emplace_back([&, t] {
                        const size_t part = PIECE * NP / 16;
                        while (!stop.load()) { memcpy(h2 + part * t, src + part * t, part); copied += part; }
                    });
            for (int warm = 0; warm < 2; ++warm) {
                hipStreamSynchronize(s); hipStreamSynchronize(s2);
                const double t0 = now();
                const uint64_t c0 = copied.load();
                for (int r = 0; r < REPS; ++r) {
                    if (mode != 1) hipMemcpyAsync(d + PIECE * (r % NP), h + PIECE * (r % NP), PIECE, hipMemcpyHostToDevice, s);
                    if (mode == 1 || mode == 2) hipMemcpyAsync(h2 + PIECE * (r % NP), d + PIECE * ((r + 2) % NP), PIECE, hipMemcpyDeviceToHost, s2);
                }
                hipStreamSynchronize(s); hipStreamSynchronize(s2);
                const double dt = now() - t0;
                if (warm) printf("node %2d  %-28s %6.1f GB/s per direction%s\n", node, mode == 0 ? "H2D" : mode == 1 ? "D2H" : mode == 2 ? "H2D + D2H" : "H2D + 16 threads of memcpy",
                                 REPS * PIECE / dt / 1e9, mode == 3 ? (", memcpy " + std::to_string((copied.load() - c0) / dt / 1e9) + " GB/s").c_str() : "");
            }
            stop.store(true);
            for (auto& t : th) t.join();
        }
        hipHostUnregister(h); hipHostUnregister(h2);
        munmap(h, PIECE * NP); munmap(h2, PIECE * NP); munmap(src, PIECE * NP);
    }
    // hipHostMalloc as the library uses it today
    uint8_t* hm = nullptr;
    if (hipHostMalloc(&hm, PIECE * NP, hipHostMallocDefault) == hipSuccess) {
        memset(hm, 1, PIECE * NP);
        int where = -1;
        syscall(SYS_get_mempolicy, &where, nullptr, 0ul, hm, 3ul /* MPOL_F_NODE | MPOL_F_ADDR */);
        for (int warm = 0; warm < 2; ++warm) {
            const double t0 = now();
            for (int r = 0; r < REPS; ++r) hipMemcpyAsync(d + PIECE * (r % NP), hm + PIECE * (r % NP), PIECE, hipMemcpyHostToDevice, s);
            hipStreamSynchronize(s);
            if (warm) printf("hipHostMalloc (first page on node %d)  H2D %6.1f GB/s\n", where, REPS * PIECE / (now() - t0) / 1e9);
        }
    }
    return 0;
}
