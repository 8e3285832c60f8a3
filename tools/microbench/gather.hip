// Random 64-byte-line gather microbenchmark for MI355X: how many independent cache-line requests per second can the
// memory system serve when every lane of every wave asks for a different line (the access pattern of the dictionary
// probe and the node-blob fetch)? Run: hipcc --offload-arch=gfx950 -O3 gather.hip -o gather && ./gather
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

template <int MLP, bool DEP>
__global__ __launch_bounds__(256) void gather(const uint4* __restrict__ buf, uint64_t nlines, int iters, uint32_t* out) {
    uint64_t key = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        uint4 v[MLP];
#pragma unroll
        for (int j = 0; j < MLP; ++j) {
            key = mix(key + j);
            const uint64_t line = (uint64_t)(((unsigned __int128)key * nlines) >> 64);
            v[j] = buf[line * 4];   // first 16 bytes of a 64-byte line
        }
#pragma unroll
        for (int j = 0; j < MLP; ++j) acc += v[j].x + v[j].w;
        if (DEP) key ^= acc;   // next address depends on the data: a pointer chase per lane
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MLP, bool DEP>
int run(const uint4* buf, uint64_t nlines, uint32_t* out, int blocks, const char* what) {
    const int iters = 64;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((gather<MLP, DEP>), dim3(blocks), dim3(256), 0, 0, buf, nlines, 4, out);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((gather<MLP, DEP>), dim3(blocks), dim3(256), 0, 0, buf, nlines, iters, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double req = (double)blocks * 256 * iters * MLP;
    printf("%-10s lines=%10llu (%7.1f MB) blocks/CU=%d MLP=%d dep=%d : %7.2f G lines/s  (%6.1f GB/s at 64 B/line)  %.3f ms\n", what,
           (unsigned long long)nlines, nlines * 64 / 1e6, blocks / 256, MLP, (int)DEP, req / ms / 1e6, req * 64 / ms / 1e6, ms);
    return 0;
}

int main() {
    const uint64_t big = (3500ull << 20) / 64;
    uint4* buf; uint32_t* out;
    CK(hipMalloc(&buf, big * 64));
    CK(hipMemset(buf, 1, big * 64));
    CK(hipMalloc(&out, 8192 * 256 * 4));
    struct { uint64_t lines; const char* name; } sizes[] = {{big, "HBM 3.5G"}, {(80ull << 20) / 64, "MALL 80M"}, {(2ull << 20) / 64, "L2 2M"}};
    for (auto& s : sizes) {
        for (int bpc : {4, 8}) {
            run<1, true>(buf, s.lines, out, 256 * bpc, s.name);
            run<2, true>(buf, s.lines, out, 256 * bpc, s.name);
            run<4, true>(buf, s.lines, out, 256 * bpc, s.name);
            run<4, false>(buf, s.lines, out, 256 * bpc, s.name);
        }
    }
    return 0;
}
