// gather_pair.hip with the table size (MiB) as argument: does the request rate of random 128-byte blocks hold for tables of tens of GB?
// Run: hipcc --offload-arch=gfx950 -O3 gather_size.hip -o gather_size && ./gather_size 27000
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

// mode 0: lines 2b, 2b+1 (one 128-byte block); mode 1: lines 2b+1, 2b+2 (two blocks); mode 2: line 2b only
__global__ __launch_bounds__(256) void gather(const uint4* __restrict__ buf, uint64_t nblocks, int iters, int mode, uint32_t* out) {
    uint64_t key = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        key = mix(key);
        const uint64_t b = (uint64_t)(((unsigned __int128)key * nblocks) >> 64);
        const uint64_t l0 = 2 * b + (mode == 1 ? 1 : 0);
        const uint4 v0 = buf[l0 * 4];
        uint4 v1 = v0;
        if (mode != 2) v1 = buf[(l0 + 1) * 4];
        acc += v0.x + v1.w;
        key ^= acc;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
    const uint64_t bytes = (argc > 1 ? strtoull(argv[1], nullptr, 10) : 3500ull) << 20, nblocks = bytes / 128 - 2;
    uint4* buf; uint32_t* out;
    CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 1, bytes));
    const int blocks = 256 * 8, iters = 64;
    CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[3] = {"pair in one 128-byte block", "pair across two blocks", "single line"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            hipLaunchKernelGGL(gather, dim3(blocks), dim3(256), 0, 0, buf, nblocks, iters, mode, out);
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(gather, dim3(blocks), dim3(256), 0, 0, buf, nblocks, iters, mode, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double n = (double)blocks * 256 * iters;
            printf("%-30s %7.2f G lane-fetches/s  %.3f ms\n", names[mode], n / ms / 1e6, ms);
        }
    return 0;
}
