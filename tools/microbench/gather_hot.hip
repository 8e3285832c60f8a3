// Random lines of a 3.5 GB table, but only `hot` distinct ones of them (scattered over the whole table): what does a working set
// between the L2s (8 x 4 MB) and the Infinity Cache (256 MB) cost per request? (bench.py under PA_SIM_TX_LIMIT shows a hump there)
// Run: hipcc --offload-arch=gfx950 -O3 gather_hot.hip -o gather_hot && ./gather_hot
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
__global__ __launch_bounds__(256) void gather(const uint4* __restrict__ buf, uint64_t nlines, uint64_t hot, int iters, uint32_t* out) {
    uint64_t key = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        key = mix(key);
        const uint64_t h = (uint64_t)(((unsigned __int128)key * hot) >> 64);                       // which hot line
        const uint64_t line = (uint64_t)(((unsigned __int128)mix(h * 0x9e3779b97f4a7c15ull + 1) * nlines) >> 64);   // where it lives
        acc += buf[line * 4].x;
        key ^= acc;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
    const uint64_t bytes = 3500ull << 20, nlines = bytes / 64;
    uint4* buf; uint32_t* out;
    CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 1, bytes));
    const int blocks = 256 * 8, iters = 64;
    CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint64_t hots[] = {1ull << 10, 1ull << 14, 1ull << 16, 1ull << 18, 1ull << 19, 1ull << 20, 1ull << 21, 1ull << 22, 1ull << 23, 1ull << 24, nlines};
    for (uint64_t hot : hots) {
        hipLaunchKernelGGL(gather, dim3(blocks), dim3(256), 0, 0, buf, nlines, hot, iters, out);
        CK(hipEventRecord(e0));
        for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(gather, dim3(blocks), dim3(256), 0, 0, buf, nlines, hot, iters, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("hot lines %9llu (%7.1f MB)  %7.2f G lane-fetches/s\n", (unsigned long long)hot, hot * 64 / 1e6, 4.0 * blocks * 256 * iters / ms / 1e6);
    }
    return 0;
}
