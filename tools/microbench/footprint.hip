// How does the rate of random 128-byte requests depend on the FOOTPRINT they are drawn from? (address translation: the anchor table of
// anchors.hpp is 7-27 GB where the dictionary is 3.3 GB.) One 16-byte load per lane from a random 128-byte block of a table of the given
// size; two dependent chains per lane. Run: hipcc --offload-arch=gfx950 -O3 footprint.hip -o footprint && ./footprint
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
template <int LOADS>
__global__ __launch_bounds__(256) void gather(const uint4* __restrict__ buf, uint64_t nblocks, int iters, uint32_t* out) {
    uint64_t key = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        uint4 v[2][LOADS];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            key = mix(key + j);
            const uint64_t b = (uint64_t)(((unsigned __int128)key * nblocks) >> 64);
#pragma unroll
            for (int l = 0; l < LOADS; ++l) v[j][l] = buf[b * 8 + l];   // LOADS x 16 bytes of one 128-byte block
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int l = 0; l < LOADS; ++l) acc += v[j][l].x + v[j][l].w;
        key ^= acc;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int LOADS>
int run(const uint4* buf, uint64_t nblocks, uint32_t* out, double gb) {
    const int iters = 64, blocks = 256 * 8;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(gather<LOADS>, dim3(blocks), dim3(256), 0, 0, buf, nblocks, 4, out);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(gather<LOADS>, dim3(blocks), dim3(256), 0, 0, buf, nblocks, iters, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double req = (double)blocks * 256 * iters * 2;
    printf("footprint %6.2f GB, %d x 16 B per block: %6.2f G blocks/s (%6.0f GB/s at 128 B)  %.3f ms\n", gb, LOADS, req / ms / 1e6, req * 128 / ms / 1e6, ms);
    return 0;
}
int main() {
    const double max_gb = 32;
    uint4* buf; uint32_t* out;
    CK(hipMalloc(&buf, (size_t)(max_gb * 1e9)));
    CK(hipMemset(buf, 1, (size_t)(max_gb * 1e9)));
    CK(hipMalloc(&out, 8192 * 256 * 4));
    for (double gb : {0.25, 1.0, 2.0, 3.0, 3.5, 4.0, 4.5, 5.0, 6.0, 7.0, 14.0, 28.0, 32.0}) {
        const uint64_t nb = (uint64_t)(gb * 1e9 / 128);
        if (run<1>(buf, nb, out, gb)) return 1;
        if (run<2>(buf, nb, out, gb)) return 1;
        if (run<4>(buf, nb, out, gb)) return 1;
        if (run<8>(buf, nb, out, gb)) return 1;
    }
    return 0;
}
