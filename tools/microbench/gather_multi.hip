// What does one more 16-byte load from a line a lane already touches cost? And what does a wave pay when 4 lanes share a
// 64-byte line instead of 64 lanes touching 64 lines? (The node visit of the mapping kernel issues 5-6 loads per lane into
// one 128-byte block; the vector L1 sees ~19 accesses per read while the L2 sees ~5 requests.)
//   mode N     every lane: N x 16 B from ONE random 128-byte block (N = 1, 2, 3, 6), all in flight together
//   mode quad  four neighbouring lanes read the four 16-byte pieces of one random 64-byte line: 16 lines per wave-instruction
//   mode dep2  16 B, then a dependent 12-byte load from the same line (the dictionary probe)
// Also prints the average issue-to-use latency seen by lane 0 of every wave (s_memtime ticks).
// Run: hipcc --offload-arch=gfx950 -O3 gather_multi.hip -o gather_multi && ./gather_multi
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

enum { M_N = 0, M_QUAD = 1, M_DEP2 = 2 };

template <int MODE, int N>
__global__ __launch_bounds__(256) void gather(const uint4* __restrict__ buf, uint64_t nblocks, int iters, uint32_t* out, unsigned long long* lat) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t key = (MODE == M_QUAD ? (tid >> 2) : tid) + 1;
    uint32_t acc = 0;
    unsigned long long ticks = 0;
    for (int i = 0; i < iters; ++i) {
        key = mix(key);
        const uint64_t blk = (uint64_t)(((unsigned __int128)key * nblocks) >> 64);
        const unsigned long long t0 = __builtin_readcyclecounter();
        if (MODE == M_N) {
            uint4 v[N];
#pragma unroll
            for (int j = 0; j < N; ++j) v[j] = buf[blk * 8 + j];
#pragma unroll
            for (int j = 0; j < N; ++j) acc += v[j].x + v[j].w;
        } else if (MODE == M_QUAD) {
            const uint4 v = buf[blk * 8 + (tid & 3)];
            acc += v.x + v.w;
        } else {
            const uint4 v = buf[blk * 8];
            const uint32_t j = (v.x + (uint32_t)key) & 3;
            const uint32_t* e = (const uint32_t*)(buf + blk * 8 + 1) + 3 * j;
            acc += v.w + e[0] + e[1] + e[2];
        }
        asm volatile("" : "+v"(acc));
        ticks += __builtin_readcyclecounter() - t0;
        key ^= acc;
    }
    out[tid] = acc;
    if ((threadIdx.x & 63) == 0) atomicAdd(lat, ticks);
}

template <int MODE, int N>
int run(const uint4* buf, uint64_t nblocks, uint32_t* out, unsigned long long* lat, int blocks, const char* what) {
    const int iters = 64;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((gather<MODE, N>), dim3(blocks), dim3(256), 0, 0, buf, nblocks, 4, out, lat);
    CK(hipMemset(lat, 0, 8));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((gather<MODE, N>), dim3(blocks), dim3(256), 0, 0, buf, nblocks, iters, out, lat);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    unsigned long long t; CK(hipMemcpy(&t, lat, 8, hipMemcpyDeviceToHost));
    const double lanes = (double)blocks * 256 * iters;
    const double lines = MODE == M_QUAD ? lanes / 4 : lanes;
    const char* mode = MODE == M_N ? "N" : MODE == M_QUAD ? "quad" : "dep2";
    printf("%-9s blocks/CU=%d mode=%-4s N=%d : %7.2f G lines/s %7.2f G lane-loads/s  %8.3f ms  latency %6.0f ticks/iter\n", what, blocks / 256, mode,
           MODE == M_N ? N : 1, lines / ms / 1e6, lanes * (MODE == M_N ? N : MODE == M_DEP2 ? 2 : 1) / ms / 1e6, ms, (double)t / ((double)blocks * 4 * iters));
    return 0;
}

int main() {
    const uint64_t big = (3500ull << 20) / 128;
    uint4* buf; uint32_t* out; unsigned long long* lat;
    CK(hipMalloc(&buf, big * 128));
    CK(hipMemset(buf, 1, big * 128));
    CK(hipMalloc(&out, 8192 * 256 * 4));
    CK(hipMalloc(&lat, 8));
    struct { uint64_t blocks; const char* name; } sizes[] = {{big, "HBM 3.5G"}, {(100ull << 20) / 128, "MALL 100M"}, {(2ull << 20) / 128, "L2 2M"}};
    for (auto& s : sizes) {
        for (int bpc : {1, 3, 8}) {
            run<M_N, 1>(buf, s.blocks, out, lat, 256 * bpc, s.name);
            run<M_N, 2>(buf, s.blocks, out, lat, 256 * bpc, s.name);
            run<M_N, 3>(buf, s.blocks, out, lat, 256 * bpc, s.name);
            run<M_N, 6>(buf, s.blocks, out, lat, 256 * bpc, s.name);
            run<M_QUAD, 1>(buf, s.blocks, out, lat, 256 * bpc, s.name);
            run<M_DEP2, 1>(buf, s.blocks, out, lat, 256 * bpc, s.name);
        }
    }
    return 0;
}
