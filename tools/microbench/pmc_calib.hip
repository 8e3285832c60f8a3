// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 in the access patterns of the mapping kernel
// (MI355X_MICROARCH.md: "calibrate on a known byte count in your own access pattern before trusting an absolute"):
//   calib_stream   coalesced 16 B per lane over 4 GiB                    -> 4.29e9 bytes read
//   calib_stream8 / calib_stream4   the same 4 GiB with 8 / 4 bytes per lane
//   calib_gather   every lane one 16-byte load of a random 64-byte line of an 8 GiB table, 2^27 loads -> 8.59e9 bytes of lines
//   calib_store    coalesced 16-byte stores, 2^27 of them                  -> 2.15e9 bytes written
//   calib_scatter  16-byte stores to random 64-byte lines of a 4 GiB table, 2^27 of them
//   calib_atomic   u64 atomicAdd (no return) on random words of a 4 GiB table, 2^27 of them
//   calib_block    the chain-block access of round 4+ (lane_steps.hpp, fwd_issue): every lane ONE random 128-byte block — four 16-byte loads from its
//                  first half (the slots) and 16 / 32 / 48 bytes from its second (sequence words), 2^27 blocks; on a 4 GiB table (HBM) and, as
//                  calib_block_small, on a 384 MiB one (the size of the config-3 chain blocks: partly Infinity-Cache resident)
//   calib_slot16   the dictionary probe: ONE 16-byte slot (any of the four) of a random 64-byte line of a 3.25 GiB table, 2^27 of them
//   calib_records  the result records: 16-byte stores to results[rid], the 64 lanes of a store spread over a window of 160 consecutive
//                  records that moves on by 63 per step (an output step of the mapping kernel), 2^27 records = 2.147e9 bytes
// Run: hipcc --offload-arch=gfx950 -O3 pmc_calib.hip -o pmc_calib; rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -- ./pmc_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

__global__ __launch_bounds__(256) void calib_stream(const uint4* __restrict__ buf, uint64_t n16, uint32_t* out) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) acc += buf[i].x;
    if (acc == 0x12345678u) out[0] = acc;
}
// the read tiles of the mapping kernel: 8 bytes per lane, lanes on consecutive words (512 B per wave and load)
__global__ __launch_bounds__(256) void calib_stream8(const uint64_t* __restrict__ buf, uint64_t n8, uint32_t* out) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (uint64_t)gridDim.x * blockDim.x) acc += (uint32_t)buf[i];
    if (acc == 0x12345678u) out[0] = acc;
}
// ... and its lengths: 4 bytes per lane
__global__ __launch_bounds__(256) void calib_stream4(const uint32_t* __restrict__ buf, uint64_t n4, uint32_t* out) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * blockDim.x) acc += buf[i];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void calib_gather(const uint4* __restrict__ buf, uint64_t nlines, int iters, uint32_t* out) {
    uint64_t key = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        key = mix(key + i);
        acc += buf[(uint64_t)(((unsigned __int128)key * nlines) >> 64) * 4].x;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void calib_store(uint4* __restrict__ buf, uint64_t n16) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) buf[i] = uint4{(uint32_t)i, 1u, 2u, 3u};
}
__global__ __launch_bounds__(256) void calib_scatter(uint4* __restrict__ buf, uint64_t nlines, int iters) {
    uint64_t key = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
    for (int i = 0; i < iters; ++i) {
        key = mix(key + i);
        buf[(uint64_t)(((unsigned __int128)key * nlines) >> 64) * 4] = uint4{(uint32_t)key, 1u, 2u, 3u};
    }
}
__global__ __launch_bounds__(256) void calib_atomic(unsigned long long* __restrict__ buf, uint64_t nwords, int iters) {
    uint64_t key = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
    for (int i = 0; i < iters; ++i) {
        key = mix(key + i);
        atomicAdd(buf + (uint64_t)(((unsigned __int128)key * nwords) >> 64), 1ull);
    }
}

template <int SEQ16>   // 16-byte pieces of the second half that a lane loads (1..3)
__global__ __launch_bounds__(256) void calib_block_k(const uint4* __restrict__ buf, uint64_t nblocks, int iters, uint32_t* out) {
    uint64_t key = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        key = mix(key + i);
        const uint4* b = buf + (uint64_t)(((unsigned __int128)key * nblocks) >> 64) * 8;
        const uint4 s0 = b[0], s1 = b[1], s2 = b[2], s3 = b[3];
        const uint32_t w = (uint32_t)(key & 1u);   // the sequence words start at word 0 or 2 of the eight
        uint4 q0 = b[4 + w], q1 = SEQ16 > 1 ? b[5 + w] : uint4{0, 0, 0, 0}, q2 = SEQ16 > 2 ? b[6 + w] : uint4{0, 0, 0, 0};
        acc += s0.x + s1.y + s2.z + s3.w + q0.x + q1.y + q2.z;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void calib_slot16(const uint4* __restrict__ buf, uint64_t nlines, int iters, uint32_t* out) {
    uint64_t key = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        key = mix(key + i);
        acc += buf[(uint64_t)(((unsigned __int128)key * nlines) >> 64) * 4 + (key & 3u)].z;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void calib_records(uint4* __restrict__ buf, int iters) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint64_t key = wave * 64 + lane + 1;
    const uint64_t base = wave * (uint64_t)iters * 64;   // this wave's records [base, base + 64 * iters)
    for (int i = 0; i < iters; ++i) {
        // record i * 64 + lane, displaced: the wave retires reads out of a window of ~160 in flight, so the records of one step are
        // 64 of the 160 after the oldest (a random subset: lane l takes position perm within its 2.5-wide stripe)
        key = mix(key + i);
        uint64_t r = (uint64_t)i * 64 + (uint64_t)((lane * 5u) >> 1) + (key & 1u);
        if (r >= (uint64_t)iters * 64) r = (uint64_t)iters * 64 - 1 - lane;
        buf[base + r] = uint4{(uint32_t)key, 1u, 2u, 3u};
    }
}

int main() {
    const uint64_t bytes = 8ull << 30;
    uint4* buf; uint32_t* out;
    CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 1, bytes)); CK(hipMalloc(&out, 64));
    const int blocks = 2048, iters = 256;   // 2048 * 256 * 256 = 2^27 lane operations
    hipLaunchKernelGGL(calib_stream, dim3(blocks * 4), dim3(256), 0, 0, buf, (4ull << 30) / 16, out);
    hipLaunchKernelGGL(calib_stream8, dim3(blocks * 4), dim3(256), 0, 0, (const uint64_t*)buf, (4ull << 30) / 8, out);
    hipLaunchKernelGGL(calib_stream4, dim3(blocks * 4), dim3(256), 0, 0, (const uint32_t*)buf, (4ull << 30) / 4, out);
    hipLaunchKernelGGL(calib_gather, dim3(blocks), dim3(256), 0, 0, buf, bytes / 64, iters, out);
    hipLaunchKernelGGL(calib_store, dim3(blocks * 4), dim3(256), 0, 0, buf, 1ull << 27);
    hipLaunchKernelGGL(calib_scatter, dim3(blocks), dim3(256), 0, 0, buf, (4ull << 30) / 64, iters);
    hipLaunchKernelGGL(calib_atomic, dim3(blocks), dim3(256), 0, 0, (unsigned long long*)buf, (4ull << 30) / 8, iters);
    hipLaunchKernelGGL(calib_block_k<1>, dim3(blocks), dim3(256), 0, 0, buf, (4ull << 30) / 128, iters, out);
    hipLaunchKernelGGL(calib_block_k<2>, dim3(blocks), dim3(256), 0, 0, buf, (4ull << 30) / 128, iters, out);
    hipLaunchKernelGGL(calib_block_k<3>, dim3(blocks), dim3(256), 0, 0, buf, (4ull << 30) / 128, iters, out);
    hipLaunchKernelGGL(calib_block_k<2>, dim3(blocks + 1), dim3(256), 0, 0, buf, (384ull << 20) / 128, iters, out);   // (grid + 1: told apart in the csv by its grid size)
    hipLaunchKernelGGL(calib_slot16, dim3(blocks), dim3(256), 0, 0, buf, (3328ull << 20) / 64, iters, out);
    hipLaunchKernelGGL(calib_records, dim3(blocks), dim3(256), 0, 0, buf, iters);
    CK(hipDeviceSynchronize());
    printf("block: 2^27 blocks = 1.718e10 B of blocks (8.59e9 B of first halves); slot16: 2^27 lines = 8.59e9 B; records: 2.147e9 B written\n");
    printf("expected: stream 4.295e9 B read; gather 2^27 lines = 8.590e9 B; store 2.147e9 B written; scatter / atomic 2^27 operations\n");
    return 0;
}
