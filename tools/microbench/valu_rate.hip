// Issue rate of the integer VALU instructions the mapping kernel leans on (gfx950): wave64 instructions per cycle per SIMD, measured as
// the time of long unrolled runs of one instruction kind on independent registers, against v_add_u32 (full rate: one per 4 cycles).
// Run: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define REP8(s) s s s s s s s s
#define KERNEL(name, decl, body)                                                              \
    __global__ __launch_bounds__(256) void name(uint64_t* out, uint32_t n, uint64_t seed) {   \
        decl;                                                                                 \
        for (uint32_t i = 0; i < n; ++i) { REP8(REP8(body)) }                                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;                       \
    }
KERNEL(k_add32, uint32_t a0 = threadIdx.x + (uint32_t)seed; uint32_t a1 = a0 * 3; uint32_t a2 = a0 * 5; uint32_t a3 = a0 * 7,
       asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(i));)
KERNEL(k_mul32, uint32_t a0 = threadIdx.x + (uint32_t)seed; uint32_t a1 = a0 * 3; uint32_t a2 = a0 * 5; uint32_t a3 = a0 * 7,
       asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(i | 1u));)
KERNEL(k_mulhi32, uint32_t a0 = threadIdx.x + (uint32_t)seed; uint32_t a1 = a0 * 3; uint32_t a2 = a0 * 5; uint32_t a3 = a0 * 7,
       asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(i | 0x80000001u));)
KERNEL(k_shr64, uint64_t a0 = threadIdx.x + seed; uint64_t a1 = a0 * 3; uint64_t a2 = a0 * 5; uint64_t a3 = a0 * 7,
       asm volatile("v_lshrrev_b64 %0, 1, %0\n v_lshrrev_b64 %1, 1, %1\n v_lshrrev_b64 %2, 1, %2\n v_lshrrev_b64 %3, 1, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
KERNEL(k_shl64, uint64_t a0 = threadIdx.x + seed; uint64_t a1 = a0 * 3; uint64_t a2 = a0 * 5; uint64_t a3 = a0 * 7,
       asm volatile("v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %1, 1, %1\n v_lshlrev_b64 %2, 1, %2\n v_lshlrev_b64 %3, 1, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
KERNEL(k_lshladd64, uint64_t a0 = threadIdx.x + seed; uint64_t a1 = a0 * 3; uint64_t a2 = a0 * 5; uint64_t a3 = a0 * 7,
       asm volatile("v_lshl_add_u64 %0, %0, 1, %0\n v_lshl_add_u64 %1, %1, 1, %1\n v_lshl_add_u64 %2, %2, 1, %2\n v_lshl_add_u64 %3, %3, 1, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
KERNEL(k_alignbit, uint32_t a0 = threadIdx.x + (uint32_t)seed; uint32_t a1 = a0 * 3; uint32_t a2 = a0 * 5; uint32_t a3 = a0 * 7,
       asm volatile("v_alignbit_b32 %0, %0, %1, %4\n v_alignbit_b32 %1, %1, %2, %4\n v_alignbit_b32 %2, %2, %3, %4\n v_alignbit_b32 %3, %3, %0, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(i & 31u));)
KERNEL(k_bcnt, uint32_t a0 = threadIdx.x + (uint32_t)seed; uint32_t a1 = a0 * 3; uint32_t a2 = a0 * 5; uint32_t a3 = a0 * 7,
       asm volatile("v_bcnt_u32_b32 %0, %0, %4\n v_bcnt_u32_b32 %1, %1, %4\n v_bcnt_u32_b32 %2, %2, %4\n v_bcnt_u32_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(i));)
KERNEL(k_cndmask, uint32_t a0 = threadIdx.x + (uint32_t)seed; uint32_t a1 = a0 * 3; uint32_t a2 = a0 * 5; uint32_t a3 = a0 * 7; const uint64_t cm = seed * 0x9E3779B97F4A7C15ull,
       asm volatile("v_cndmask_b32 %0, %0, %1, %4\n v_cndmask_b32 %1, %1, %2, %4\n v_cndmask_b32 %2, %2, %3, %4\n v_cndmask_b32 %3, %3, %0, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(cm));)
KERNEL(k_and32, uint32_t a0 = threadIdx.x + (uint32_t)seed; uint32_t a1 = a0 * 3; uint32_t a2 = a0 * 5; uint32_t a3 = a0 * 7,
       asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(~i));)
KERNEL(k_shl32, uint32_t a0 = threadIdx.x + (uint32_t)seed; uint32_t a1 = a0 * 3; uint32_t a2 = a0 * 5; uint32_t a3 = a0 * 7,
       asm volatile("v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
KERNEL(k_bfe, uint32_t a0 = threadIdx.x + (uint32_t)seed; uint32_t a1 = a0 * 3; uint32_t a2 = a0 * 5; uint32_t a3 = a0 * 7,
       asm volatile("v_bfe_u32 %0, %0, 1, 31\n v_bfe_u32 %1, %1, 1, 31\n v_bfe_u32 %2, %2, 1, 31\n v_bfe_u32 %3, %3, 1, 31" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
KERNEL(k_add3, uint32_t a0 = threadIdx.x + (uint32_t)seed; uint32_t a1 = a0 * 3; uint32_t a2 = a0 * 5; uint32_t a3 = a0 * 7,
       asm volatile("v_add3_u32 %0, %0, %4, %4\n v_add3_u32 %1, %1, %4, %4\n v_add3_u32 %2, %2, %4, %4\n v_add3_u32 %3, %3, %4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(i));)
KERNEL(k_cmp, uint32_t a0 = threadIdx.x + (uint32_t)seed; uint32_t a1 = a0 * 3; uint32_t a2 = a0 * 5; uint32_t a3 = a0 * 7,
       asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_cmp_lt_u32 vcc, %1, %4\n v_cmp_lt_u32 vcc, %2, %4\n v_cmp_lt_u32 vcc, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(i) : "vcc");)
KERNEL(k_mad64, uint64_t a0 = threadIdx.x + seed; uint64_t a1 = a0 * 3; uint64_t a2 = a0 * 5; uint64_t a3 = a0 * 7,
       asm volatile("v_mad_u64_u32 %0, vcc, %4, %4, %0\n v_mad_u64_u32 %1, vcc, %4, %4, %1\n v_mad_u64_u32 %2, vcc, %4, %4, %2\n v_mad_u64_u32 %3, vcc, %4, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(i | 1u) : "vcc");)

template <class K>
int run(K k, const char* what, uint64_t* out, double* base) {
    const uint32_t n = 2000, blocks = 256 * 8;   // 8 waves per SIMD: the pipe is never short of independent work
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 10u, (uint64_t)1);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, n, (uint64_t)1);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double insts = (double)blocks * 4 * n * 64 * 4;   // wave instructions
    const double per_simd_per_us = insts / 1024.0 / (ms * 1000.0);
    if (*base == 0) *base = ms;
    printf("%-18s %8.3f ms  %7.1f wave-instructions per SIMD per us  x%.2f of v_add_u32's time\n", what, ms, per_simd_per_us, ms / *base);
    return 0;
}
int main() {
    uint64_t* out;
    CK(hipMalloc(&out, 256 * 8 * 256 * 8));
    double base = 0;
    if (run(k_add32, "v_add_u32", out, &base)) return 1;
    run(k_and32, "v_and_b32", out, &base); run(k_shl32, "v_lshlrev_b32", out, &base); run(k_bfe, "v_bfe_u32", out, &base); run(k_add3, "v_add3_u32", out, &base);
    run(k_cmp, "v_cmp_lt_u32", out, &base); run(k_cndmask, "v_cndmask_b32", out, &base); run(k_alignbit, "v_alignbit_b32", out, &base); run(k_bcnt, "v_bcnt_u32_b32", out, &base);
    run(k_shr64, "v_lshrrev_b64", out, &base); run(k_shl64, "v_lshlrev_b64", out, &base); run(k_lshladd64, "v_lshl_add_u64", out, &base);
    run(k_mul32, "v_mul_lo_u32", out, &base); run(k_mulhi32, "v_mul_hi_u32", out, &base); run(k_mad64, "v_mad_u64_u32", out, &base);
    return 0;
}
