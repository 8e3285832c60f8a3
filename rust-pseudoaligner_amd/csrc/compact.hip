// Compact per-read records for the way back to the host (SURVEY §8d measures host to host: the 16-byte pa_read_result records of a
// 100 M-read batch are 1.6 GB over a link whose other direction carries the 4 GB of reads — and the two directions slow each other
// down). What map_read returns (src/pseudoaligner.rs:361-384) fits 8 bytes when the class is an index class (by reference) and
// the few that are not travel as {length, ids...} in ONE packed stream in read order:
//
//   pa_compact_len_kernel   words a read's class takes in the packed stream (0: by reference / empty / unmapped; else 1 + class_len)
//   rocPRIM exclusive scan  where they start; the last entry is the stream's length
//   pa_compact_write_kernel the 8-byte records and the packed classes (copied out of the launch's arena, whose waves' private slices
//                           are two thirds padding)
#include <hip/hip_runtime.h>
#include <rocprim/device/device_scan.hpp>

#include "kernels.hpp"
#include "pa_common.hpp"

namespace pa {
namespace {

__device__ __forceinline__ bool packed_class(const pa_read_result& r, uint64_t arena_cap) {
    return (r.mismatches & PA_MAPPED_BIT) && !(r.class_off & PA_CLASS_REF) && r.class_len != 0 && (uint64_t)r.class_off + r.class_len <= arena_cap;
}

__global__ __launch_bounds__(256) void pa_compact_len_kernel(const pa_read_result* __restrict__ results, uint64_t n, uint64_t arena_cap, uint32_t* __restrict__ len) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    uint32_t l = 0;
    if (i < n) {
        const pa_read_result r = results[i];
        if (packed_class(r, arena_cap)) l = 1u + r.class_len;
    }
    len[i] = l;   // (len[n] = 0: the scan's last entry is the total)
}

__global__ __launch_bounds__(256) void pa_compact_write_kernel(const pa_read_result* __restrict__ results, const uint32_t* __restrict__ arena, uint64_t n, uint64_t arena_cap,
                                                               const uint64_t* __restrict__ off, uint64_t* __restrict__ compact, uint32_t* __restrict__ packed,
                                                               uint64_t packed_cap, unsigned long long* __restrict__ packed_words) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *packed_words = off[n];
    if (i >= n) return;
    const pa_read_result r = results[i];
    const bool mapped = r.mismatches & PA_MAPPED_BIT;
    uint32_t lo = (r.coverage & 0x3FFFu) | ((r.mismatches & 0x3FFFu) << 14) | (mapped ? PA_COMPACT_MAPPED : 0u), hi = 0;
    if (mapped && r.class_len != 0) {
        if (r.class_off & PA_CLASS_REF) { lo |= PA_COMPACT_BY_REF; hi = r.class_off & ~PA_CLASS_REF; }
        else if (packed_class(r, arena_cap)) {
            const uint64_t o = off[i];
            lo |= PA_COMPACT_PACKED;
            hi = (uint32_t)o;   // (the low 32 bits: classes follow each other in read order, a reader that walks the records needs none of it)
            if (o + 1 + r.class_len <= packed_cap) {
                packed[o] = r.class_len;
                for (uint32_t j = 0; j < r.class_len; ++j) packed[o + 1 + j] = arena[r.class_off + j];
            }
        } else lo |= PA_COMPACT_PACKED | PA_COMPACT_BY_REF;   // a class whose ids did not fit the launch's arena (PA_ERR_ARENA_FULL is what pa_map_finish said): both bits = lost
    }
    compact[i] = (uint64_t)lo | ((uint64_t)hi << 32);
}

size_t scan_bytes(uint64_t n) {
    size_t bytes = 0;
    (void)rocprim::exclusive_scan(nullptr, bytes, (const uint32_t*)nullptr, (uint64_t*)nullptr, (uint64_t)0, (size_t)(n + 1), rocprim::plus<uint64_t>(), (hipStream_t) nullptr);
    return (bytes + 255) & ~(size_t)255;
}

}  // namespace
}  // namespace pa

using namespace pa;

extern "C" size_t pa_compact_scratch_bytes(uint64_t n_reads) {
    return (((size_t)(n_reads + 1) * 4 + 255) & ~(size_t)255) + (((size_t)(n_reads + 1) * 8 + 255) & ~(size_t)255) + scan_bytes(n_reads);
}

extern "C" int pa_results_compact_device(pa_index* idx, const pa_read_result* d_results, const uint32_t* d_arena, uint64_t arena_cap, uint64_t n_reads, uint64_t* d_compact,
                                         uint32_t* d_packed, uint64_t packed_cap, uint64_t* d_packed_words, void* d_scratch, size_t scratch_bytes, void* stream) {
    if (!idx || !d_results || !d_compact || !d_packed_words || !d_scratch || (packed_cap && !d_packed)) return fail(PA_ERR_INVALID_ARG, "null argument");
    if (scratch_bytes < pa_compact_scratch_bytes(n_reads)) return fail(PA_ERR_INVALID_ARG, "scratch of %zu bytes, %zu needed", scratch_bytes, pa_compact_scratch_bytes(n_reads));
    int device = 0;
    const uint32_t *h_ec = nullptr, *h_ref = nullptr;
    index_host_classes(idx, &h_ec, &h_ref, &device);
    if (hipSetDevice(device) != hipSuccess) return fail(PA_ERR_HIP, "hipSetDevice(%d) failed", device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    uint8_t* base = static_cast<uint8_t*>(d_scratch);
    uint32_t* d_len = reinterpret_cast<uint32_t*>(base);
    uint64_t* d_off = reinterpret_cast<uint64_t*>(base + (((size_t)(n_reads + 1) * 4 + 255) & ~(size_t)255));
    void* d_tmp = reinterpret_cast<uint8_t*>(d_off) + (((size_t)(n_reads + 1) * 8 + 255) & ~(size_t)255);
    const size_t tmp_bytes = scan_bytes(n_reads);
    const uint32_t blocks = (uint32_t)((n_reads + 1 + 255) / 256);
    hipLaunchKernelGGL(pa_compact_len_kernel, dim3(blocks), dim3(256), 0, s, d_results, n_reads, arena_cap, d_len);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = rocprim::exclusive_scan(d_tmp, const_cast<size_t&>(tmp_bytes), (const uint32_t*)d_len, d_off, (uint64_t)0, (size_t)(n_reads + 1), rocprim::plus<uint64_t>(), s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(pa_compact_write_kernel, dim3(blocks), dim3(256), 0, s, d_results, d_arena, n_reads, arena_cap, (const uint64_t*)d_off, d_compact, d_packed, packed_cap,
                           reinterpret_cast<unsigned long long*>(d_packed_words));
        e = hipGetLastError();
    }
    if (e != hipSuccess) return fail(PA_ERR_HIP, "compact records: %s", hipGetErrorString(e));
    return PA_OK;
}
