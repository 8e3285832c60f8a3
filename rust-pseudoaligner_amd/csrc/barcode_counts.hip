// Per-barcode (single-cell) class counts — SURVEY.md §8f.3, the use the reference was written for (README.md:3: a
// pseudo-alignment tool for single-cell RNA-seq): every read carries the index of its cell barcode, and what downstream wants
// is the SPARSE matrix (barcode, equivalence class) -> reads. On the GPU: one 64-bit key per read (barcode << 32 | column, the
// columns being those of the dense count table: class id, or novel / empty / unmapped), a radix sort of the keys and a
// run-length encode — sorted unique keys with their counts, ready for a CSR / triplet matrix on the host.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "kernel_utils.hpp"
#include "kernels.hpp"
#include "pa_common.hpp"

using namespace pa;

namespace {

__global__ __launch_bounds__(256) void pa_barcode_keys_kernel(const pa_read_result* __restrict__ results, const uint32_t* __restrict__ arena,
                                                              const uint32_t* __restrict__ barcode, uint64_t n_reads, const DevIndexView ix,
                                                              const uint32_t* __restrict__ class_table, uint64_t class_table_size,
                                                              unsigned long long* __restrict__ keys) {
    const uint32_t num_classes = ix.num_classes;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_reads) return;
    const pa_read_result r = results[i];
    uint32_t col;
    if (!(r.mismatches & PA_MAPPED_BIT)) col = num_classes + 2;
    else if (r.class_len == 0) col = num_classes + 1;
    else {
        uint32_t c = (r.class_off & PA_CLASS_REF) ? (r.class_off & ~PA_CLASS_REF) : class_of_list(arena + r.class_off, r.class_len, ix, class_table, class_table_size);
        col = c == 0xFFFFFFFFu ? num_classes : c;
    }
    keys[i] = ((unsigned long long)barcode[i] << 32) | col;
}

}  // namespace

namespace pa {

// device_index.hip hands over the pieces of the index this needs
int barcode_counts(const DevIndexView& ix, const uint32_t* class_table, uint64_t class_table_size, const pa_read_result* d_results,
                   const uint32_t* d_arena, const uint32_t* d_barcode, uint64_t n, uint32_t barcode_bits, uint64_t* d_keys, uint32_t* d_vals,
                   uint64_t* n_entries, hipStream_t stream) {
    *n_entries = 0;
    if (n == 0) return PA_OK;
    if (n > 0x7FFFFFFFull) return fail(PA_ERR_UNSUPPORTED, "at most 2^31-1 reads per call");
    unsigned long long *keys_in = nullptr, *keys_sorted = nullptr;
    unsigned int* d_runs = nullptr;
    void* tmp = nullptr;
    auto done = [&](int rc) {
        for (void* p : {(void*)keys_in, (void*)keys_sorted, (void*)d_runs, tmp})
            if (p) (void)hipFree(p);
        return rc;
    };
#define TRY_(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return done(fail(PA_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_))); } while (0)
    TRY_(hipMalloc(&keys_in, n * 8));
    TRY_(hipMalloc(&keys_sorted, n * 8));
    TRY_(hipMalloc(&d_runs, 4));
    hipLaunchKernelGGL(pa_barcode_keys_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, d_results, d_arena, d_barcode, n, ix, class_table,
                       class_table_size, keys_in);
    TRY_(hipGetLastError());
    uint32_t col_bits = 1;
    while (col_bits < 32 && (1ull << col_bits) < (uint64_t)ix.num_classes + 3) ++col_bits;
    const int end_bit = (int)(32 + (barcode_bits ? (barcode_bits > 32 ? 32 : barcode_bits) : 32));
    size_t sort_bytes = 0, rle_bytes = 0;
    // only the bits that can differ are sorted: the column's low bits and the barcode's
    TRY_(hipcub::DeviceRadixSort::SortKeys(nullptr, sort_bytes, keys_in, keys_sorted, (int)n, 0, end_bit, stream));
    TRY_(hipcub::DeviceRunLengthEncode::Encode(nullptr, rle_bytes, keys_sorted, (unsigned long long*)d_keys, d_vals, d_runs, (int)n, stream));
    TRY_(hipMalloc(&tmp, sort_bytes > rle_bytes ? sort_bytes : rle_bytes));
    (void)col_bits;
    TRY_(hipcub::DeviceRadixSort::SortKeys(tmp, sort_bytes, keys_in, keys_sorted, (int)n, 0, end_bit, stream));
    TRY_(hipcub::DeviceRunLengthEncode::Encode(tmp, rle_bytes, keys_sorted, (unsigned long long*)d_keys, d_vals, d_runs, (int)n, stream));
    unsigned int runs = 0;
    TRY_(hipMemcpyAsync(&runs, d_runs, 4, hipMemcpyDeviceToHost, stream));
    TRY_(hipStreamSynchronize(stream));
#undef TRY_
    *n_entries = runs;
    return done(PA_OK);
}

}  // namespace pa
