// HIP kernels for gfx950 (CDNA4, wave64) around the mapping kernel (map_pool.hip). Hand-written for this target only.
//
//  pa_encode_kernel    DnaString::from_dna_string (src/pseudoaligner.rs:449-450): ASCII -> 2-bit tiles
//  pa_simulate_kernel  synthetic reads (bench / tests), same function as synth.cpp
//  pa_count_kernel     equivalence-class count table from stored results (the map kernel normally fuses this)
#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "lane_steps.hpp"
#include "kernel_utils.hpp"
#include "synth_common.hpp"

namespace pa {

// ---------------------------------------------------------------------------------------------- encode
// One thread = one 64-bit word (32 bases) of one read. Non-ACGT bytes encode as A (what DnaString::from_dna_string is
// understood to do — unpinned), either case.
__global__ __launch_bounds__(256) void pa_encode_kernel(const uint8_t* __restrict__ ascii, const uint64_t* __restrict__ offsets,
                                                        uint64_t n_reads, uint32_t wpr, uint64_t* __restrict__ tiles,
                                                        uint32_t* __restrict__ lens) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t ntiles = (n_reads + 63) >> 6;
    if (gid >= ntiles * wpr * 64) return;
    const uint32_t r = (uint32_t)(gid & 63);
    const uint64_t tw = gid >> 6;
    const uint32_t w = (uint32_t)(tw % wpr);
    const uint64_t rid = (tw / wpr) * 64 + r;
    uint64_t v = 0;
    if (rid < n_reads) {
        const uint64_t o = offsets[rid];
        uint64_t len = offsets[rid + 1] - o;
        if (len > (uint64_t)wpr * 32) len = (uint64_t)wpr * 32;
        if (w == 0) lens[rid] = (uint32_t)len;
        const uint64_t b0 = 32ull * w;
        const uint32_t nb = len > b0 ? (uint32_t)(len - b0 < 32 ? len - b0 : 32) : 0;
        const uint8_t* src = ascii + o + b0;
        for (uint32_t j = 0; j < nb; ++j) {
            const uint8_t c = src[j] & 0xDF;   // upper-case
            const uint64_t code = c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 0u;
            v |= code << (2 * j);
        }
    }
    tiles[gid] = v;
}

// ---------------------------------------------------------------------------------------------- simulate
__global__ __launch_bounds__(256) void pa_simulate_kernel(const uint64_t* __restrict__ packed, const uint64_t* __restrict__ tx_start,
                                                          const uint64_t* __restrict__ cum, uint32_t num_tx, uint64_t total,
                                                          uint32_t read_len, uint64_t seed, uint32_t ppm, uint64_t first_read,
                                                          uint64_t n_reads, uint32_t wpr, uint64_t* __restrict__ tiles,
                                                          uint32_t* __restrict__ lens) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t ntiles = (n_reads + 63) >> 6;
    if (i >= ntiles * 64) return;
    uint64_t words[PA_MAX_SIM_READ_LEN / 32 + 1];
    const uint32_t nw = (read_len + 31) / 32;
    uint64_t* dst = tiles + ((i >> 6) * wpr) * 64 + (i & 63);
    if (i < n_reads) {
        synth::simulate_read(packed, tx_start, cum, num_tx, total, read_len, seed, ppm, first_read + i, words);
        lens[i] = read_len;
        for (uint32_t w = 0; w < wpr; ++w) dst[(uint64_t)w * 64] = w < nw ? words[w] : 0;
    } else {
        for (uint32_t w = 0; w < wpr; ++w) dst[(uint64_t)w * 64] = 0;
    }
}

// ---------------------------------------------------------------------------------------------- counts
// counts[c] for reads whose class is index class c; [nc] novel non-empty, [nc+1] mapped-but-empty, [nc+2] unmapped.
// A result that is a strict subset of every visited class is looked up by content in the class-list hash table.
__global__ __launch_bounds__(256) void pa_count_kernel(const pa_read_result* __restrict__ results, const uint32_t* __restrict__ arena,
                                                       const uint32_t* __restrict__ colour, uint64_t n_reads, const DevIndexView ix,
                                                       const uint32_t* __restrict__ class_table, uint64_t class_table_size,
                                                       unsigned long long* __restrict__ counts) {
    const uint32_t num_classes = ix.num_classes;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_reads) return;
    const pa_read_result r = results[i];
    uint32_t slot;
    if (!(r.mismatches & PA_MAPPED_BIT)) slot = num_classes + 2;
    else if (r.class_len == 0) slot = num_classes + 1;
    else {
        uint32_t c = (r.class_off & PA_CLASS_REF) ? (r.class_off & ~PA_CLASS_REF) : colour ? colour[i] : 0xFFFFFFFFu;
        if (c == 0xFFFFFFFFu) c = class_of_list(arena + r.class_off, r.class_len, ix, class_table, class_table_size);
        slot = c == 0xFFFFFFFFu ? num_classes : c;
    }
    atomicAdd(counts + slot, 1ull);
}

// ---------------------------------------------------------------------------------------------- launchers
int launch_encode(const uint8_t* ascii, const uint64_t* offsets, uint64_t n, uint32_t wpr, uint64_t* tiles, uint32_t* lens,
                  hipStream_t stream) {
    const uint64_t threads = ((n + 63) >> 6) * wpr * 64;
    if (threads == 0) return 0;
    hipLaunchKernelGGL(pa_encode_kernel, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, stream, ascii, offsets, n, wpr, tiles, lens);
    return (int)hipGetLastError();
}

int launch_simulate(const uint64_t* packed, const uint64_t* tx_start, const uint64_t* cum, uint32_t num_tx, uint64_t total,
                    uint32_t read_len, uint64_t seed, uint32_t ppm, uint64_t first_read, uint64_t n, uint32_t wpr, uint64_t* tiles,
                    uint32_t* lens, hipStream_t stream) {
    const uint64_t threads = ((n + 63) >> 6) * 64;
    if (threads == 0) return 0;
    hipLaunchKernelGGL(pa_simulate_kernel, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, stream, packed, tx_start, cum, num_tx,
                       total, read_len, seed, ppm, first_read, n, wpr, tiles, lens);
    return (int)hipGetLastError();
}

int launch_count(const pa_read_result* results, const uint32_t* arena, const uint32_t* colour, uint64_t n, const DevIndexView& ix,
                 const uint32_t* class_table, uint64_t class_table_size, unsigned long long* counts, hipStream_t stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(pa_count_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, results, arena, colour, n, ix,
                       class_table, class_table_size, counts);
    return (int)hipGetLastError();
}

}  // namespace pa
