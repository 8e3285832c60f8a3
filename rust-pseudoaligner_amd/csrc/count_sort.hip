// The equivalence-class count table (the unit the GPUs of a node reduce over RCCL) from the KEY STREAMS of a
// mapping launch. The map kernel (map_pool.hip) appends one key per finished read — the slot of the table the read counts
// in — to its wave's stream; the streams are fully coalesced writes (0.4 GB per 100 M reads). Here:
//
//   one bin  (table of <= 32768 slots, e.g. gencode_small)     pa_keys_count_kernel straight over the streams
//   several  pa_keys_hist_kernel     keys per bin (bin = key >> 15: 32768 consecutive slots = 128 KiB of LDS counters)
//            pa_keys_scatter_kernel  keys partitioned by bin into `sorted` — as 16-bit keys, the position names the bin — (counting sort: a workgroup counts its tile's
//                                    keys per bin in LDS, reserves the runs with one global atomic per bin, writes every
//                                    key to its run; the lines of a run fill up inside the L2 within one tile)
//            pa_keys_count_kernel    one LDS table per workgroup and bin: LDS atomics over its share of the bin's keys,
//                                    then the non-zero counters are added to the caller's u64 table
//
// Round 2 counted inside the map kernel with one device-scope atomic per read into per-XCD replicas of the table; the table
// (1.9 MB per replica at config 3) does not survive in an L2 that 3 GB of dictionary lines, node blobs and read tiles stream
// through per launch, and a device-scope atomic that misses is forwarded to the memory side: 100 M random 32-byte requests
// per 100 M reads, 8-9 % of the kernel. The same counts by sorting move 4 x 0.4 GB of coalesced traffic.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "kernels.hpp"
#include "pa_common.hpp"

namespace pa {
namespace {

constexpr uint32_t NO_KEY = 0xFFFFFFFFu;
#ifndef PA_MAX_BINS
#define PA_MAX_BINS 256   // (-DPA_MAX_BINS=2: the test build in which a table of 100 k classes already is "beyond MAX_BINS", _build.build_maxbins_variant)
#endif
constexpr uint32_t MAX_BINS = PA_MAX_BINS;               // tables of up to 8.4 M slots; beyond, keys are counted with plain atomics
constexpr uint64_t PA_COUNT_DIRECT_MAX_READS = 1u << 16; // ... and so are the keys of batches this small
constexpr uint32_t BIN_SLOTS = 1u << PA_KEY_BIN_SHIFT;
constexpr uint32_t CS_BLOCK = 1024;

// The keys of a launch: what the waves of the map kernel appended (whole chunks of PA_KEY_CHUNK, *keys_top entries) and, right
// behind them in the same buffer, one key per deferred read from pa_resolve_kernel (resolve.hip: as many as the stream of deferred
// reads holds, a multiple of PA_DEFER_CHUNK; *extra_top, may be null). Both are multiples of four.
struct KeyTops {
    const unsigned long long* top;
    uint64_t cap;
    const unsigned long long* extra_top;
    uint64_t extra_cap;
};
__device__ __forceinline__ uint64_t stream_len(const KeyTops k, uint64_t unused = 0) {
    (void)unused;
    unsigned long long t = *k.top, x = k.extra_top ? *k.extra_top : 0ull;
    if (t > k.cap) t = k.cap;
    if (x > k.extra_cap) x = k.extra_cap;
    return t + x;
}

// Counting sort of the keys by bin without a single global atomic: the stream is cut into one slice per workgroup;
//   pa_keys_hist_kernel     wg_hist[b * G + g] = keys of bin b in slice g
//   pa_keys_scan_kernel     (one workgroup per bin) wg_base[b * G + g] = where slice g's keys of bin b go: the bin's base (the bins
//                           before it) + the slices before g; hist[b] = the bin's total
//   pa_keys_scatter_kernel  the same slices again: every workgroup partitions its tiles inside LDS and appends the runs at its own
//                           cursors (LDS), writing whole lines
// (One shared cursor per bin, bumped once per tile, is a hot word: 26 k dependent atomics per bin and launch = 0.3 ms.)
constexpr uint32_t SC_BLOCK = 256, SC_PER = 16, SC_TILE = SC_BLOCK * SC_PER;
constexpr uint32_t SC_MAX_GRID = 2048;   // slices (the scan kernel holds one bin's slice counts in LDS)

__device__ __forceinline__ void slice_of(uint64_t n, uint32_t g, uint32_t G, uint64_t& a, uint64_t& b) {   // whole tiles
    const uint64_t tiles = (n + SC_TILE - 1) / SC_TILE;
    a = tiles * g / G * SC_TILE;
    b = tiles * (g + 1) / G * SC_TILE;
    if (b > n) b = n;
    if (a > n) a = n;
}

__global__ __launch_bounds__(SC_BLOCK) void pa_keys_hist_kernel(const uint32_t* __restrict__ keys, const KeyTops keys_top, uint32_t nbins, uint32_t* __restrict__ wg_hist) {
    __shared__ uint32_t h[MAX_BINS];
    for (uint32_t i = threadIdx.x; i < nbins; i += SC_BLOCK) h[i] = 0;
    __syncthreads();
    uint64_t a, b;
    slice_of(stream_len(keys_top), blockIdx.x, gridDim.x, a, b);
    const uint4* k4 = reinterpret_cast<const uint4*>(keys);   // (chunks and tiles are multiples of 4 entries: 16-byte loads)
    for (uint64_t i = a / 4 + threadIdx.x; i < b / 4; i += SC_BLOCK) {
        const uint4 v = k4[i];
        if (v.x != NO_KEY) atomicAdd(&h[v.x >> PA_KEY_BIN_SHIFT], 1u);
        if (v.y != NO_KEY) atomicAdd(&h[v.y >> PA_KEY_BIN_SHIFT], 1u);
        if (v.z != NO_KEY) atomicAdd(&h[v.z >> PA_KEY_BIN_SHIFT], 1u);
        if (v.w != NO_KEY) atomicAdd(&h[v.w >> PA_KEY_BIN_SHIFT], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nbins; i += SC_BLOCK) wg_hist[(uint64_t)i * gridDim.x + blockIdx.x] = h[i];
}

// blockIdx.x = bin. hist[bin] = its total; wg_base[bin * G + g] = base of the bin + slices before g. The base of a bin needs
// the totals of the bins before it: every workgroup sums those rows itself (nbins * G words, L2-resident).
__global__ __launch_bounds__(1024) void pa_keys_scan_kernel(const uint32_t* __restrict__ wg_hist, uint32_t G, uint32_t* __restrict__ wg_base,
                                                            uint32_t* __restrict__ hist) {
    __shared__ uint32_t red[1024];
    const uint32_t bin = blockIdx.x, t = threadIdx.x;
    uint32_t before = 0;   // keys of the bins before this one
    for (uint64_t i = t; i < (uint64_t)bin * G; i += 1024) before += wg_hist[i];
    red[t] = before;
    __syncthreads();
    for (uint32_t o = 512; o; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); }
    const uint32_t bin_base = red[0];
    __syncthreads();
    // exclusive scan of this bin's G slice counts: two per thread
    const uint32_t i0 = 2 * t, i1 = 2 * t + 1;
    const uint32_t c0 = i0 < G ? wg_hist[(uint64_t)bin * G + i0] : 0u, c1 = i1 < G ? wg_hist[(uint64_t)bin * G + i1] : 0u;
    red[t] = c0 + c1;
    __syncthreads();
    for (uint32_t o = 1; o < 1024; o <<= 1) {   // Hillis-Steele inclusive scan over the pair sums
        const uint32_t x = t >= o ? red[t - o] : 0u;
        __syncthreads();
        red[t] += x;
        __syncthreads();
    }
    const uint32_t excl = red[t] - (c0 + c1);
    if (i0 < G) wg_base[(uint64_t)bin * G + i0] = bin_base + excl;
    if (i1 < G) wg_base[(uint64_t)bin * G + i1] = bin_base + excl + c0;
    if (t == 1023) hist[bin] = red[t];
}

__global__ __launch_bounds__(SC_BLOCK) void pa_keys_scatter_kernel(const uint32_t* __restrict__ keys, const KeyTops keys_top, uint32_t nbins, const uint32_t* __restrict__ wg_base,
                                                                   uint16_t* __restrict__ sorted) {
    __shared__ uint32_t cnt[MAX_BINS], lbase[MAX_BINS], gbase[MAX_BINS], cursor[MAX_BINS], stage[SC_TILE], total;
    for (uint32_t b = threadIdx.x; b < nbins; b += SC_BLOCK) cursor[b] = wg_base[(uint64_t)b * gridDim.x + blockIdx.x];
    uint64_t sa, sb;
    slice_of(stream_len(keys_top), blockIdx.x, gridDim.x, sa, sb);
    for (uint64_t t0 = sa; t0 < sb; t0 += SC_TILE) {
        for (uint32_t b = threadIdx.x; b < nbins; b += SC_BLOCK) cnt[b] = 0;
        __syncthreads();
        uint32_t k[SC_PER], pos[SC_PER];
        const uint4* src = reinterpret_cast<const uint4*>(keys + t0) + threadIdx.x;
#pragma unroll
        for (uint32_t j = 0; j < SC_PER / 4; ++j) {   // coalesced 16-byte loads, all in flight together
            const uint64_t i = t0 + 4ull * (j * SC_BLOCK + threadIdx.x);
            const uint4 v = i < sb ? src[j * SC_BLOCK] : uint4{NO_KEY, NO_KEY, NO_KEY, NO_KEY};
            k[4 * j] = v.x; k[4 * j + 1] = v.y; k[4 * j + 2] = v.z; k[4 * j + 3] = v.w;
        }
#pragma unroll
        for (uint32_t j = 0; j < SC_PER; ++j) pos[j] = k[j] != NO_KEY ? atomicAdd(&cnt[k[j] >> PA_KEY_BIN_SHIFT], 1u) : 0u;   // rank inside the tile's run of that bin
        __syncthreads();
        for (uint32_t b = threadIdx.x; b < nbins; b += SC_BLOCK) {
            uint32_t s = 0;
            for (uint32_t j = 0; j < b; ++j) s += cnt[j];
            lbase[b] = s;
            gbase[b] = cursor[b];
            cursor[b] += cnt[b];
            if (b == nbins - 1) total = s + cnt[b];
        }
        __syncthreads();
#pragma unroll
        for (uint32_t j = 0; j < SC_PER; ++j)
            if (k[j] != NO_KEY) stage[lbase[k[j] >> PA_KEY_BIN_SHIFT] + pos[j]] = k[j];
        __syncthreads();
        const uint32_t tot = total;
        for (uint32_t i = threadIdx.x; i < tot; i += SC_BLOCK) {   // consecutive threads, consecutive words of a run; the key names its bin
            const uint32_t key = stage[i], b = key >> PA_KEY_BIN_SHIFT;
            sorted[gbase[b] + (i - lbase[b])] = (uint16_t)(key & (BIN_SLOTS - 1));   // the position names the bin: 15 bits of the key are left
        }
        __syncthreads();
    }
}

// counts[(bin << 15) + i] += occurrences of key i in the bin's run of `sorted` (16-bit keys: what scatter left of them). Workgroup
// (bin, part) takes part `part` of `parts` of the run.
__global__ __launch_bounds__(CS_BLOCK) void pa_keys_count_kernel(const uint16_t* __restrict__ src, const uint32_t* __restrict__ hist, uint32_t parts,
                                                                 unsigned long long* __restrict__ counts, uint64_t counts_len) {
    extern __shared__ uint32_t tab[];   // BIN_SLOTS counters
    const uint32_t bin = blockIdx.x / parts, part = blockIdx.x % parts;
    const uint64_t slot0 = (uint64_t)bin << PA_KEY_BIN_SHIFT;
    const uint32_t nslots = (uint32_t)(counts_len - slot0 < BIN_SLOTS ? counts_len - slot0 : BIN_SLOTS);
    for (uint32_t i = threadIdx.x; i < nslots; i += CS_BLOCK) tab[i] = 0;
    __syncthreads();
    uint64_t lo = 0;
    for (uint32_t j = 0; j < bin; ++j) lo += hist[j];
    const uint64_t n = hist[bin];
    const uint64_t a = lo + n * part / parts, b = lo + n * (part + 1) / parts;
    // 16-byte loads (eight keys), four in flight per thread; the few keys before the first and after the last aligned octet one by one
    const uint64_t a8 = (a + 7) & ~7ull, b8 = b & ~7ull;
    if (a8 >= b8) {
        for (uint64_t i = a + threadIdx.x; i < b; i += CS_BLOCK) atomicAdd(&tab[src[i]], 1u);
    } else {
        if (a + threadIdx.x < a8) atomicAdd(&tab[src[a + threadIdx.x]], 1u);
        if (b8 + threadIdx.x < b) atomicAdd(&tab[src[b8 + threadIdx.x]], 1u);
        const uint4* q = reinterpret_cast<const uint4*>(src);
        constexpr uint32_t DEPTH = 4;
        for (uint64_t i0 = a8 / 8; i0 < b8 / 8; i0 += (uint64_t)DEPTH * CS_BLOCK) {
            uint4 v[DEPTH];
            bool in[DEPTH];
#pragma unroll
            for (uint32_t j = 0; j < DEPTH; ++j) {
                const uint64_t i = i0 + (uint64_t)j * CS_BLOCK + threadIdx.x;
                in[j] = i < b8 / 8;
                v[j] = in[j] ? q[i] : uint4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (uint32_t j = 0; j < DEPTH; ++j)
                if (in[j]) {
                    atomicAdd(&tab[v[j].x & 0xFFFFu], 1u); atomicAdd(&tab[v[j].x >> 16], 1u);
                    atomicAdd(&tab[v[j].y & 0xFFFFu], 1u); atomicAdd(&tab[v[j].y >> 16], 1u);
                    atomicAdd(&tab[v[j].z & 0xFFFFu], 1u); atomicAdd(&tab[v[j].z >> 16], 1u);
                    atomicAdd(&tab[v[j].w & 0xFFFFu], 1u); atomicAdd(&tab[v[j].w >> 16], 1u);
                }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nslots; i += CS_BLOCK) {
        const uint32_t c = tab[i];
        if (c) atomicAdd(counts + slot0 + i, (unsigned long long)c);   // (launches on several streams may count into one table)
    }
}

// a table of one bin: straight over the raw streams (32-bit keys, padding skipped); workgroup `blockIdx.x` of `gridDim.x` takes its share
__global__ __launch_bounds__(CS_BLOCK) void pa_keys_count_raw_kernel(const uint32_t* __restrict__ src, const KeyTops keys_top, unsigned long long* __restrict__ counts,
                                                                     uint64_t counts_len) {
    extern __shared__ uint32_t tab[];   // counts_len counters
    for (uint32_t i = threadIdx.x; i < counts_len; i += CS_BLOCK) tab[i] = 0;
    __syncthreads();
    const uint64_t n4 = stream_len(keys_top) / 4;
    const uint64_t a = n4 * blockIdx.x / gridDim.x, b = n4 * (blockIdx.x + 1) / gridDim.x;
    const uint4* q = reinterpret_cast<const uint4*>(src);
    for (uint64_t i = a + threadIdx.x; i < b; i += CS_BLOCK) {
        const uint4 v = q[i];
        if (v.x != NO_KEY) atomicAdd(&tab[v.x], 1u);
        if (v.y != NO_KEY) atomicAdd(&tab[v.y], 1u);
        if (v.z != NO_KEY) atomicAdd(&tab[v.z], 1u);
        if (v.w != NO_KEY) atomicAdd(&tab[v.w], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < counts_len; i += CS_BLOCK) {
        const uint32_t c = tab[i];
        if (c) atomicAdd(counts + i, (unsigned long long)c);
    }
}

// tables beyond MAX_BINS bins: plain atomics per key
__global__ __launch_bounds__(256) void pa_keys_count_direct_kernel(const uint32_t* __restrict__ keys, const KeyTops keys_top, unsigned long long* __restrict__ counts) {
    const uint64_t n = stream_len(keys_top);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t k = keys[i];
        if (k != NO_KEY) atomicAdd(counts + k, 1ull);
    }
}


}  // namespace
// leaves one chunk partly used at its exit
uint64_t key_stream_capacity(uint64_t n_reads, uint32_t nwaves) {   // chunks are filled to the last entry; every wave leaves one partly used (padded)
    return (n_reads / PA_KEY_CHUNK + nwaves + 2) * PA_KEY_CHUNK;
}

size_t count_keys_ctl_bytes(uint64_t counts_len) {
    const uint64_t nbins = (counts_len + BIN_SLOTS - 1) >> PA_KEY_BIN_SHIFT;
    return (MAX_BINS + 2 * (size_t)std::min<uint64_t>(nbins, MAX_BINS) * SC_MAX_GRID) * 4;
}

int launch_count_keys(const uint32_t* keys, const unsigned long long* keys_top_ptr, uint64_t keys_cap, const unsigned long long* extra_top, uint64_t extra_cap,
                      uint32_t* sorted, uint32_t* ctl, unsigned long long* counts, uint64_t counts_len, int num_cus, hipStream_t stream, uint64_t n_reads) {
    if (counts_len == 0) return 0;
    const KeyTops keys_top{keys_top_ptr, keys_cap, extra_top, extra_cap};
    const uint64_t nbins = (counts_len + BIN_SLOTS - 1) >> PA_KEY_BIN_SHIFT;
    const uint32_t cus = num_cus > 0 ? (uint32_t)num_cus : 256u;
    const size_t lds = (size_t)(counts_len < BIN_SLOTS ? counts_len : BIN_SLOTS) * 4;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pa_keys_count_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pa_keys_count_raw_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    // plain atomics per key: tables beyond MAX_BINS bins — and SMALL batches of a multi-bin table, where four kernels over a few
    // thousand keys cost more than the atomics they avoid (one launch instead of four; every small-batch test runs this kernel)
    if (nbins > MAX_BINS || (nbins > 1 && n_reads <= PA_COUNT_DIRECT_MAX_READS)) {
        const uint32_t blocks = (uint32_t)std::min<uint64_t>((uint64_t)cus * 8, (keys_cap + extra_cap + 255) / 256 + 1);
        hipLaunchKernelGGL(pa_keys_count_direct_kernel, dim3(blocks), dim3(256), 0, stream, keys, keys_top, counts);
        return (int)hipGetLastError();
    }
    if (nbins == 1) {
        hipLaunchKernelGGL(pa_keys_count_raw_kernel, dim3(cus), dim3(CS_BLOCK), lds, stream, keys, keys_top, counts, counts_len);
        return (int)hipGetLastError();
    }
    // ctl: hist[MAX_BINS] | wg_hist[nbins * G] | wg_base[nbins * G]
    const uint32_t G = std::min<uint32_t>(SC_MAX_GRID, cus * 8);
    uint32_t* hist = ctl;
    uint32_t* wg_hist = ctl + MAX_BINS;
    uint32_t* wg_base = wg_hist + (size_t)nbins * G;
    hipLaunchKernelGGL(pa_keys_hist_kernel, dim3(G), dim3(SC_BLOCK), 0, stream, keys, keys_top, (uint32_t)nbins, wg_hist);
    hipLaunchKernelGGL(pa_keys_scan_kernel, dim3((uint32_t)nbins), dim3(1024), 0, stream, (const uint32_t*)wg_hist, G, wg_base, hist);
    hipLaunchKernelGGL(pa_keys_scatter_kernel, dim3(G), dim3(SC_BLOCK), 0, stream, keys, keys_top, (uint32_t)nbins, (const uint32_t*)wg_base, reinterpret_cast<uint16_t*>(sorted));
    // one workgroup per CU at most (an LDS table of 128 KiB each) and ONE round of them: 270 workgroups on 256 CUs take as long as 512
    uint32_t parts = std::max<uint32_t>(1u, (uint32_t)(cus / nbins));
    parts = std::max<uint32_t>(1u, (uint32_t)knob_int("PA_COUNT_PARTS", (int)parts));   // (A/B knob; never zero: the count kernel divides by it)
    hipLaunchKernelGGL(pa_keys_count_kernel, dim3((uint32_t)nbins * parts), dim3(CS_BLOCK), lds, stream, reinterpret_cast<const uint16_t*>(sorted), (const uint32_t*)hist, parts, counts,
                       counts_len);
    return (int)hipGetLastError();
}

}  // namespace pa
