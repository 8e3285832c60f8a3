// HIP kernels for gfx950 (CDNA4, wave64). Hand-written for this target only.
//
//  pa_map_kernel       the hot path: map_read_with_mismatch (src/pseudoaligner.rs:361-376) for a batch of reads
//  pa_encode_kernel    DnaString::from_dna_string (src/pseudoaligner.rs:449-450): ASCII -> 2-bit tiles
//  pa_simulate_kernel  synthetic reads (bench / tests), same function as synth.cpp
//  pa_count_kernel     equivalence-class count table
//
// pa_map_kernel design:
//   * one lane = one read; a wave owns a contiguous range of 64-read tiles and keeps its 64 lanes busy by REFILLING
//     finished lanes from its next tile (reads per node visit are heavy-tailed, so lock-step tiles would idle).
//   * per-lane state machine (lane_steps.hpp): SEEK (one 64-byte dictionary bucket), FWD / LEFT (one node blob),
//     FINISH (class intersection + output). Each wave iteration runs ONE state for all lanes that are in it — the
//     state with the most lanes — so every executed instruction serves many lanes and each iteration has exactly one
//     dependent HBM/L2 round trip; other waves of the CU hide it.
//   * LDS per wave: the 64 packed reads, word-major ([word][lane], bank = lane, conflict-free for any per-lane word
//     index) and the per-lane list of distinct colours seen; no cross-wave communication, no barriers.
//   * outputs: 16-byte record per read (input order) + class ids in an arena; a wave carves its ids out of a private
//     chunk and only touches the global bump pointer once per chunk (a single hot atomic word saturates at
//     ~88 M ops/s on this chip, far below the read rate).
#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "lane_steps.hpp"
#include "synth_common.hpp"

namespace pa {

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(v, d, 64);
        if ((int)lane >= d) v += t;
    }
    return v;
}

// hash of a sorted id list; must equal list_hash_host (device_index.hip)
__device__ __forceinline__ uint64_t list_hash_dev(const uint32_t* v, uint32_t n) {
    uint64_t h = 0x243f6a8885a308d3ull ^ n;
    for (uint32_t i = 0; i < n; ++i) h = pa_mix64(h ^ v[i]) + 0x9e3779b97f4a7c15ull;
    return h;
}

// index class whose id list equals v[0..n), or 0xFFFFFFFF (content lookup in the class-list hash table)
__device__ __forceinline__ uint32_t class_of_list(const uint32_t* v, uint32_t n, const uint32_t* ec_off, const uint32_t* ec_ids,
                                                  const uint32_t* class_table, uint64_t class_table_size) {
    uint64_t j = list_hash_dev(v, n) % class_table_size;
    for (;;) {
        const uint32_t cand = class_table[j];
        if (cand == 0xFFFFFFFFu) return cand;
        const uint32_t st = ec_off[cand], ln = ec_off[cand + 1] - st;
        if (ln == n) {
            bool eq = true;
            for (uint32_t t = 0; t < ln && eq; ++t) eq = ec_ids[st + t] == v[t];
            if (eq) return cand;
        }
        if (++j == class_table_size) j = 0;
    }
}

template <bool TRACE>
__global__ __launch_bounds__(PA_MAP_BLOCK) void pa_map_kernel(const MapParams p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t lane = lane_id();
    const uint32_t wave_in_block = threadIdx.x >> 6;
    const uint32_t waves_per_block = PA_MAP_BLOCK / 64;
    const uint64_t wave = (uint64_t)blockIdx.x * waves_per_block + wave_in_block;
    const uint64_t nwaves = (uint64_t)gridDim.x * waves_per_block;

    const uint32_t wave_bytes = (p.wpr + 1) * 512 + p.col_cap * 256;
    uint64_t* rd_base = reinterpret_cast<uint64_t*>(smem + wave_in_block * wave_bytes);
    uint32_t* col_base = reinterpret_cast<uint32_t*>(smem + wave_in_block * wave_bytes + (p.wpr + 1) * 512);
    const ReadRef rd{rd_base + lane, 64};
    const ColRef cols{col_base + lane, 64, p.col_cap, p.spill + ((uint64_t)wave * 64 + lane) * p.spill_cap, p.spill_cap,
                      TRACE ? p.trace + ((uint64_t)wave * 64 + lane) * p.spill_cap : nullptr};

    // static partition of the tiles over the waves (no global work queue: one hot atomic would cap the rate)
    const uint64_t ntiles = (p.n_reads + 63) >> 6;
    uint64_t next = (ntiles * wave / nwaves) << 6;
    uint64_t end = (ntiles * (wave + 1) / nwaves) << 6;
    if (end > p.n_reads) end = p.n_reads;
    if (next > end) next = end;

    Lane s;
    s.st = ST_EMPTY;
    s.rid = s.L = s.kp = s.cov = s.mism = s.h = s.off = s.ro = s.rem = s.snp = s.ra = s.ph = s.ncol = s.flags = s.ntrace = 0;
    uint64_t chunk_cur = 0, chunk_end = 0;   // wave-uniform: private slice of the class arena

    for (;;) {
        const uint64_t mE = __ballot(s.st == ST_EMPTY);
        const uint64_t mS = __ballot(s.st == ST_SEEK);
        const uint64_t mF = __ballot(s.st == ST_FWD);
        const uint64_t mL = __ballot(s.st == ST_LEFT);
        const uint32_t nS = __popcll(mS), nF = __popcll(mF), nL = __popcll(mL);
        const uint32_t nFin = 64 - __popcll(mE) - nS - nF - nL;
        const uint64_t left = end - next;
        const uint32_t nR = (uint32_t)(left < (uint64_t)__popcll(mE) ? left : (uint64_t)__popcll(mE));
        // pick the state with the most lanes (ties: refill, seek, fwd, finish, left)
        uint32_t best = nR, sel = 0;
        if (nS > best) { best = nS; sel = 1; }
        if (nF > best) { best = nF; sel = 2; }
        if (nFin > best) { best = nFin; sel = 3; }
        if (nL > best) { best = nL; sel = 4; }
        if (best == 0) break;

        if (sel == 0) {   // ---- REFILL: empty lanes take the next reads of this wave's range (coalesced by rank)
            const uint32_t rank = __popcll(mE & ((1ull << lane) - 1));
            if (s.st == ST_EMPTY && rank < nR) {
                const uint64_t rid = next + rank;
                uint32_t L = p.lens[rid];
                if (L > p.wpr * 32) L = p.wpr * 32;
                const uint64_t* src = p.tiles + ((rid >> 6) * p.wpr) * 64 + (rid & 63);
                for (uint32_t w = 0; w < p.wpr; ++w) rd_base[w * 64 + lane] = src[(uint64_t)w * 64];
                rd_base[p.wpr * 64 + lane] = 0;
                lane_start(s, (uint32_t)rid, L, p.ix.k);
            }
            next += nR;
        } else if (sel == 1) {
            if (s.st == ST_SEEK) seek_step(s, p.ix, rd);
        } else if (sel == 2) {
            if (s.st == ST_FWD) fwd_step<TRACE>(s, p.ix, rd, cols, p.allowed);
        } else if (sel == 4) {
            if (s.st == ST_LEFT) left_step<TRACE>(s, p.ix, rd, cols, p.allowed);
        } else {          // ---- FINISH: nodes_to_eq_class + output
            const bool fin = s.st == ST_ISECT || s.st == ST_NONE;
            Isect is{0, 0, 0, 0, 0};
            if (s.st == ST_ISECT) is = isect_count(s, p.ix, cols);
            const uint32_t cnt = fin ? is.count : 0;
            const uint32_t incl = wave_incl_scan(cnt, lane);
            const uint32_t total = __shfl(incl, 63, 64);
            if (total > 0) {
                if (chunk_cur + total > chunk_end) {   // wave-uniform branch
                    const uint64_t want = total > PA_ARENA_CHUNK ? total : PA_ARENA_CHUNK;
                    unsigned long long base = 0;
                    if (lane == 0) base = atomicAdd(p.arena_top, (unsigned long long)want);
                    base = __shfl(base, 0, 64);
                    chunk_cur = base;
                    chunk_end = base + want;
                }
            }
            const uint64_t my_off = chunk_cur + (incl - cnt);
            chunk_cur += total;
            if (fin) {
                pa_read_result r{0, 0, 0, 0};
                uint32_t colour = 0xFFFFFFFFu;
                if (s.st == ST_ISECT) {
                    r.coverage = s.cov;
                    r.mismatches = s.mism | PA_MAPPED_BIT;
                    r.class_len = cnt;
                    r.class_off = (uint32_t)my_off;
                    if (cnt) {
                        if (my_off + cnt <= p.arena_cap) isect_write(s, p.ix, cols, is, p.arena + my_off);
                        else atomicOr(p.status, PA_STATUS_ARENA_FULL);
                    }
                    if (cnt == is.base_len) colour = is.base_colour;
                    if (s.flags & F_SPILL_OVERFLOW) atomicOr(p.status, PA_STATUS_SPILL_OVERFLOW);
                }
                reinterpret_cast<U4*>(p.results)[s.rid] = U4{r.coverage, r.mismatches, r.class_off, r.class_len};
                if (p.colour_out) p.colour_out[s.rid] = colour;
                if (p.counts) {   // fused class-count table: fire-and-forget atomics overlap the other lanes' walks
                    uint32_t slot = p.ix.num_classes + 2;                      // unmapped
                    if (s.st == ST_ISECT) {
                        if (cnt == 0) slot = p.ix.num_classes + 1;             // mapped, empty class
                        else {
                            if (colour == 0xFFFFFFFFu && my_off + cnt <= p.arena_cap)   // strict subset of every visited class
                                colour = class_of_list(p.arena + my_off, cnt, p.ix.ec_off, p.ix.ec_ids, p.class_table, p.class_table_size);
                            slot = colour == 0xFFFFFFFFu ? p.ix.num_classes : colour;
                        }
                    }
                    atomicAdd(p.counts + slot, 1ull);
                }
                if (TRACE) {   // node lists, read-major, stride spill_cap (map_read_to_nodes test surface)
                    const uint32_t nn = s.st == ST_ISECT ? (s.ntrace < p.spill_cap ? s.ntrace : p.spill_cap) : 0;
                    p.nodes_len[s.rid] = s.st == ST_ISECT ? s.ntrace : 0;
                    for (uint32_t j = 0; j < nn; ++j) p.nodes_out[(uint64_t)s.rid * p.spill_cap + j] = cols.trace[j];
                }
                s.st = ST_EMPTY;
            }
        }
    }
}

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
    uint64_t words[PA_MAX_READ_LEN / 32 + 1];
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
                                                       const uint32_t* __restrict__ colour, uint64_t n_reads,
                                                       const uint32_t* __restrict__ ec_off, const uint32_t* __restrict__ ec_ids,
                                                       const uint32_t* __restrict__ class_table, uint64_t class_table_size,
                                                       uint32_t num_classes, unsigned long long* __restrict__ counts) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_reads) return;
    const pa_read_result r = results[i];
    uint32_t slot;
    if (!(r.mismatches & PA_MAPPED_BIT)) slot = num_classes + 2;
    else if (r.class_len == 0) slot = num_classes + 1;
    else {
        uint32_t c = colour ? colour[i] : 0xFFFFFFFFu;
        if (c == 0xFFFFFFFFu) c = class_of_list(arena + r.class_off, r.class_len, ec_off, ec_ids, class_table, class_table_size);
        slot = c == 0xFFFFFFFFu ? num_classes : c;
    }
    atomicAdd(counts + slot, 1ull);
}

// ---------------------------------------------------------------------------------------------- launchers
int launch_map(const MapParams& p, uint32_t grid, size_t lds_bytes, hipStream_t stream) {
    if (lds_bytes > 48 * 1024) {   // long-read tiles: opt in to more than the default dynamic LDS limit
        const void* fn = p.trace ? reinterpret_cast<const void*>(&pa_map_kernel<true>) : reinterpret_cast<const void*>(&pa_map_kernel<false>);
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    if (p.trace) hipLaunchKernelGGL(pa_map_kernel<true>, dim3(grid), dim3(PA_MAP_BLOCK), lds_bytes, stream, p);
    else hipLaunchKernelGGL(pa_map_kernel<false>, dim3(grid), dim3(PA_MAP_BLOCK), lds_bytes, stream, p);
    return (int)hipGetLastError();
}

int map_kernel_occupancy(size_t lds_bytes, int* blocks_per_cu) {
    return (int)hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, pa_map_kernel<false>, PA_MAP_BLOCK, lds_bytes);
}

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

int launch_count(const pa_read_result* results, const uint32_t* arena, const uint32_t* colour, uint64_t n, const uint32_t* ec_off,
                 const uint32_t* ec_ids, const uint32_t* class_table, uint64_t class_table_size, uint32_t num_classes,
                 unsigned long long* counts, hipStream_t stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(pa_count_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, results, arena, colour, n, ec_off, ec_ids,
                       class_table, class_table_size, num_classes, counts);
    return (int)hipGetLastError();
}

}  // namespace pa
