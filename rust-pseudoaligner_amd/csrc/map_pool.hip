// pa_map_pool_kernel — the hot path (map_read_with_mismatch, src/pseudoaligner.rs:361-376) with POOLED scheduling.
//
// The per-read state machine of lane_steps.hpp has data-dependent length (node visits per read are heavy-tailed) and
// several kinds of step (dictionary probe, node visit, left extension, class intersection, output). Running it with one
// fixed read per lane leaves most lanes idle in every step: the lanes of a wave are never all in the same state.
// Here a wave owns a POOL of S read slots (S > 64: 128 at 150 bp) that lives in LDS — packed read, 32-byte lane state,
// class windows — and ONE state byte per slot. Each iteration the wave
//     1. counts the slots in every state (ballots over the state bytes: lane i watches slots i and i + 64), picks a state
//        (one that fills a wave of 64, preferring the states nearest to completion; otherwise the most populated one),
//        compacts the first 64 slots in that state into its lanes and loads their lane state from LDS,
//     2. runs that state's step for all of them (one dependent HBM round trip; every lane does the same thing),
//     3. stores the lane state back and writes every slot's new state byte.
// Rare states simply wait until enough slots have gathered in them, so they are executed at full width too.
// Nothing is shared between waves: no locks, no barriers, no atomics besides the arena chunk grab and the count table.
//
// LDS per wave (S slots, wpr words per read):   [960 B fixed: arena chunk, key chunk, statistics, state bytes, pop list |
//                                                 rd u64[wpr][S] | st {u32 x 8}[S] | win {u32 x 4}[S] | {class id, read id}[S]]
// HBM per slot: a row of spill_cap u32 that holds the class lists of a read in list mode (lane_steps.hpp, ColRef) and,
// in TRACE builds, a second row with the visited node ids.
#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "lane_steps.hpp"
#include "kernel_utils.hpp"

#ifndef PA_FILL   // A/B builds: -DPA_FILL=0 (output steps do not take EMPTY slots along)
#define PA_FILL 1
#endif
#ifndef PA_COOP_MIN
#define PA_COOP_MIN 2u
#endif
#ifndef PA_REFILL_ALIGN   // A/B builds: -DPA_REFILL_ALIGN=16 (refills take whole 128-byte lines of the read tiles; measured and not kept: see the output step)
#define PA_REFILL_ALIGN 1u
#endif
#ifndef PA_RARE_MIN   // A/B builds: -DPA_RARE_MIN=0 (rare states compete by population only)
#define PA_RARE_MIN 16u   // (round 5, same-box A/B of 10 / 16 / 24 on the chain-block layout: config 5 -2.1 % time at 16, config 2 -1.4 %, config 3 -0.5 %; 24 is slower than 10)
#endif

// -DPA_ISA_MARKS: comments in the ISA listing that tools/isa_sections.py counts instructions between (static cost of the
// sections of an iteration); never defined in a build that runs
#ifdef PA_ISA_MARKS
#define PA_MARK(name) asm volatile("; PA_MARK " name ::: "memory")
#else
#define PA_MARK(name)
#endif

namespace pa {
namespace {

typedef __attribute__((address_space(3))) uint64_t* lds_u64;
typedef __attribute__((address_space(3))) uint32_t* lds_u32;
typedef __attribute__((address_space(3))) uint8_t* lds_u8;
typedef __attribute__((address_space(3))) uint16_t* lds_u16;
typedef __attribute__((address_space(3))) unsigned long long* lds_u64w;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4* lds_v4;
typedef __attribute__((address_space(1))) u32x4* glb_v4w;
typedef __attribute__((address_space(1))) const u32x4* glb_v4;
typedef __attribute__((address_space(1))) uint32_t* glb_u32w;
typedef __attribute__((address_space(1))) const uint32_t* glb_u32;
typedef __attribute__((address_space(1))) unsigned long long* glb_u64w;
// the 16-byte record of the common output step (PA_NT bit 1: written once, read by nobody on the device)
__device__ __forceinline__ void store_result(glb_v4w dst, u32x4 v) {
    if (PA_NT & 2) __builtin_nontemporal_store(v, dst);
    else *dst = v;
}

typedef __attribute__((address_space(4))) const MapParams* karg_ptr;   // the kernel's parameter block in the kernarg segment

__device__ __forceinline__ DevIndexView view_of(karg_ptr p) {   // member-wise: only the fields a step uses are actually loaded
    DevIndexView v;
    v.table = p->ix.table; v.nbuckets = p->ix.nbuckets; v.blobs = p->ix.blobs; v.ledge = p->ix.ledge;
    v.seg_g = p->ix.seg_g; v.seg_nid = p->ix.seg_nid; v.ec = p->ix.ec; v.class_ref = p->ix.class_ref; v.class_len = p->ix.class_len;
    v.wtable = p->ix.wtable; v.wbuckets = p->ix.wbuckets; v.kmask = p->ix.kmask; v.kmask_hi = p->ix.kmask_hi; v.k = p->ix.k; v.num_nodes = p->ix.num_nodes; v.num_classes = p->ix.num_classes; v.num_segs = p->ix.num_segs; v.stream_nt = p->ix.stream_nt; v.bitmap_min = p->ix.bitmap_min; v.bitmap_words = p->ix.bitmap_words;
    return v;
}

constexpr uint32_t ST_NSTAT = ST_COUNT + 4;   // statistics entries: one per state, the dual (forward + probe) iterations, and the plain forward step
                                               // split into issue / wait / compute (PA_MAP_STATS only)
constexpr uint32_t ST_DUAL = ST_COUNT;
constexpr uint32_t POOL_FIXED = 960;   // per wave: arena chunk {cur, end} (16 B), statistics, key chunk {cur, end} (8 B at +256), state bytes (128 B), pop list (64 B)
constexpr uint32_t POOL_MAX_SLOTS = 128;   // every lane watches the state bytes of two slots (lane, lane + 64)
constexpr uint32_t LIST_ROW_HDR = 12;  // list mode row: refs[4], lens[4], cids[4], then (ref, len, class id, -) quads

__device__ __forceinline__ uint32_t rank_in(uint64_t m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }

__device__ __forceinline__ uint64_t shfl64(uint64_t v, uint32_t src) {
    return ((uint64_t)(uint32_t)__shfl((int)(v >> 32), (int)src, 64) << 32) | (uint32_t)__shfl((int)v, (int)src, 64);
}

// wave-uniform: reserve cnt_alloc arena entries per lane out of the wave's private chunk; returns this lane's offset
__device__ __forceinline__ uint64_t arena_alloc(uint32_t cnt_alloc, uint32_t lane, karg_ptr p, lds_u64w chunk) {
    const uint32_t incl = wave_incl_scan(cnt_alloc);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    // LDS that one lane writes and other lanes read later: the compiler reasons per thread and would otherwise reuse a value
    // this lane loaded before another lane's store ("memory" = reload)
    asm volatile("" ::: "memory");
    unsigned long long chunk_cur = chunk[0];
    if (total > 0) {
        if (chunk_cur + total > chunk[1]) {   // take a new private slice of the class arena (one global atomic per chunk)
            const unsigned long long want = total > PA_ARENA_CHUNK ? total : PA_ARENA_CHUNK;
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(p->arena_top, want);
            base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            chunk_cur = base;
            if (lane == 0) chunk[1] = base + want;
        }
        if (lane == 0) chunk[0] = chunk_cur + total;
    }
    asm volatile("" ::: "memory");
    return chunk_cur + (incl - cnt_alloc);
}

// The class-count table is NOT updated here. Every finished read has one KEY — the slot of the table it counts in (class id,
// or the novel / empty / unmapped slot at the table's end) — and the wave appends the keys of a step to a stream of its own in
// HBM: private chunks of PA_KEY_CHUNK entries (one global atomic per chunk), the lanes of a step write consecutive entries
// (one or two lines per step). count_sort.hip turns the streams into the table afterwards (partition by key range, count in
// LDS). Round 2 added into per-XCD replicas of the table with one atomic per read: 100 M random 32-byte requests forwarded to
// the memory side per 100 M reads — 3 GB of write traffic and 8-9 % of the kernel; the streams are 0.4 GB, fully coalesced.
constexpr uint32_t NO_KEY = 0xFFFFFFFFu;   // padding of a chunk's unused tail (count_sort.hip skips it)
__device__ __forceinline__ void append_keys(uint32_t key, uint32_t lane, karg_ptr p, lds_u32 kchunk) {
    const uint64_t m = __ballot(key != NO_KEY);
    if (m == 0) return;
    const uint32_t cnt = (uint32_t)__popcll(m);
    const glb_u32w keys = (glb_u32w)p->keys;
    asm volatile("" ::: "memory");   // (LDS words one lane writes and all lanes read: see arena_alloc)
    const uint32_t cur = kchunk[0], room = kchunk[1] - cur;
    uint32_t nbase = 0;
    if (cnt > room) {   // the step's keys straddle the chunk's end: the first `room` fill it up, the rest open the next chunk
        if (lane == 0) {
            nbase = (uint32_t)atomicAdd(p->keys_top, (unsigned long long)PA_KEY_CHUNK);
            if ((uint64_t)nbase + PA_KEY_CHUNK > p->keys_cap) { atomicOr(p->status, PA_STATUS_SPILL_OVERFLOW); nbase = 0; }   // (the host sizes the stream for every read: never taken; the launch then fails instead of writing out of bounds)
        }
        nbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)nbase);
    }
    const uint32_t r = rank_in(m);
    if (key != NO_KEY) keys[r < room ? cur + r : nbase + (r - room)] = key;
    if (lane == 0) {
        if (cnt > room) { kchunk[0] = nbase + (cnt - room); kchunk[1] = nbase + PA_KEY_CHUNK; }
        else kchunk[0] = cur + cnt;
    }
    asm volatile("" ::: "memory");
}

// A finished read whose class is not known to be an index class — a window result that is a strict subset of every class seen
// (3 % of the config-3 reads), or a list-mode intersection that dropped ids — is NOT resolved here. The wave appends a 32-byte
// entry to a stream of its own (chunks of PA_DEFER_CHUNK entries) and frees the slot at once; pa_resolve_kernel (resolve.hip)
// looks the id set up by content afterwards, at full width, and writes the record / count key / colour. Round 2 resolved such
// reads in a state of the pool (ST_F_NOVEL): steps of ~11 lanes and two dependent round trips each, 9 % of the wave time.
//   window entry  {rid, coverage, mismatches, DEFER_WINDOW | count} {base1, mask1, base2, mask2}
//   list entry    {rid, coverage, mismatches, DEFER_LIST | count}   {arena offset, 0, 0, 0}     (record and ids already written)
__device__ __forceinline__ void append_deferred(bool has, u32x4 e0, u32x4 e1, uint32_t lane, karg_ptr p, lds_u32 dchunk) {
    const uint64_t m = __ballot(has);
    if (m == 0) return;
    const uint32_t cnt = (uint32_t)__popcll(m);
    const glb_v4w out = (glb_v4w)p->defer;
    asm volatile("" ::: "memory");
    const uint32_t cur = dchunk[0], room = dchunk[1] - cur;
    uint32_t nbase = 0;
    if (cnt > room) {   // (as append_keys: fill the chunk up, go on in the next one)
        if (lane == 0) {
            nbase = (uint32_t)atomicAdd(p->defer_top, (unsigned long long)PA_DEFER_CHUNK);
            if ((uint64_t)nbase + PA_DEFER_CHUNK > p->defer_cap) { atomicOr(p->status, PA_STATUS_SPILL_OVERFLOW); nbase = 0; }   // (as append_keys)
        }
        nbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)nbase);
    }
    if (has) {
        const uint32_t r = rank_in(m);
        const uint64_t at = 2ull * (r < room ? cur + r : nbase + (r - room));
        out[at] = e0;
        out[at + 1] = e1;
    }
    if (lane == 0) {
        if (cnt > room) { dchunk[0] = nbase + (cnt - room); dchunk[1] = nbase + PA_DEFER_CHUNK; }
        else dchunk[0] = cur + cnt;
    }
    asm volatile("" ::: "memory");
}

// which of the eight base ids b[] occur in the list of record `xref` (nch 16-byte chunks; the loop runs to the wave-uniform
// maxch so that every lane keeps four loads in flight per round trip)
__device__ __forceinline__ uint32_t list_hits(glb_u32 ec, uint32_t xref, uint32_t nch, uint32_t maxch, const uint32_t (&b)[8]) {
    const glb_v4 xrec = (glb_v4)(ec + 4ull * xref);
    uint32_t acc = 0;
    for (uint32_t c0 = 0; c0 < maxch; c0 += 4) {
        u32x4 w[4];
#pragma unroll
        for (uint32_t t = 0; t < 4; ++t) w[t] = xrec[c0 + t < nch ? c0 + t : 0];
#pragma unroll
        for (uint32_t t = 0; t < 4; ++t)
            if (c0 + t < nch) {
                const U4 ww{w[t].x, w[t].y, w[t].z, w[t].w};
                acc |= scan_words(ww, c0 + t == 0, b);
            }
    }
    return acc;
}

__device__ __forceinline__ uint32_t wave_max(uint32_t v) {
    for (uint32_t o = 32; o; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, (int)o, 64));
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

// ids of the list of record `xref` (`len` ids) inside the windows [b1, b1 + 32) and [b2, b2 + 32), as masks (list_window_mask of
// lane_steps.hpp, one list per lane). Lists of up to 64 ids are scanned whole, 16-byte chunks, four in flight; in a longer one the
// first id >= b1 (>= b2) is found by binary search and the nine chunks from there are scanned (32 ids and the chunk they start
// in). Every loop runs to the wave's longest range; lanes with act == false do nothing.
__device__ __forceinline__ void window_hits(glb_u32 ec, uint32_t xref, uint32_t len, bool act, uint32_t b1, uint32_t b2, uint32_t bitmap_min, uint32_t& m1, uint32_t& m2) {
    // a class with a membership bitmap (every window-less class of bitmap_min ids and more: device_layout.hpp): the 32 bits from b1 and from
    // b2 on — four loads, all in flight at once, whatever the length of the list
    const bool bm = act && bitmap_min != 0 && len >= bitmap_min;
    uint32_t bw[4] = {0u, 0u, 0u, 0u};
    if (__ballot(bm)) {
        if (bm) {
            const glb_u32 bits = ec + class_bitmap(xref, len);
            bw[0] = bits[b1 >> 5]; bw[1] = bits[(b1 >> 5) + 1]; bw[2] = bits[b2 >> 5]; bw[3] = bits[(b2 >> 5) + 1];
        }
    }
    act = act && !bm;
    const glb_v4 rec = (glb_v4)(ec + 4ull * xref);
    const uint32_t nch = act ? (len + 4) >> 2 : 0u;
    uint32_t c[2] = {0u, 0u}, e[2] = {nch, 0u};
    const bool lng = act && len > 64;
    if (__ballot(lng)) {
        if (lng) {
            const glb_u32 ids = ec + 4ull * xref + 1;
#pragma unroll
            for (uint32_t r = 0; r < 2; ++r) {
                const uint32_t b = r ? b2 : b1;
                uint32_t lo = 0, hi = len;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (ids[mid] < b) lo = mid + 1; else hi = mid;
                }
                c[r] = (1 + lo) >> 2;                                      // id j is word 1 + j of the record
                e[r] = min(nch, ((1 + lo + CLASS_WINDOW - 1) >> 2) + 1);   // ids lo .. lo + 31 (sorted: no id of the window lies beyond)
            }
        }
    }
    m1 = m2 = 0;
#pragma unroll
    for (uint32_t r = 0; r < 2; ++r) {
        const uint32_t mx = wave_max(e[r] > c[r] ? e[r] - c[r] : 0u);
        for (uint32_t c0 = 0; c0 < mx; c0 += 4) {
            u32x4 w[4];
#pragma unroll
            for (uint32_t t = 0; t < 4; ++t) w[t] = rec[c[r] + c0 + t < e[r] ? c[r] + c0 + t : 0u];
#pragma unroll
            for (uint32_t t = 0; t < 4; ++t)
                if (c[r] + c0 + t < e[r]) {
                    const uint32_t v[4] = {w[t].x, w[t].y, w[t].z, w[t].w};
#pragma unroll
                    for (uint32_t i = 0; i < 4; ++i) {
                        if (i == 0 && c[r] + c0 + t == 0) continue;   // word 0 of a record is its class id (0xFFFFFFFF padding never falls into a window)
                        const uint32_t d1 = v[i] - b1, d2 = v[i] - b2;
                        if (d1 < CLASS_WINDOW) m1 |= 1u << d1;
                        if (d2 < CLASS_WINDOW) m2 |= 1u << d2;
                    }
                }
        }
    }
    if (bm) {
        m1 = (uint32_t)((((uint64_t)bw[1] << 32) | bw[0]) >> (b1 & 31u));
        m2 = (uint32_t)((((uint64_t)bw[3] << 32) | bw[2]) >> (b2 & 31u));
    }
}

}  // namespace

namespace narrow {
#define PA_K_WIDE 0
#include "map_pool_kernel.inc"
#undef PA_K_WIDE
}  // namespace narrow

namespace wide {
#define PA_K_WIDE 1
#include "map_pool_kernel.inc"
#undef PA_K_WIDE
}  // namespace wide

size_t pool_slot_bytes(uint32_t wpr) {   // LDS bytes per read slot: reads of more than PA_LDS_READ_WORDS words stay in HBM and take the wide lane state
    return wpr > PA_LDS_READ_WORDS ? (size_t)wide::SLOT_FIXED_BYTES : 8 * (size_t)wpr + narrow::SLOT_FIXED_BYTES;
}
size_t pool_fixed_bytes() { return POOL_FIXED; }
uint32_t pool_max_slots() { return POOL_MAX_SLOTS; }

size_t pool_lds_bytes(uint32_t wpr, uint32_t slots) {
    const size_t wave_bytes = (POOL_FIXED + (size_t)slots * pool_slot_bytes(wpr) + 15) & ~(size_t)15;
    return wave_bytes * (PA_MAP_BLOCK / 64);
}

template <class K>
static int launch_kernel(K kernel, const MapParams& p, uint32_t grid, size_t lds_bytes, hipStream_t stream) {
    if (lds_bytes > 48 * 1024) {   // opt in to more than the default dynamic LDS limit
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(PA_MAP_BLOCK), lds_bytes, stream, p);
    return (int)hipGetLastError();
}

int launch_map_pool(const MapParams& p, uint32_t grid, size_t lds_bytes, hipStream_t stream) {
    const bool gread = p.wpr > PA_LDS_READ_WORDS;   // the read stays in its HBM tile: the wide lane state (lane_steps.hpp), reads of up to PA_MAX_READ_LEN bases
    if (gread) return p.trace ? launch_kernel(&wide::pa_map_pool_kernel<true, true, false, false>, p, grid, lds_bytes, stream)
                              : launch_kernel(&wide::pa_map_pool_kernel<false, true, false, false>, p, grid, lds_bytes, stream);
    if (p.trace) return launch_kernel(&narrow::pa_map_pool_kernel<true, false, false, false>, p, grid, lds_bytes, stream);
#ifdef PA_DEBUG_KNOBS   // the statistics / ablation instantiation exists in A/B builds only
    if (p.dbg || p.ablate) return launch_kernel(&narrow::pa_map_pool_kernel<false, false, true, false>, p, grid, lds_bytes, stream);
#endif
    if (p.ix.k > 32)   // two-word k-mers: their probes ride in forward iterations through the two-word dictionary's own issue / complete pair
        return p.pool_slots == 128 ? launch_kernel(&narrow::pa_map_pool_kernel<false, false, false, true, true>, p, grid, lds_bytes, stream)
                                   : launch_kernel(&narrow::pa_map_pool_kernel<false, false, false, false, true>, p, grid, lds_bytes, stream);
    if (p.pool_slots == 128) return launch_kernel(&narrow::pa_map_pool_kernel<false, false, false, true>, p, grid, lds_bytes, stream);
    return launch_kernel(&narrow::pa_map_pool_kernel<false, false, false, false>, p, grid, lds_bytes, stream);
}

int pool_kernel_occupancy(size_t lds_bytes, int* blocks_per_cu) {
    return (int)hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, reinterpret_cast<const void*>(&narrow::pa_map_pool_kernel<false, false, false, false>),
                                                             PA_MAP_BLOCK, lds_bytes);
}

}  // namespace pa
