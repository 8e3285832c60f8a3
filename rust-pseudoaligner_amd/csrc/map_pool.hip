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

// node list of a finished read (TRACE builds: the map_read_to_nodes test surface), read-major, stride spill_cap
template <bool TRACE>
__device__ __forceinline__ void trace_out(const Lane& s, bool mapped, uint32_t gslot, karg_ptr p) {
    if (!TRACE) return;
    const uint32_t spill_cap = p->spill_cap;
    const uint32_t nt = l_ntrace(s);
    const uint32_t nn = mapped ? (nt < spill_cap ? nt : spill_cap) : 0;
    ((glb_u32w)p->nodes_len)[s.rid] = mapped ? nt : 0;
    const glb_u32w tr = (glb_u32w)p->trace + (uint64_t)gslot * spill_cap;
    const glb_u32w out = (glb_u32w)p->nodes_out + (uint64_t)s.rid * spill_cap;
    for (uint32_t j = 0; j < nn; ++j) out[j] = tr[j];
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

// list mode: record of one finished read; returns its count key and leaves the lane in ST_EMPTY. `defer` is set (and NO_KEY
// returned) when the class of a strict-subset result still has to be looked up by content before it can be counted
template <bool TRACE>
__device__ __forceinline__ uint32_t emit_record(Lane& s, uint32_t cnt, uint32_t cnt_alloc, uint64_t my_off, uint32_t base_len,
                                                uint32_t base_colour, uint32_t gslot, karg_ptr p, bool& defer) {
    uint32_t colour = NO_CLASS, class_off = (uint32_t)my_off;
    bool novel = false;
    if (my_off + cnt_alloc > p->arena_cap) atomicOr(p->status, PA_STATUS_ARENA_FULL);
    if (cnt == base_len) {   // the class IS index class base_colour: returned by reference, nothing was written to the arena
        colour = base_colour;
        class_off = PA_CLASS_REF | base_colour;
    } else novel = cnt != 0;
    if (l_flags(s) & F_SPILL_OVERFLOW) atomicOr(p->status, PA_STATUS_SPILL_OVERFLOW);
    ((glb_v4w)p->results)[s.rid] = u32x4{l_cov(s), l_mism(s) | PA_MAPPED_BIT, class_off, cnt};
    trace_out<TRACE>(s, true, gslot, p);
    const glb_u32w colour_out = (glb_u32w)p->colour_out;
    if (novel && (p->keys != nullptr || colour_out != nullptr) && my_off + cnt_alloc <= p->arena_cap) {   // content lookup: deferred (resolve.hip)
        defer = true;
        s.lk = 0;   // ST_EMPTY
        return NO_KEY;
    }
    if (colour_out) colour_out[s.rid] = colour;
    s.lk = 0;   // ST_EMPTY
    const uint32_t num_classes = p->ix.num_classes;
    return cnt == 0 ? num_classes + 1 : colour == NO_CLASS ? num_classes : colour;   // the read's count key
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

// the queue a slot goes to after a step
__device__ __forceinline__ uint32_t queue_of(Lane& s, uint32_t K) {
    uint32_t st = l_st(s);
    if (st == ST_ISECT) {   // the walk just ended: window mode has nothing left to intersect (but for pending classes); list mode goes to its tier
        const uint32_t fl = l_flags(s);
        if (!(fl & F_LISTS)) {
            const uint32_t todo = window_todo(s);
            if (todo == 2) { restart_lists(s, K); return ST_SEEK; }   // nothing but classes without windows: once more, collecting lists
            st = todo ? ST_F_MASK : ST_F_BITS;
        } else st = l_ncol(s) <= 3 ? ST_F_LIGHT : (fl & F_SMALL_BASE) ? ST_F_SCAN : ST_F_COOP;
        l_set_st(s, st);
    }
    return st == ST_NONE ? (uint32_t)ST_F_BITS : st;   // unmapped reads share the output queue
}

// a free slot takes read `rid`: packed words from the tile into LDS (lanes of consecutive reads: coalesced), fresh lane state
template <bool GREAD>
__device__ __forceinline__ void refill_slot(Lane& s, uint64_t rid, uint32_t slot, karg_ptr p, lds_u64 rd, lds_u32 wc, uint32_t S, uint32_t wpr,
                                            uint32_t K) {
    const uint32_t* lens = p->lens;
    uint32_t L = lens ? lens[rid] : p->uniform_len;   // (a wave-uniform choice: a uniform batch does not fetch lengths at all)
    if (L > wpr * 32) L = wpr * 32;
    if (!GREAD) {
        const uint64_t* src = p->tiles + ((rid >> 6) * wpr) * 64 + (rid & 63);
        // the common read lengths in straight-line code (all loads in flight, then the LDS stores; a loop over a run-time word count
        // costs a scalar branch around every load and every store): 100 bp = 4 words, 150 bp = 5
        if (wpr == 5) {
            const bool nt = (PA_NT & 1) || p->ix.stream_nt;
            const uint64_t v0 = ld_stream(src, nt), v1 = ld_stream(src + 64, nt), v2 = ld_stream(src + 128, nt), v3 = ld_stream(src + 192, nt), v4 = ld_stream(src + 256, nt);
            rd[slot] = v0; rd[S + slot] = v1; rd[2 * S + slot] = v2; rd[3 * S + slot] = v3; rd[4 * S + slot] = v4;
        } else if (wpr == 4) {
            const bool nt = (PA_NT & 1) || p->ix.stream_nt;
            const uint64_t v0 = ld_stream(src, nt), v1 = ld_stream(src + 64, nt), v2 = ld_stream(src + 128, nt), v3 = ld_stream(src + 192, nt);
            rd[slot] = v0; rd[S + slot] = v1; rd[2 * S + slot] = v2; rd[3 * S + slot] = v3;
        } else
        for (uint32_t w0 = 0; w0 < wpr; w0 += 8) {   // eight words in flight per round trip
            uint64_t v[8];
#pragma unroll
            for (uint32_t i = 0; i < 8; ++i) v[i] = w0 + i < wpr ? PA_LD(1, src + (uint64_t)(w0 + i) * 64) : 0ull;
#pragma unroll
            for (uint32_t i = 0; i < 8; ++i)
                if (w0 + i < wpr) rd[(w0 + i) * S + slot] = v[i];
        }
    }
    lane_start(s, (uint32_t)rid, L, K);
    wc[2 * slot + 1] = (uint32_t)rid;
}

constexpr uint32_t SLOT_FIXED_BYTES = 32 + 16 + 8;   // lane state, class windows, {class id, read id} (the state byte lives in the fixed area)

}  // namespace

// GREAD: reads too long for the LDS (more than PA_LDS_READ_WORDS words: long transcripts mapped onto their own graph,
// src/build_index.rs:309) stay in their HBM tile and every step fetches the words it needs from there; a slot then holds
// only state, windows and ids.
// DBG: the statistics (PA_MAP_STATS) and ablation (PA_MAP_ABLATE) build of the same text; the production build has neither the
// clock reads nor the parameter loads they need.
#ifndef PA_MAP_MIN_BLOCKS
#define PA_MAP_MIN_BLOCKS 3   // workgroups per CU the register budget is sized for (A/B builds: -DPA_MAP_MIN_BLOCKS=4)
#endif
// S128: the pool has 128 slots per wave (reads of up to 5 words: every short-read batch) — the stride of the LDS rows of read words is
// then a shift instead of a multiply in every step that touches the read (and the compiler keeps fewer scalars: 16 spilled instead of 26).
template <bool TRACE, bool GREAD, bool DBG, bool S128 = false>
__global__ __launch_bounds__(PA_MAP_BLOCK, PA_MAP_MIN_BLOCKS) void pa_map_pool_kernel(const MapParams p_arg) {
    // The ~50 words of parameters are NOT kept in registers across the loop (the allocator would spill most of them to
    // VGPR lanes and pay a v_readlane + hazard nops at every use): each iteration re-reads what its step needs from the
    // kernarg segment with scalar loads (scalar cache hits). The empty asm makes the pointer opaque per iteration.
    karg_ptr kp = (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    (void)p_arg;
#define p (*kp)
#define PA_DBG (DBG && p.dbg)
#define PA_ABLATE(bit) (DBG && (p.ablate & (bit)))
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t lane = lane_id();
    const uint32_t wave_in_block = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // uniform, and the compiler should know:
                                                                                                        // next / end / gslot arithmetic then runs on the scalar unit
    const uint32_t waves_per_block = PA_MAP_BLOCK / 64;
    const uint32_t wave = blockIdx.x * waves_per_block + wave_in_block;
    const uint32_t nwaves = gridDim.x * waves_per_block;
    const uint32_t S = S128 ? 128u : p.pool_slots, wpr = p.wpr;
    const uint32_t lwpr = GREAD ? 0u : wpr;   // words of a read kept in LDS

    const uint32_t wave_bytes = (POOL_FIXED + S * (8 * lwpr + SLOT_FIXED_BYTES) + 15) & ~15u;
    uint8_t* const wbase = smem + wave_in_block * wave_bytes;
    const lds_u64w chunk = (lds_u64w)wbase;
    const lds_u32 dbg = (lds_u32)(wbase + 16);                            // [0..ST_NSTAT) iterations, [ST_NSTAT..2*ST_NSTAT) slots served; entry ST_COUNT = dual iterations
    const lds_u64w dbg_clk = (lds_u64w)(wbase + 16 + 8 * ST_NSTAT);        // wall ticks per state
    const lds_u32 kchunk = (lds_u32)(wbase + 256);   // {cur, end} of this wave's chunk of the key stream (both 0: none taken yet)
    const lds_u32 dchunk = (lds_u32)(wbase + 264);   // the same for the stream of deferred reads
    const lds_u64 rd = (lds_u64)(wbase + POOL_FIXED);
    // the lane state as TWO arrays of one 16-byte vector per slot (not one array of 32-byte records: with a 32-byte stride the
    // vectors of 64 random slots fall into 4 bank groups, with 16 bytes into 8 — half of the LDS cycles were bank conflicts)
    const lds_v4 stA = (lds_v4)(wbase + POOL_FIXED + 8 * lwpr * S);
    const lds_v4 stB = (lds_v4)(wbase + POOL_FIXED + (8 * lwpr + 16) * S);
    const lds_v4 win = (lds_v4)(wbase + POOL_FIXED + (8 * lwpr + 32) * S); // {base1, mask1, base2, mask2}
    const lds_u32 wc = (lds_u32)(wbase + POOL_FIXED + (8 * lwpr + 48) * S);   // {class id, read id} per slot
    // Scheduling state: ONE byte per slot = the state the slot waits in (0xFF: no such slot), laid out so that lane i reads the
    // bytes of slots i and i + 64 with one 16-bit load. There are no queues: every iteration the lanes look at their two
    // bytes, ballots give the population of every state, and the batch of a step is "the first 64 slots in that state"
    // (compacted through the 64-byte pop list). A slot changes state by ONE byte store.
    const lds_u8 sb = (lds_u8)(wbase + 768);
    const lds_u8 poplist = (lds_u8)(wbase + 896);
    if (lane < 64) ((lds_u32)wbase)[lane] = 0;
    if (lane < 4) kchunk[lane] = 0;   // (kchunk and dchunk)
    ((lds_u16)sb)[lane] = (uint16_t)((lane < S ? (uint32_t)ST_EMPTY : 0xFFu) | ((lane + 64 < S ? (uint32_t)ST_EMPTY : 0xFFu) << 8));

    // Work distribution: chunks of up to 16 tiles (1024 reads). Chunk w is wave w's first one; further chunks come from a
    // global counter, so that the waves finish together whatever their reads cost (a static split left the chip 9 % idle
    // at the end of a 100 M-read launch), and shrink towards the end of the launch (a wave needs ~0.4 ms for 1024 reads:
    // fixed chunks left the chip half idle for that long). One grab per ~100 iterations: far from the ~88 M ops/s of one
    // hot atomic word.
    const uint32_t ntiles = (uint32_t)((p.n_reads + 63) >> 6);
    const uint32_t chunk_tiles = ntiles / (nwaves * 4) >= 16 ? 16u : ntiles / (nwaves * 4) >= 1 ? ntiles / (nwaves * 4) : 1u;
    uint64_t next = (uint64_t)wave * chunk_tiles << 6;
    uint64_t end = (uint64_t)(wave + 1) * chunk_tiles << 6;
    if (end > p.n_reads) end = p.n_reads;
    if (next > end) next = end;
    bool more = true;   // chunks may be left
    uint32_t seen = nwaves * chunk_tiles;   // tiles known to be handed out

    const bool counting = p.keys != nullptr;   // class-count keys are wanted (pa_map_count_batch_device)

#define PA_CNT(t) ((uint32_t)__popcll(__ballot(st_lo == (t))) + (uint32_t)__popcll(__ballot(st_hi == (t))))
    // the first nn slots in state t, one per lane (lanes >= nn: slot 0)
#define PA_POP(t, nn, out)                                                                         \
    {                                                                                              \
        const uint64_t m0_ = __ballot(st_lo == (t)), m1_ = __ballot(st_hi == (t));                 \
        const uint32_t r1_ = (uint32_t)__popcll(m0_) + rank_in(m1_);                               \
        if (st_lo == (t)) poplist[rank_in(m0_)] = (uint8_t)lane;                                   \
        if (st_hi == (t) && r1_ < 64) poplist[r1_] = (uint8_t)(lane + 64);                         \
        asm volatile("" ::: "memory");                                                             \
        out = lane < (nn) ? (uint32_t)poplist[lane] : 0u;                                          \
        asm volatile("" ::: "memory");                                                             \
    }

    for (;;) {
        PA_MARK("loop_top");
        asm volatile("" : "+s"(kp) : : "memory");   // also: slots and queues in LDS change hands between lanes every iteration
        const DevIndexView ix = view_of(kp);
        const glb_u32 ec = (glb_u32)ix.ec;
        const uint32_t K = ix.k, allowed = p.allowed, spill_cap = p.spill_cap;
        if (next == end && more) {   // this wave's chunk is used up: take the next one
            // guided: half of an even share of what was left at this wave's previous grab, 2..16 tiles (the estimate is one
            // chunk old, so the sizes decay geometrically towards the end and the last chunks are ~128 reads)
            const uint32_t share = (ntiles > seen ? ntiles - seen : 0u) / (2 * nwaves);
            const uint32_t take = share >= 16 ? 16u : share >= 2 ? share : 2u;
            uint32_t t0 = 0;
            if (lane == 0) t0 = atomicAdd(p.tile_ctr, take);
            t0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)t0) + nwaves * chunk_tiles;
            seen = t0 + take;
            if (t0 >= ntiles) more = false;
            else {
                next = (uint64_t)t0 << 6;
                end = (uint64_t)(t0 + take) << 6;
                if (end > p.n_reads) end = p.n_reads;
            }
        }
        // ---- 1. pick a state: the first one (states nearest to completion first) that fills a wave, else the most populated
        const uint32_t st2 = (uint32_t)((lds_u16)sb)[lane];
        const uint32_t st_lo = st2 & 0xFFu, st_hi = st2 >> 8;
        const uint64_t left = end - next;
        const uint32_t nempty = PA_CNT(ST_EMPTY);
        const uint32_t nrefill = (uint32_t)(left < (uint64_t)nempty ? left : (uint64_t)nempty);
        uint32_t best = 0, sel = ST_EMPTY;
        // `best` is the WEIGHT of the choice (64 and more = "fills a wave": the first such state in this order wins), `bestn` the
        // slots it really holds. A rare state weighs as a full wave from PA_RARE_MIN slots on: slots parked in a rare state are
        // slots the common steps cannot use (with the plain "most populated" rule the rare states only ran once they
        // outnumbered the common ones, i.e. when a good part of the pool was parked); the best trade between a low-width rare
        // iteration and parked slots is at 12...17 slots for the rare states of configs 3 and 5
        uint32_t bestn = 0;
#define PA_CONSIDER(t, c) { const uint32_t c_ = (c); if (best < 64 && c_ > best) { best = c_; bestn = c_; sel = (t); } }
#define PA_CONSIDER_RARE_MIN(t, c, mn) { const uint32_t c_ = (c), w_ = (PA_RARE_MIN && c_ >= (mn)) ? 64u + c_ : c_; \
                                         if (best < 64 && w_ > best) { best = w_; bestn = c_; sel = (t); } }
#define PA_CONSIDER_RARE(t, c) PA_CONSIDER_RARE_MIN(t, c, PA_RARE_MIN)
        // the five rare states (left extension, the list-mode tiers, the pending classes of window mode) are only counted when some slot is in one
        // of them: one ballot instead of ten in most iterations (the order of consideration is the same either way)
        constexpr uint32_t RARE = (1u << ST_LEFT) | (1u << ST_F_LIGHT) | (1u << ST_F_SCAN) | (1u << ST_F_COOP) | (1u << ST_F_MASK);
        const bool any_rare = __ballot((((RARE >> (st_lo & 31u)) | (RARE >> (st_hi & 31u))) & 1u) != 0) != 0;   // (0xFF, no slot: bit 31, not rare)
        const uint32_t n_bits_q = PA_CNT(ST_F_BITS), n_seek_q = PA_CNT(ST_SEEK), n_fwd_q = PA_CNT(ST_FWD);
        if (any_rare) {
            PA_CONSIDER_RARE_MIN(ST_F_COOP, PA_CNT(ST_F_COOP), PA_COOP_MIN)   // (the wave takes its reads one at a time: nothing to gain from gathering them)
            PA_CONSIDER_RARE(ST_F_SCAN, PA_CNT(ST_F_SCAN))
            PA_CONSIDER_RARE(ST_F_LIGHT, PA_CNT(ST_F_LIGHT))
            PA_CONSIDER_RARE(ST_F_MASK, PA_CNT(ST_F_MASK))
            PA_CONSIDER(ST_F_BITS, n_bits_q)
            PA_CONSIDER(ST_FWD, n_fwd_q)
            PA_CONSIDER_RARE(ST_LEFT, PA_CNT(ST_LEFT))
            PA_CONSIDER(ST_SEEK, n_seek_q)
            PA_CONSIDER(ST_EMPTY, nrefill)
        } else {
            // the common iteration (no slot in a rare state): the same rule on four populations, as scalar min / max — the first of
            // output, forward, probe, refill that fills a wave, else the most populated (ties: the earlier one)
            const uint32_t cb = min(n_bits_q, 64u), cf = min(n_fwd_q, 64u), cs = min(n_seek_q, 64u), ce = min(nrefill, 64u);
            best = max(max(cb, cf), max(cs, ce));
            sel = cb == best ? (uint32_t)ST_F_BITS : cf == best ? (uint32_t)ST_FWD : cs == best ? (uint32_t)ST_SEEK : (uint32_t)ST_EMPTY;
            bestn = cb == best ? n_bits_q : cf == best ? n_fwd_q : cs == best ? n_seek_q : nrefill;
        }
#undef PA_CONSIDER
#undef PA_CONSIDER_RARE
#undef PA_CONSIDER_RARE_MIN
        if (best == 0) break;
        // DUAL iteration: a forward step and a dictionary probe are each one dependent round trip and touch different parts
        // of the memory system (node blobs: MALL / L2; dictionary: HBM). When both queues hold work the wave pops BOTH, lets
        // the probe's slot load and the node fetch go out back to back, and does the probe's arithmetic while the
        // node lines are on their way: two round trips in flight per wave instead of one (K <= 32 only: the two-word dictionary's
        // probe is a step of its own).
#ifndef PA_SEEK_MIN   // probes ride with a forward step only from this many waiting slots on (or when little forward work is left): the probe half is
#define PA_SEEK_MIN 32u   // the whole wave's instructions however few lanes it serves (same-box A/B of 1 / 32 / 48: config 3 -1.8 %, config 5 -3.8 % time at 32)
#endif
        const bool seek_ok = n_seek_q >= PA_SEEK_MIN || n_fwd_q < 24;
        const bool dual = (sel == ST_FWD || sel == ST_SEEK) && K <= 32 && n_seek_q != 0 && n_fwd_q != 0 && seek_ok && !PA_ABLATE(4u);
        if (dual) sel = ST_FWD;
        uint32_t n_own = dual ? (n_fwd_q < 64 ? n_fwd_q : 64) : bestn < 64 ? bestn : 64;
        if (PA_REFILL_ALIGN > 1 && sel == ST_EMPTY && n_own >= PA_REFILL_ALIGN && (uint64_t)n_own < left) n_own &= ~(PA_REFILL_ALIGN - 1u);   // (A/B builds: see the output step)
        // an output step that does not fill the wave takes EMPTY slots into its idle lanes: they are refilled by the same text
        // (slots freed by the rare finishing steps otherwise wait, parked, for a refill step of their own)
        const uint32_t n_fill = (PA_FILL && sel == ST_F_BITS && left != 0) ? (64 - n_own < nempty ? 64 - n_own : nempty) : 0u;
        const uint32_t n = n_own + n_fill;
        const uint32_t n2 = dual ? (n_seek_q < 64 ? n_seek_q : 64) : 0u;   // lanes of the second (SEEK) batch
        if (PA_DBG && lane == 0) {
            dbg[dual ? ST_DUAL : sel] += 1;
            dbg[ST_NSTAT + (dual ? ST_DUAL : sel)] += n + n2;
        }
        const unsigned long long t_sec = PA_DBG ? __builtin_readcyclecounter() : 0ull;

        // ---- 2. pop n slots (and n2 slots of the SEEK queue)
        PA_MARK("picked");
        const bool active = lane < n;
        uint32_t slot, slot2 = 0;
        PA_POP(sel, n_own, slot)
        if (n_fill) {   // lanes n_own .. n - 1: the first n_fill EMPTY slots
            uint32_t eslot;
            PA_POP((uint32_t)ST_EMPTY, n_fill, eslot)
            const uint32_t mine = (uint32_t)__shfl((int)eslot, (int)((lane - n_own) & 63u), 64);
            if (lane >= n_own && active) slot = mine;
        }
        const bool active2 = lane < n2;
        if (dual) PA_POP((uint32_t)ST_SEEK, n2, slot2)
        const uint32_t gslot = wave * S + slot;
        Lane s;
        {
            const u32x4 a = stA[slot], b = stB[slot];
            s.lk = a.x; s.cm = a.y; s.h = a.z; s.of = a.w; s.rr = b.x; s.rm = b.y; s.ph = b.z; s.nc = b.w;
            s.rid = wc[2 * slot + 1];
            if (n_fill && lane >= n_own) s.lk = 0;   // an EMPTY slot taken along by an output step (its stored state may be stale)
        }
        const ReadRef rr = GREAD ? ReadRef{p.tiles + ((uint64_t)(s.rid >> 6) * wpr) * 64 + (s.rid & 63), 64u, wpr} : ReadRef{(const uint64_t*)(rd + slot), S, wpr, true};
        const glb_u32w row = (glb_u32w)p.spill + (uint64_t)gslot * spill_cap;
        const ColRef cols{(uint32_t*)&win[slot], (uint32_t*)(wc + 2 * slot), (uint32_t*)row, (uint32_t*)(row + 4), (uint32_t*)(row + 8),
                          (uint32_t*)(row + LIST_ROW_HDR), spill_cap - LIST_ROW_HDR, (uint32_t*)row,
                          TRACE ? (uint32_t*)((glb_u32w)p.trace + (uint64_t)gslot * spill_cap) : nullptr};

        uint32_t nq2 = 0xFFu;   // DUAL: the queue the second batch's slot goes to
        PA_MARK("popped");
        const unsigned long long t_pop = PA_DBG ? __builtin_readcyclecounter() : 0ull;
        // ---- 3. the step
        if (sel == ST_EMPTY) {   // REFILL: free slots take the next reads of this wave's range (coalesced: lane = consecutive read)
            if (active) refill_slot<GREAD>(s, next + lane, slot, kp, rd, wc, S, wpr, K);
            next += n;
        } else if (sel == ST_SEEK) {
            if (active) seek_step(s, ix, rr);
        } else if (sel == ST_FWD) {
            // One text for the plain forward step and the DUAL iteration (n2 = 0: the probe half runs on all-zero states and is
            // thrown away). Straight-line issue: every lane executes every load (a lane without a slot carries an all-zero state:
            // blob 0, bucket of whatever slot 0 holds) and nothing branches between the loads and their first use, so that the
            // waits stay exact: first the probe's slot, then the node.
            // Nothing the wave asked memory for is to be in flight when this iteration's loads go out. Every load of an iteration is consumed
            // inside it, but the compiler cannot prove that across the loop's back edge (fwd_finish sits behind `if (active)`: for all it
            // knows a wave may skip it and carry the block's loads along), so it guarded the first write to one of their registers in the
            // NEXT iteration with a wait for everything in flight — and that write came a few instructions behind the dictionary probe's
            // load: a whole memory round trip before the block's loads even went out, two round trips in series per iteration instead of
            // the two in flight together this step is built for. An explicit wait here (free: at most the stores of an output step are
            // still on their way) tells the compiler where it stands.
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), nothing else
            Lane s2;
            {
                const u32x4 a = stA[slot2], b = stB[slot2];
                s2.lk = active2 ? a.x : 0u; s2.cm = active2 ? a.y : 0u; s2.h = active2 ? a.z : 0u; s2.of = active2 ? a.w : 0u;
                s2.rr = active2 ? b.x : 0u; s2.rm = active2 ? b.y : 0u; s2.ph = active2 ? b.z : 0u; s2.nc = active2 ? b.w : 0u;
                s2.rid = 0;   // (not used by the probe; the slot keeps its read id in `wc`)
            }
            if (!active) { s.lk = s.cm = s.h = s.of = s.rr = s.rm = s.ph = s.nc = 0; }
            const uint32_t rid2 = GREAD && active2 ? wc[2 * slot2 + 1] : 0u;   // (a lane without a slot probes with read 0)
            const ReadRef rr2 = GREAD ? ReadRef{p.tiles + ((uint64_t)(rid2 >> 6) * wpr) * 64 + (rid2 & 63), 64u, wpr} : ReadRef{(const uint64_t*)(rd + slot2), S, wpr};
            SeekProbe pq;
            FwdLoad fl;
            PA_MARK("dual_state2");
            pq.klo = pq.khi = pq.pending = 0; pq.v = U4{0u, 0u, NO_HANDLE, 0u};
            if (n2) seek_issue(s2, ix, rr2, pq);                       // one slot of the k-mer's bucket (HBM); (n2: wave-uniform — a plain forward step skips the probe half)
            // a lane whose scan is past a miss probes the next position of the scan as well (another line, in flight together)
            // (only in steps where at least eight lanes do: the second k-mer and hash are the whole wave's instructions)
            const bool two_l = active2 && seek_two(s2, K);
            const bool pairs = __popcll(__ballot(two_l)) >= 8;
            const bool two = two_l && pairs;
            SeekProbe pq1;
            pq1.klo = pq1.khi = pq1.pending = 0; pq1.v = U4{0u, 0u, NO_HANDLE, 0u};
            if (pairs) seek_issue(s2, ix, rr2, pq1, PA_SEEK_STRIDE, two);   // (a wave-uniform branch: the waits behind it stay exact)
            fwd_issue(s, ix, fl);                                      // node header + sequence words (MALL / L2)
            __builtin_amdgcn_sched_barrier(0);                         // (left alone the scheduler finishes the probe first and only then issues the node loads)
            PA_MARK("dual_issued");
            // the probe half completes first: its loads are the oldest of the iteration, a probe never needs a second one (a key that is
            // not in the slot looked at costs the lane another step: lane_steps.hpp), and its registers are free before the forward half computes
            if (active2) {
                seek_complete2(s2, K, pq, two, pq1);
                nq2 = queue_of(s2, K);   // (may rewrite the state: before the store)
                stA[slot2] = u32x4{s2.lk, s2.cm, s2.h, s2.of};
                stB[slot2] = u32x4{s2.rr, s2.rm, s2.ph, s2.nc};
            }
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long t1 = PA_DBG ? __builtin_readcyclecounter() : 0ull;
            PA_MARK("dual_second");
            if (active) fwd_finish<TRACE>(s, ix, rr, cols, allowed, fl);
            PA_MARK("dual_fwd_done");
            if (PA_DBG && lane == 0) {   // statistics only: issue | wait + compute of the forward half
                const unsigned long long t3 = __builtin_readcyclecounter();
                dbg[ST_COUNT + 1] += 1; dbg[ST_COUNT + 3] += 1;
                dbg_clk[ST_COUNT + 1] += t1 - t_pop; dbg_clk[ST_COUNT + 3] += t3 - t1;
            }
            PA_MARK("dual_seek_done");
        } else if (sel == ST_LEFT) {
            if (active) left_step<TRACE>(s, ix, rr, cols, allowed);
        } else if (sel == ST_F_BITS) {
            PA_MARK("bits_begin");
            // output of window-mode reads and of unmapped reads: no loads. A non-empty window that is a strict subset of
            // every class seen goes on to NOVEL (is it an index class all the same?) and is written there.
            uint32_t ckey = NO_KEY;
            bool dfr = false;
            u32x4 d0{0u, 0u, 0u, 0u}, d1{0u, 0u, 0u, 0u};
            if (active && lane < n_own) {
                const bool mapped = l_st(s) != ST_NONE;
                const u32x4 w = win[slot];
                const uint32_t cand = wc[2 * slot];
                const uint32_t count = mapped ? (uint32_t)(__popc(w.y) + __popc(w.w)) : 0u;
                if (mapped && count != 0 && cand == NO_CLASS) {   // a strict subset of every class seen: is it an index class all the same? resolve.hip finds out
                    dfr = true;
                    d0 = u32x4{s.rid, l_cov(s), l_mism(s), PA_DEFER_WINDOW | count};
                    d1 = w;
                    trace_out<TRACE>(s, true, gslot, kp);
                    s.lk = 0;   // ST_EMPTY: the slot is free at once
                } else {
                    const bool is_ref = mapped && count != 0;
                    if (!PA_ABLATE(1u))
                        store_result((glb_v4w)p.results + s.rid, mapped ? u32x4{l_cov(s), l_mism(s) | PA_MAPPED_BIT, is_ref ? PA_CLASS_REF | cand : 0u, count}
                                                                        : u32x4{0u, 0u, 0u, 0u});
                    trace_out<TRACE>(s, mapped, gslot, kp);
                    const glb_u32w colour_out = (glb_u32w)p.colour_out;
                    if (colour_out) colour_out[s.rid] = is_ref ? cand : NO_CLASS;
                    ckey = !mapped ? ix.num_classes + 2 : count == 0 ? ix.num_classes + 1 : cand;
                    s.lk = 0;   // ST_EMPTY
                }
            }
            // ... and the slots that just became free take the wave's next reads in the same step (the tile loads are in
            // flight together with the result stores: one wait instead of an output step and a refill step)
            {
                const uint64_t freed = __ballot(active && s.lk == 0);
                const uint32_t nfree = (uint32_t)__popcll(freed);
                uint32_t take = (uint32_t)(left < (uint64_t)nfree ? left : (uint64_t)nfree);
                // (measured, round 5: whole lines save 0.06 read requests per read — a refill that ends inside a line leaves the rest to the
                // next one, by which time the line has left the L2 — and cost more in slots left EMPTY for an iteration: config 3 +0.5 %,
                // config 5 +2 % time. Not the default.)
                if (PA_REFILL_ALIGN > 1 && take >= PA_REFILL_ALIGN && (uint64_t)take < left) take &= ~(PA_REFILL_ALIGN - 1u);
                if (take && !PA_ABLATE(8u)) {
                    if (active && s.lk == 0 && rank_in(freed) < take) refill_slot<GREAD>(s, next + rank_in(freed), slot, kp, rd, wc, S, wpr, K);
                    next += take;
                }
            }
            if (counting && !PA_ABLATE(2u)) append_keys(ckey, lane, kp, kchunk);
            append_deferred(dfr, d0, d1, lane, kp, dchunk);
            PA_MARK("bits_end");
        } else if (sel == ST_F_MASK) {
            // Window mode, classes without windows pending (lane_steps.hpp, mask_pending): ONE PENDING CLASS PER LANE. The
            // waiting reads are packed into the wave, one lane per pending class; the lane streams that class's ids and
            // notes which fall into the read's two windows; the masks of a read's lanes are ANDed (segmented reduction)
            // and applied to its window. The read then is a plain window-mode result (ST_F_BITS). Three round trips per
            // pass: the (ref, len) pair, the ids, and nothing else.
            const u32x4 w = active ? win[slot] : u32x4{0u, 0u, 0u, 0u};
            const uint32_t np_mine = active ? l_npend(s) : 0u;
            uint32_t a1 = w.y, a2 = w.w;
            for (uint64_t todo = __ballot(active && np_mine <= 64); todo;) {   // (as ST_F_SCAN: the longest prefix of the waiting reads that fits in 64 lanes)
                const uint32_t want = ((todo >> lane) & 1ull) ? np_mine : 0u;
                const uint32_t incl = wave_incl_scan(want);
                const uint32_t start = incl - want;
                const bool inpass = want != 0 && incl <= 64;   // the first waiting read always is
                const uint64_t pass = __ballot(inpass);
                todo &= ~pass;
                uint32_t Lr = 64, jg = 0, seg_end = 0;   // the read this lane works for, which of its pending classes, where its lanes end
                for (uint64_t m = pass; m; m &= m - 1) {   // uniform: hand the lanes out
                    const uint32_t o = (uint32_t)(__ffsll((unsigned long long)m) - 1);
                    const uint32_t st = (uint32_t)__builtin_amdgcn_readlane((int)start, (int)o);
                    const uint32_t nc = (uint32_t)__builtin_amdgcn_readlane((int)np_mine, (int)o);
                    if (lane - st < nc) { Lr = o; jg = lane - st; seg_end = st + nc; }
                }
                const bool gact = Lr < 64;
                const int src = (int)(Lr & 63u);
                const uint32_t b1 = (uint32_t)__shfl((int)w.x, src, 64), b2 = (uint32_t)__shfl((int)w.z, src, 64), slotL = (uint32_t)__shfl((int)slot, src, 64);
                const glb_u32w rowL = (glb_u32w)p.spill + (uint64_t)(wave * S + slotL) * spill_cap;
                uint32_t xref = 0, xlen = 0;
                if (gact) { xref = rowL[2 * jg]; xlen = rowL[2 * jg + 1]; }
                uint32_t m1, m2;
                window_hits(ec, xref, xlen, gact, b1, b2, ix.bitmap_min, m1, m2);
                for (uint32_t o = 1; o < 64; o <<= 1) {   // AND over the lanes of a read: lane `start` ends up with all of them
                    const uint32_t t1 = (uint32_t)__shfl_down((int)m1, o, 64), t2 = (uint32_t)__shfl_down((int)m2, o, 64);
                    if (gact && lane + o < seg_end) { m1 &= t1; m2 &= t2; }
                }
                const uint32_t r1 = (uint32_t)__shfl((int)m1, (int)(start & 63u), 64), r2 = (uint32_t)__shfl((int)m2, (int)(start & 63u), 64);
                if (inpass) { a1 &= r1; a2 &= r2; }
            }
            for (uint64_t todo = __ballot(active && np_mine > 64); todo; todo &= todo - 1) {   // more pending classes: the whole wave per read
                const uint32_t Lr = (uint32_t)(__ffsll((unsigned long long)todo) - 1);
                const uint32_t b1 = (uint32_t)__shfl((int)w.x, (int)Lr, 64), b2 = (uint32_t)__shfl((int)w.z, (int)Lr, 64);
                const uint32_t npL = (uint32_t)__shfl((int)np_mine, (int)Lr, 64), slotL = (uint32_t)__shfl((int)slot, (int)Lr, 64);
                const glb_u32w rowL = (glb_u32w)p.spill + (uint64_t)(wave * S + slotL) * spill_cap;
                uint32_t r1 = 0xFFFFFFFFu, r2 = 0xFFFFFFFFu;
                for (uint32_t g = 0; g < npL; g += 64) {
                    const uint32_t ci = g + lane;
                    const bool gact = ci < npL;
                    uint32_t xref = 0, xlen = 0;
                    if (gact) { xref = rowL[2 * ci]; xlen = rowL[2 * ci + 1]; }
                    uint32_t m1, m2;
                    window_hits(ec, xref, xlen, gact, b1, b2, ix.bitmap_min, m1, m2);
                    if (gact) { r1 &= m1; r2 &= m2; }
                }
                for (uint32_t o = 32; o; o >>= 1) {
                    r1 &= (uint32_t)__shfl_xor((int)r1, (int)o, 64);
                    r2 &= (uint32_t)__shfl_xor((int)r2, (int)o, 64);
                }
                if (lane == Lr) { a1 &= r1; a2 &= r2; }
            }
            if (active) {
                if (a1 != w.y || a2 != w.w) wc[2 * slot] = NO_CLASS;   // a strict subset of the window classes seen (and no class without windows fits a window)
                win[slot] = u32x4{w.x, a1, w.z, a2};
                s.nc = (s.nc & ~NC_COL_MASK) | 1u;
                l_set_st(s, ST_F_BITS);
            }
        } else if (sel == ST_F_SCAN) {
            // List mode, base list of <= 8 ids, other lists of any number and length: ONE LIST PER LANE. The waiting reads
            // are packed into the wave, ncol lanes each; lane j of a read's segment loads the read's base ids, streams the
            // chunks of class j's list (four 16-byte loads in flight) and notes which base ids it has seen; a base id
            // survives when no lane of the segment misses it (one ballot per base id, masked by the segment). Reads of more
            // than 64 classes take the whole wave, 64 lists at a time. Three or four round trips per pass however many
            // classes the reads met — the per-lane scan these reads used to take (isect_scan) cost ~45 round trips for a
            // 40-class read and stalled its whole step.
            Isect is;
            is.base_len = is.base_ref = is.base_colour = 0;
            if (active) isect_pick(s, cols, is);
            const uint32_t ncol_mine = l_ncol(s);
            if (active && ncol_mine > 3) is.base_colour = ec[4ull * is.base_ref];   // only needed at the end: not waited for here
            uint32_t my_alive = 0;
            // reads of at most 64 classes are packed into the wave: read r takes ncol_r consecutive lanes (lane start_r + j =
            // class j of read r); a pass takes the longest prefix of the waiting reads that fits in 64 lanes
            for (uint64_t todo = __ballot(active && ncol_mine > 0 && ncol_mine <= 64); todo;) {
                const uint32_t want = ((todo >> lane) & 1ull) ? ncol_mine : 0u;
                const uint32_t incl = wave_incl_scan(want);
                const uint32_t start = incl - want;
                const bool inpass = want != 0 && incl <= 64;   // the first waiting read always is
                const uint64_t pass = __ballot(inpass);
                todo &= ~pass;
                uint32_t Lr = 64, jg = 0;   // the read this lane works for, and which of its classes
                for (uint64_t m = pass; m; m &= m - 1) {   // uniform: hand the lanes out
                    const uint32_t o = (uint32_t)(__ffsll((unsigned long long)m) - 1);
                    const uint32_t st = (uint32_t)__builtin_amdgcn_readlane((int)start, (int)o);
                    const uint32_t nc = (uint32_t)__builtin_amdgcn_readlane((int)ncol_mine, (int)o);
                    if (lane - st < nc) { Lr = o; jg = lane - st; }
                }
                const bool gact = Lr < 64;
                const int src = (int)(Lr & 63u);
                // (every shuffle with all lanes active: a source lane that is masked off returns garbage)
                const uint32_t bref = (uint32_t)__shfl((int)is.base_ref, src, 64), blen = (uint32_t)__shfl((int)is.base_len, src, 64),
                               slotL = (uint32_t)__shfl((int)slot, src, 64);
                const glb_u32w rowL = (glb_u32w)p.spill + (uint64_t)(wave * S + slotL) * spill_cap;
                const glb_v4 brec = (glb_v4)(ec + 4ull * bref);
                const u32x4 q0 = brec[0], q1 = brec[1], q2 = brec[2];
                const uint32_t b[8] = {q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, blen > 7 ? q2.x : 0xFFFFFFFFu};
                uint32_t xref = bref, xlen = 0;
                if (gact) {
                    if (jg < LDS_CLASSES) { xref = rowL[jg]; xlen = rowL[4 + jg]; }
                    else { const u32x4 qd = *(glb_v4)(rowL + LIST_ROW_HDR + 4 * (jg - LDS_CLASSES)); xref = qd.x; xlen = qd.y; }
                }
                const bool mine = xref != bref;   // a list other than the base
                const uint32_t nch = mine ? (xlen + 4) >> 2 : 0u;
                uint32_t maxch = nch;
                for (uint32_t o = 32; o; o >>= 1) maxch = max(maxch, (uint32_t)__shfl_xor((int)maxch, (int)o, 64));
                maxch = (uint32_t)__builtin_amdgcn_readfirstlane((int)maxch);
                const uint32_t acc = list_hits(ec, xref, nch, maxch, b);
                const uint32_t miss = mine ? ~acc & 0xFFu : 0u;   // base ids this lane's list lacks
                const uint64_t seg = inpass ? (want == 64 ? ~0ull : ((1ull << want) - 1) << start) : 0ull;   // the lanes of this read
                uint32_t alive = (1u << is.base_len) - 1;
#pragma unroll
                for (uint32_t kk = 0; kk < 8; ++kk)
                    if (__ballot((miss >> kk) & 1u) & seg) alive &= ~(1u << kk);
                if (inpass) my_alive = alive;
            }
            for (uint64_t todo = __ballot(active && ncol_mine > 64); todo; todo &= todo - 1) {   // more classes: the whole wave per read
                const uint32_t Lr = (uint32_t)(__ffsll((unsigned long long)todo) - 1);
                const uint32_t bref = (uint32_t)__shfl((int)is.base_ref, (int)Lr, 64);
                const uint32_t blen = (uint32_t)__shfl((int)is.base_len, (int)Lr, 64);
                const uint32_t ncolL = (uint32_t)__shfl((int)ncol_mine, (int)Lr, 64);
                const uint32_t slotL = (uint32_t)__shfl((int)slot, (int)Lr, 64);
                const glb_u32w rowL = (glb_u32w)p.spill + (uint64_t)(wave * S + slotL) * spill_cap;
                const glb_v4 brec = (glb_v4)(ec + 4ull * bref);
                const u32x4 q0 = brec[0], q1 = brec[1], q2 = brec[2];
                const uint32_t b[8] = {q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, blen > 7 ? q2.x : 0xFFFFFFFFu};
                uint32_t alive = (1u << blen) - 1;
                for (uint32_t g = 0; g < ncolL && alive; g += 64) {
                    const uint32_t ci = g + lane;
                    uint32_t xref = bref, xlen = 0;
                    if (ci < ncolL) {
                        if (ci < LDS_CLASSES) { xref = rowL[ci]; xlen = rowL[4 + ci]; }
                        else { const u32x4 qd = *(glb_v4)(rowL + LIST_ROW_HDR + 4 * (ci - LDS_CLASSES)); xref = qd.x; xlen = qd.y; }
                    }
                    const bool mine = xref != bref;                     // a list other than the base
                    const uint32_t nch = mine ? (xlen + 4) >> 2 : 0u;
                    uint32_t maxch = nch;
                    for (uint32_t o = 32; o; o >>= 1) maxch = max(maxch, (uint32_t)__shfl_xor((int)maxch, (int)o, 64));
                    maxch = (uint32_t)__builtin_amdgcn_readfirstlane((int)maxch);
                    const uint32_t acc = list_hits(ec, xref, nch, maxch, b);
#pragma unroll
                    for (uint32_t kk = 0; kk < 8; ++kk)
                        if (__ballot(mine && !((acc >> kk) & 1u))) alive &= ~(1u << kk);
                }
                if (lane == Lr) my_alive = alive;
            }
            const uint32_t cnt = active ? (uint32_t)__popc(my_alive) : 0u;
            const uint32_t cnt_alloc = active && cnt != is.base_len ? cnt : 0u;   // a result that is an index class is returned by reference
            const uint64_t my_off = arena_alloc(cnt_alloc, lane, kp, chunk);
            uint32_t ckey = NO_KEY;
            bool dfr = false;
            u32x4 d0{0u, 0u, 0u, 0u}, d1{0u, 0u, 0u, 0u};
            if (active) {
                if (cnt_alloc && my_off + cnt_alloc <= p.arena_cap) {
                    const glb_u32w dst = (glb_u32w)p.arena + my_off;
                    const glb_u32 bids = ec + 4ull * is.base_ref + 1;
                    uint32_t k = 0;
                    for (uint32_t t = my_alive; t; t &= t - 1) dst[k++] = bids[__ffs((int)t) - 1];
                }
                const uint32_t cov_ = l_cov(s), mm_ = l_mism(s);
                ckey = emit_record<TRACE>(s, cnt, cnt_alloc, my_off, is.base_len, is.base_colour, gslot, kp, dfr);
                d0 = u32x4{s.rid, cov_, mm_, PA_DEFER_LIST | cnt};
                d1 = u32x4{(uint32_t)my_off, 0u, 0u, 0u};
            }
            if (counting) append_keys(ckey, lane, kp, kchunk);
            append_deferred(dfr, d0, d1, lane, kp, dchunk);
        } else if (sel == ST_F_COOP) {
            // the whole wave works on one read at a time (list mode, base list of more than 8 ids). Lane e owns base ids
            // e, e+64, ...; membership in every other list is a scan of 16-byte loads (short lists) or a binary search;
            // survivors are compacted with a wave ballot straight into the read's arena slice.
            Isect is;
            is.count = 0;
            is.base_len = is.base_ref = is.base_colour = 0;
            if (active) isect_pick(s, cols, is);
            if (active && l_ncol(s) > 3) is.base_colour = ec[4ull * is.base_ref];   // only needed at the end: not waited for here
            const uint32_t cnt_alloc = active ? is.base_len : 0u;   // upper bound: the survivors are a subset of the base list
            const uint64_t my_off = arena_alloc(cnt_alloc, lane, kp, chunk);
            const uint64_t arena_cap = p.arena_cap;
            const glb_u32w arena_g = (glb_u32w)p.arena;
            const uint32_t ncol_mine = l_ncol(s);
            uint32_t my_count = 0;
            for (uint32_t Lr = 0; Lr < n; ++Lr) {   // uniform: every lane sees the same read
                const uint32_t bref = (uint32_t)__shfl((int)is.base_ref, (int)Lr, 64);
                const uint32_t blen = (uint32_t)__shfl((int)is.base_len, (int)Lr, 64);
                const uint32_t ncolL = (uint32_t)__shfl((int)ncol_mine, (int)Lr, 64);
                const uint32_t slotL = (uint32_t)__shfl((int)slot, (int)Lr, 64);
                const uint64_t off = shfl64(my_off, Lr);
                const bool fits = off + blen <= arena_cap;
                const glb_u32w rowL = (glb_u32w)p.spill + (uint64_t)(wave * S + slotL) * spill_cap;
                uint32_t total = 0;
                // lane i keeps (ref, len) of the read's i-th list: one row read per read, not per base block
                uint32_t cref = 0, clen = 0;
                if (lane < ncolL) {
                    const glb_u32w e = lane < LDS_CLASSES ? rowL + lane : rowL + LIST_ROW_HDR + 4 * (lane - LDS_CLASSES);
                    cref = e[0];
                    clen = lane < LDS_CLASSES ? e[4] : e[1];
                }
                for (uint32_t c = 0; c < blen; c += 64) {
                    const uint32_t j = c + lane;
                    const bool valid = j < blen;
                    const uint32_t v = valid ? ec[4ull * bref + 1 + j] : 0u;
                    bool ok = valid;
                    for (uint32_t i = 0; i < ncolL; ++i) {
                        uint32_t ref, len;
                        if (i < 64) {
                            ref = (uint32_t)__shfl((int)cref, (int)i, 64);
                            len = (uint32_t)__shfl((int)clen, (int)i, 64);
                        } else {
                            const glb_u32w e = rowL + LIST_ROW_HDR + 4 * (i - LDS_CLASSES);
                            ref = e[0];
                            len = e[1];
                        }
                        if (ref == bref) continue;   // uniform
                        bool hit = false;
                        if (len <= 64) {             // short list: scan it, no dependent loads
                            const glb_v4 rec = (glb_v4)(ec + 4ull * ref);
                            const uint32_t nchunks = (len + 4) >> 2;
                            for (uint32_t q0 = 0; q0 < nchunks; q0 += 4) {   // four loads in flight per round trip
                                u32x4 x[4];
#pragma unroll
                                for (uint32_t t = 0; t < 4; ++t) x[t] = rec[q0 + t < nchunks ? q0 + t : q0];
#pragma unroll
                                for (uint32_t t = 0; t < 4; ++t)
                                    if (q0 + t < nchunks) hit |= (q0 + t != 0 && x[t].x == v) | (x[t].y == v) | (x[t].z == v) | (x[t].w == v);
                            }
                        } else {                     // long list: binary_search (:404)
                            const glb_u32 ids = ec + 4ull * ref + 1;
                            uint32_t lo = 0, hi = len;
                            while (lo < hi) {
                                const uint32_t mid = (lo + hi) >> 1;
                                if (ids[mid] < v) lo = mid + 1; else hi = mid;
                            }
                            hit = lo < len && ids[lo] == v;
                        }
                        ok = ok && hit;
                    }
                    const uint64_t bm = __ballot(ok);
                    if (ok && fits) arena_g[off + total + rank_in(bm)] = v;
                    total += (uint32_t)__popcll(bm);
                }
                if (lane == Lr) my_count = total;
            }
            uint32_t ckey = NO_KEY;
            bool dfr = false;
            u32x4 d0{0u, 0u, 0u, 0u}, d1{0u, 0u, 0u, 0u};
            if (active) {
                const uint32_t cov_ = l_cov(s), mm_ = l_mism(s);
                ckey = emit_record<TRACE>(s, my_count, cnt_alloc, my_off, is.base_len, is.base_colour, gslot, kp, dfr);
                d0 = u32x4{s.rid, cov_, mm_, PA_DEFER_LIST | my_count};
                d1 = u32x4{(uint32_t)my_off, 0u, 0u, 0u};
            }
            if (counting) append_keys(ckey, lane, kp, kchunk);
            append_deferred(dfr, d0, d1, lane, kp, dchunk);
        } else {   // ST_F_LIGHT: list mode — pick a tier, intersect, write
            Isect is;
            is.count = 0;
            is.base_len = 0xFFFFFFFFu;
            is.base_ref = is.base_colour = 0;
            is.alive = 0;
            is.in_regs = false;
#pragma unroll
            for (int j = 0; j < 7; ++j) is.ids[j] = 0;
            bool emit_now = active;
            if (active) {
                const uint32_t tier = isect_pick(s, cols, is);
                if (tier == 0) isect_light(s, ix, cols, is);
                else {   // whole-wave steps: one list per lane (base <= 8 ids) or one base id per lane
                    l_set_st(s, tier == 1 ? ST_F_SCAN : ST_F_COOP);
                    emit_now = false;
                }
            }
            const uint32_t cntv2 = emit_now ? is.count : 0u;
            const uint32_t cnt_alloc = cntv2 == is.base_len ? 0u : cntv2;   // a result that is an index class is returned by reference
            const uint64_t my_off = arena_alloc(cnt_alloc, lane, kp, chunk);
            uint32_t ckey = NO_KEY;
            bool dfr = false;
            u32x4 d0{0u, 0u, 0u, 0u}, d1{0u, 0u, 0u, 0u};
            if (emit_now) {
                if (cnt_alloc && my_off + cnt_alloc <= p.arena_cap) {
                    const glb_u32w dst = (glb_u32w)p.arena + my_off;
                    const uint32_t alive = (uint32_t)is.alive;
#pragma unroll
                    for (int j = 0; j < 7; ++j)   // survivors straight from registers
                        if ((alive >> j) & 1u) dst[__popc(alive & ((1u << j) - 1))] = is.ids[j];
                }
                const uint32_t cov_ = l_cov(s), mm_ = l_mism(s);
                ckey = emit_record<TRACE>(s, cntv2, cnt_alloc, my_off, is.base_len, is.base_colour, gslot, kp, dfr);
                d0 = u32x4{s.rid, cov_, mm_, PA_DEFER_LIST | cntv2};
                d1 = u32x4{(uint32_t)my_off, 0u, 0u, 0u};
            }
            if (counting) append_keys(ckey, lane, kp, kchunk);
            append_deferred(dfr, d0, d1, lane, kp, dchunk);
        }

        const unsigned long long t_step = PA_DBG ? __builtin_readcyclecounter() : 0ull;
        PA_MARK("step_done");
        // ---- 4. store the lane state, push every slot onto the queue of its new state
        const uint32_t nq = active ? queue_of(s, K) : 0xFFu;
        if (active) {
            stA[slot] = u32x4{s.lk, s.cm, s.h, s.of};
            stB[slot] = u32x4{s.rr, s.rm, s.ph, s.nc};
        }
        if (active) sb[2 * (slot & 63u) + (slot >> 6)] = (uint8_t)nq;        // the slot's new state: one byte
        if (active2) sb[2 * (slot2 & 63u) + (slot2 >> 6)] = (uint8_t)nq2;
        if (PA_DBG && lane == 0) {   // [ST_ISECT] = pick + pop, [ST_NONE] = store + push (statistics only)
            const unsigned long long t_end = __builtin_readcyclecounter();
            dbg_clk[dual ? ST_DUAL : sel] += t_end - t_sec;
            dbg[ST_ISECT] += 1;
            dbg_clk[ST_ISECT] += t_pop - t_sec;
            dbg[ST_NONE] += 1;
            dbg_clk[ST_NONE] += t_end - t_step;
        }
        PA_MARK("stored");
    }
    {   // the unused tail of this wave's last chunk of deferred reads is padding
        asm volatile("" ::: "memory");
        const uint32_t cur = dchunk[0], end = dchunk[1];
        for (uint32_t i = cur + lane; i < end; i += 64) ((glb_v4w)p.defer)[2ull * i] = u32x4{NO_KEY, 0u, 0u, 0u};
    }
    if (counting) {   // the unused tail of this wave's last chunk of the key stream is padding
        asm volatile("" ::: "memory");
        const uint32_t cur = kchunk[0], end = kchunk[1];
        for (uint32_t i = cur + lane; i < end; i += 64) ((glb_u32w)p.keys)[i] = NO_KEY;
    }
    if (PA_DBG && lane < 2 * ST_NSTAT) atomicAdd(p.dbg + lane, (unsigned long long)dbg[lane]);
    if (PA_DBG && lane < ST_NSTAT) atomicAdd(p.dbg + 2 * ST_NSTAT + lane, dbg_clk[lane]);
#undef PA_CNT
#undef PA_POP
#undef PA_DBG
#undef PA_ABLATE
#undef p
}

size_t pool_slot_bytes(uint32_t wpr) { return 8 * (size_t)(wpr > PA_LDS_READ_WORDS ? 0 : wpr) + SLOT_FIXED_BYTES; }   // longer reads stay in HBM
size_t pool_fixed_bytes() { return POOL_FIXED; }
uint32_t pool_max_slots() { return POOL_MAX_SLOTS; }

size_t pool_lds_bytes(uint32_t wpr, uint32_t slots) {
    const size_t wave_bytes = (POOL_FIXED + (size_t)slots * pool_slot_bytes(wpr) + 15) & ~(size_t)15;
    return wave_bytes * (PA_MAP_BLOCK / 64);
}

template <bool TRACE, bool GREAD, bool DBG, bool S128 = false>
static int launch_one(const MapParams& p, uint32_t grid, size_t lds_bytes, hipStream_t stream) {
    const void* fn = reinterpret_cast<const void*>(&pa_map_pool_kernel<TRACE, GREAD, DBG, S128>);
    if (lds_bytes > 48 * 1024) {   // opt in to more than the default dynamic LDS limit
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((pa_map_pool_kernel<TRACE, GREAD, DBG, S128>), dim3(grid), dim3(PA_MAP_BLOCK), lds_bytes, stream, p);
    return (int)hipGetLastError();
}

int launch_map_pool(const MapParams& p, uint32_t grid, size_t lds_bytes, hipStream_t stream) {
    const bool gread = p.wpr > PA_LDS_READ_WORDS;
    if (p.trace) return gread ? launch_one<true, true, false>(p, grid, lds_bytes, stream) : launch_one<true, false, false>(p, grid, lds_bytes, stream);
    if (gread) return launch_one<false, true, false>(p, grid, lds_bytes, stream);
#ifdef PA_DEBUG_KNOBS   // the statistics / ablation instantiation exists in A/B builds only
    if (p.dbg || p.ablate) return launch_one<false, false, true>(p, grid, lds_bytes, stream);
#endif
    if (p.pool_slots == 128) return launch_one<false, false, false, true>(p, grid, lds_bytes, stream);
    return launch_one<false, false, false>(p, grid, lds_bytes, stream);
}

int pool_kernel_occupancy(size_t lds_bytes, int* blocks_per_cu) {
    return (int)hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, reinterpret_cast<const void*>(&pa_map_pool_kernel<false, false, false>),
                                                             PA_MAP_BLOCK, lds_bytes);
}

}  // namespace pa
