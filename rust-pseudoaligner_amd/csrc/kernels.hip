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
#include "kernel_utils.hpp"
#include "synth_common.hpp"

namespace pa {

// ---- state sections as separately register-allocated device functions -------------------------------------------
// Inlined into one loop body the four sections cost >100 VGPRs (the allocator keeps every section's temporaries alive
// around the scheduler loop); as real calls each section gets its own allocation (27-41 VGPRs) and the loop only
// carries the 9-register lane state, so the kernel fits 8 waves per SIMD without scratch spills. LDS pointers are
// passed with their address space so that the callee emits ds_* (not flat_*) accesses.
typedef __attribute__((address_space(3))) uint64_t* lds_u64;
typedef __attribute__((address_space(3))) uint32_t* lds_u32;
typedef __attribute__((address_space(1))) const uint8_t* glb_u8;     // global pointers keep their address space across the
typedef __attribute__((address_space(1))) const uint32_t* glb_u32;   // call so that the callee emits global_* (not flat_*) loads
typedef __attribute__((address_space(1))) uint32_t* glb_u32w;

__device__ __forceinline__ ReadRef make_read_ref(lds_u64 rdp, uint32_t wmax) { return ReadRef{(const uint64_t*)rdp, 64, wmax}; }
__device__ __forceinline__ ColRef make_col_ref(lds_u32 refs, glb_u32w spill_base, uint32_t slot, uint32_t spill_cap, glb_u32w trace_base) {
    return ColRef{(uint32_t*)refs, (uint32_t*)refs, (uint32_t*)(refs + 64 * LDS_CLASSES), (uint32_t*)(refs + 128 * LDS_CLASSES),
                  (uint32_t*)(spill_base + (uint64_t)slot * spill_cap), spill_cap,
                  trace_base ? (uint32_t*)(trace_base + (uint64_t)slot * spill_cap) : nullptr};
}

__device__ __attribute__((noinline)) Lane seek_call(Lane s, glb_u32 table, uint32_t nbuckets, uint64_t kmask, uint32_t k, lds_u64 rdp,
                                                    uint32_t wmax) {
    DevIndexView ix{};
    ix.table = (const uint32_t*)table;
    ix.nbuckets = nbuckets;
    ix.kmask = kmask;
    ix.k = k;
    seek_step(s, ix, make_read_ref(rdp, wmax));
    return s;
}

template <bool TRACE, int EXP = 0>
__device__ __attribute__((noinline)) Lane fwd_call(Lane s, glb_u8 blobs, glb_u32 nid_of_handle, uint32_t k, lds_u64 rdp, uint32_t wmax, lds_u32 refs,
                                                   glb_u32w spill_base, uint32_t slot, uint32_t spill_cap, glb_u32w trace_base,
                                                   uint32_t allowed) {
    DevIndexView ix{};
    ix.blobs = (const uint8_t*)blobs;
    ix.nid_of_handle = (const uint32_t*)nid_of_handle;
    ix.k = k;
    fwd_step<TRACE, EXP>(s, ix, make_read_ref(rdp, wmax), make_col_ref(refs, spill_base, slot, spill_cap, trace_base), allowed);
    return s;
}

template <bool TRACE>
__device__ __attribute__((noinline)) Lane left_call(Lane s, glb_u8 blobs, glb_u32 ledge, glb_u32 nid_of_handle, uint32_t k, lds_u64 rdp, uint32_t wmax,
                                                    lds_u32 refs, glb_u32w spill_base, uint32_t slot, uint32_t spill_cap,
                                                    glb_u32w trace_base, uint32_t allowed) {
    DevIndexView ix{};
    ix.blobs = (const uint8_t*)blobs;
    ix.ledge = (const uint32_t*)ledge;
    ix.nid_of_handle = (const uint32_t*)nid_of_handle;
    ix.k = k;
    left_step<TRACE>(s, ix, make_read_ref(rdp, wmax), make_col_ref(refs, spill_base, slot, spill_cap, trace_base), allowed);
    return s;
}

// ---- finishing states: nodes_to_eq_class + output -------------------------------------------------------------------
// A lane whose walk has ended is sorted into one of five finishing states by what its intersection needs (isect_pick):
// LIGHT (registers only), SCAN (base in registers, other lists streamed), COOP (whole wave per read), COPY (single
// class) or NONE (unmapped). Like the walk states they are scheduled by population, so a rare expensive case (a long
// class list, a strict-subset result that needs the class hash table) never stalls 63 cheap lanes: its lane simply
// waits until enough lanes of the same kind have gathered. Every finishing call runs with the whole wave active.
typedef __attribute__((address_space(3))) const MapParams* lds_params;
typedef __attribute__((address_space(3))) unsigned long long* lds_u64w;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u32x4* glb_v4w;
typedef __attribute__((address_space(1))) const u32x4* glb_v4;
typedef __attribute__((address_space(1))) unsigned long long* glb_u64w;

__device__ __forceinline__ uint64_t shfl64(uint64_t v, uint32_t src) {
    return ((uint64_t)(uint32_t)__shfl((int)(v >> 32), (int)src, 64) << 32) | (uint32_t)__shfl((int)v, (int)src, 64);
}

// wave-uniform: reserve cnt_alloc arena entries per lane out of the wave's private chunk; returns this lane's offset
__device__ __forceinline__ uint64_t arena_alloc(uint32_t cnt_alloc, uint32_t lane, lds_params pp, lds_u64w chunk) {
    const uint32_t incl = wave_incl_scan(cnt_alloc);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    unsigned long long chunk_cur = chunk[0];
    if (total > 0) {
        if (chunk_cur + total > chunk[1]) {   // take a new private slice of the class arena (one global atomic per chunk)
            const unsigned long long want = total > PA_ARENA_CHUNK ? total : PA_ARENA_CHUNK;
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd((unsigned long long*)pp->arena_top, want);
            base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            chunk_cur = base;
            if (lane == 0) chunk[1] = base + want;
        }
        if (lane == 0) chunk[0] = chunk_cur + total;
    }
    return chunk_cur + (incl - cnt_alloc);
}

// record + class-count update of one finished read; returns the lane's next state (EMPTY, or F_NOVEL when the class of a
// strict-subset result still has to be looked up by content before it can be counted)
template <bool TRACE>
__device__ __forceinline__ Lane emit_record(Lane s, bool mapped, uint32_t cnt, uint32_t cnt_alloc, uint64_t my_off, uint32_t base_len,
                                            uint32_t base_colour, uint32_t slot, lds_params pp, glb_u32w results_g, glb_u32w counts_g) {
    pa_read_result r{0, 0, 0, 0};
    uint32_t colour = 0xFFFFFFFFu;
    bool novel = false;
    if (mapped) {
        r.coverage = l_cov(s);
        r.mismatches = l_mism(s) | PA_MAPPED_BIT;
        r.class_len = cnt;
        r.class_off = (uint32_t)my_off;
        if (my_off + cnt_alloc > pp->arena_cap) atomicOr(pp->status, PA_STATUS_ARENA_FULL);
        if (cnt == base_len) {   // the class IS index class base_colour: returned by reference, nothing was written to the arena
            colour = base_colour;
            r.class_off = PA_CLASS_REF | base_colour;
        } else novel = cnt != 0;
        if (l_flags(s) & F_SPILL_OVERFLOW) atomicOr(pp->status, PA_STATUS_SPILL_OVERFLOW);
    }
    ((glb_v4w)results_g)[s.rid] = u32x4{r.coverage, r.mismatches, r.class_off, r.class_len};
    if (TRACE) {   // node lists, read-major, stride spill_cap (map_read_to_nodes test surface)
        const uint32_t spill_cap = pp->spill_cap;
        const uint32_t nt = l_ntrace(s);
        const uint32_t nn = mapped ? (nt < spill_cap ? nt : spill_cap) : 0;
        ((glb_u32w)pp->nodes_len)[s.rid] = mapped ? nt : 0;
        const glb_u32w tr = (glb_u32w)pp->trace + (uint64_t)slot * spill_cap;
        const glb_u32w out = (glb_u32w)pp->nodes_out + (uint64_t)s.rid * spill_cap;
        for (uint32_t j = 0; j < nn; ++j) out[j] = tr[j];
    }
    const glb_u32w colour_out = (glb_u32w)pp->colour_out;
    const bool want_class = counts_g != nullptr || colour_out != nullptr;
    if (novel && want_class && !(pp->ablate & 128u) && my_off + cnt_alloc <= pp->arena_cap) {   // defer the content lookup to the F_NOVEL state
        s.h = (uint32_t)my_off;
        s.rr = cnt;
        l_set_st(s, ST_F_NOVEL);
        return s;
    }
    if (colour_out) colour_out[s.rid] = colour;
    if (counts_g) {   // fused class-count table: fire-and-forget atomic
        const uint32_t num_classes = pp->ix.num_classes;
        const uint32_t cslot = !mapped ? num_classes + 2 : cnt == 0 ? num_classes + 1 : colour == 0xFFFFFFFFu ? num_classes : colour;
        atomicAdd((unsigned long long*)(glb_u64w)counts_g + cslot, 1ull);
    }
    s.lk = 0;   // ST_EMPTY
    return s;
}

// LIGHT / SCAN intersections and the emit step are three separate calls so that each stays within the caller-saved
// VGPRs (<= 56 at 6 waves per SIMD): a callee that needs more must save/restore callee-saved registers through scratch
// on EVERY call, which measured as 5 GB of extra HBM writes per 10 M reads.
__device__ __attribute__((noinline)) Isect light_call(uint32_t nc, lds_u32 refs_lane, glb_u32 ec) {
    DevIndexView ix{};
    ix.ec = (const uint32_t*)ec;
    Lane s{};
    s.nc = nc;
    const ColRef cols = make_col_ref(refs_lane, nullptr, 0, 0, nullptr);   // LIGHT never touches spilled classes
    Isect is;
    isect_pick(s, cols, is);
    isect_light(s, ix, cols, is);
    return is;
}

__device__ __attribute__((noinline)) Isect scan_call(uint32_t nc, lds_u32 refs_lane, glb_u32w spill_base, uint32_t slot, uint32_t spill_cap,
                                                     glb_u32 ec) {
    DevIndexView ix{};
    ix.ec = (const uint32_t*)ec;
    Lane s{};
    s.nc = nc;
    const ColRef cols = make_col_ref(refs_lane, spill_base, slot, spill_cap, nullptr);
    Isect is;
    isect_pick(s, cols, is);
    isect_scan(s, ix, cols, is);
    return is;
}

// arena allocation + class ids + record + count for the lanes in `mine` (whole wave enters: the scan needs every lane)
template <bool TRACE>
__device__ __attribute__((noinline)) Lane emit_call(Lane s, bool mine, bool mapped, uint32_t count, uint32_t base_len, uint32_t base_colour,
                                                    uint32_t base_ref, uint32_t alive, uint32_t mode, uint32_t id0, uint32_t id1, uint32_t id2,
                                                    uint32_t id3, uint32_t id4, uint32_t id5, uint32_t id6, uint32_t lane, uint32_t slot,
                                                    lds_params pp, lds_u64w chunk) {
    const uint32_t cnt = mine ? count : 0u;
    const uint32_t cnt_alloc = cnt == base_len ? 0u : cnt;   // a result that is an index class is returned by reference
    const uint64_t my_off = arena_alloc(cnt_alloc, lane, pp, chunk);
    if (!mine) return s;
    const glb_u32w arena_g = (glb_u32w)pp->arena;
    if (cnt_alloc && my_off + cnt_alloc <= pp->arena_cap) {
        const glb_u32w dst = arena_g + my_off;
        if (mode == 2) {   // window mode: ids = window base + bit positions
            uint32_t k = 0;
            for (uint32_t t = alive; t; t &= t - 1) dst[k++] = id0 + (uint32_t)(__ffs((int)t) - 1);
        } else if (mode == 1) {   // LIGHT tier: survivors straight from registers
            const uint32_t ids[7] = {id0, id1, id2, id3, id4, id5, id6};
#pragma unroll
            for (int j = 0; j < 7; ++j)
                if ((alive >> j) & 1u) dst[__popc(alive & ((1u << j) - 1))] = ids[j];
        } else {         // SCAN tier: base of <= 8 ids, re-read from its record
            const glb_u32 bids = (glb_u32)pp->ix.ec + 4ull * base_ref + 1;
            uint32_t k = 0;
            for (uint32_t t = alive; t; t &= t - 1) dst[k++] = bids[__ffs((int)t) - 1];
        }
    }
    return emit_record<TRACE>(s, mapped, cnt, cnt_alloc, my_off, base_len, base_colour, slot, pp, (glb_u32w)pp->results, (glb_u32w)pp->counts);
}

// COOP: the whole wave works on one read at a time (base list of more than 8 ids and at least two classes). Lane e owns
// base ids e, e+64, ...; membership in every other list is a scan of 16-byte loads (short lists) or a binary search;
// survivors are compacted with a wave ballot straight into the read's arena slice (base_len entries were reserved).
template <bool TRACE>
__device__ __attribute__((noinline)) Lane fin_coop_call(Lane s, uint32_t lane, uint32_t slot, lds_u32 refs_lane, lds_params pp, lds_u64w chunk,
                                                        glb_u32 ec, glb_u32w arena_g, glb_u32w results_g, glb_u32w counts_g) {
    const bool mine = l_st(s) == ST_F_COOP;
    const glb_u32w spill_base = (glb_u32w)pp->spill;
    const uint32_t spill_cap = pp->spill_cap;
    const ColRef cols = make_col_ref(refs_lane, spill_base, slot, spill_cap, nullptr);
    Isect is;
    is.count = 0;
    is.base_len = 0;
    is.base_ref = 0;
    is.base_colour = 0;
    if (mine) isect_pick(s, cols, is);
    const uint32_t cnt_alloc = mine ? is.base_len : 0u;   // upper bound: the survivors are a subset of the base list
    const uint64_t my_off = arena_alloc(cnt_alloc, lane, pp, chunk);
    const uint64_t arena_cap = pp->arena_cap;
    const lds_u32 wave_refs = refs_lane - lane * LDS_CLASSES;
    const uint32_t ncol_mine = l_ncol(s);
    uint32_t my_count = 0;
    uint64_t mask = __ballot(mine);
    while (mask) {
        const uint32_t L = (uint32_t)(__ffsll((unsigned long long)mask) - 1);
        mask &= mask - 1;
        // cross-lane reads with every lane active
        const uint32_t bref = (uint32_t)__shfl((int)is.base_ref, (int)L, 64);
        const uint32_t blen = (uint32_t)__shfl((int)is.base_len, (int)L, 64);
        const uint32_t ncolL = (uint32_t)__shfl((int)ncol_mine, (int)L, 64);
        const uint64_t off = shfl64(my_off, L);
        const bool fits = off + blen <= arena_cap;
        const lds_u32 refsL = wave_refs + L * LDS_CLASSES;
        uint32_t total = 0;
        for (uint32_t c = 0; c < blen; c += 64) {   // uniform bounds: every lane sees the same read
            const uint32_t j = c + lane;
            const bool valid = j < blen;
            const uint32_t v = valid ? ec[4ull * bref + 1 + j] : 0u;
            bool ok = valid;
            for (uint32_t i = 0; i < ncolL; ++i) {
                uint32_t ref, len;
                if (i < LDS_CLASSES) {
                    ref = refsL[i];
                    len = refsL[64 * LDS_CLASSES + i];
                } else {
                    const glb_u32w sp = spill_base + (uint64_t)(slot - lane + L) * spill_cap + 4 * (i - LDS_CLASSES);
                    ref = sp[0];
                    len = sp[1];
                }
                if (ref == bref) continue;   // uniform
                bool hit = false;
                if (len <= 64) {             // short list: scan it, no dependent loads
                    const glb_v4 rec = (glb_v4)(ec + 4ull * ref);
                    const uint32_t nchunks = (len + 4) >> 2;
#pragma unroll 2
                    for (uint32_t q = 0; q < nchunks; ++q) {
                        const u32x4 w = rec[q];
                        hit |= (q != 0 && w.x == v) | (w.y == v) | (w.z == v) | (w.w == v);
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
            if (ok && fits) arena_g[off + total + (uint32_t)__popcll(bm & ((1ull << lane) - 1))] = v;
            total += (uint32_t)__popcll(bm);
        }
        if (lane == L) my_count = total;
    }
    if (!mine) return s;
    return emit_record<TRACE>(s, true, my_count, cnt_alloc, my_off, is.base_len, is.base_colour, slot, pp, results_g, counts_g);
}

// NOVEL: the result is a strict subset of every visited class; find out whether it equals some index class (content
// lookup in the class-list hash table), then count it
__device__ __attribute__((noinline)) Lane fin_novel_call(Lane s, lds_params pp, glb_u32 ec, glb_u32w arena_g, glb_u32w counts_g) {
    if (l_st(s) != ST_F_NOVEL) return s;
    DevIndexView ix{};
    ix.ec = (const uint32_t*)ec;
    ix.class_ref = (const uint32_t*)(glb_u32)pp->ix.class_ref;
    ix.class_len = (const uint32_t*)(glb_u32)pp->ix.class_len;
    const uint32_t colour = class_of_list((const uint32_t*)arena_g + s.h, s.rr, ix, (const uint32_t*)(glb_u32)pp->class_table, pp->class_table_size);
    const glb_u32w colour_out = (glb_u32w)pp->colour_out;
    if (colour_out) colour_out[s.rid] = colour;
    if (counts_g) atomicAdd((unsigned long long*)(glb_u64w)counts_g + (colour == 0xFFFFFFFFu ? pp->ix.num_classes : colour), 1ull);
    s.lk = 0;   // ST_EMPTY
    return s;
}

constexpr uint32_t PA_LDS_PARAMS_BYTES = (sizeof(MapParams) + 15) / 16 * 16;
constexpr uint32_t PA_LDS_WAVE_FIXED = 256;   // per-wave: arena chunk state {cur, end} (16 B), scheduler statistics [2*ST_COUNT] u32 + section clocks [ST_COUNT] u64

template <bool TRACE, int WAVES>
__global__ __launch_bounds__(PA_MAP_BLOCK, WAVES) void pa_map_kernel(const MapParams p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t lane = lane_id();
    const uint32_t wave_in_block = threadIdx.x >> 6;
    const uint32_t waves_per_block = PA_MAP_BLOCK / 64;
    const uint32_t wave = blockIdx.x * waves_per_block + wave_in_block;
    const uint32_t nwaves = gridDim.x * waves_per_block;
    const uint32_t slot = wave * 64 + lane;   // this lane's slice of the spill / trace scratch

    // LDS: [kernel parameters][per wave: arena chunk state + statistics | read tile (wpr+1 words x 64 lanes) | class refs | class lens]
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&p);
        uint32_t* dst = reinterpret_cast<uint32_t*>(smem);
        for (uint32_t i = threadIdx.x; i < sizeof(MapParams) / 4; i += PA_MAP_BLOCK) dst[i] = src[i];
    }
    const uint32_t wave_bytes = PA_LDS_WAVE_FIXED + (p.wpr + 1) * 512 + 3 * LDS_CLASSES * 256;
    uint8_t* const wbase = smem + PA_LDS_PARAMS_BYTES + wave_in_block * wave_bytes;
    const lds_u64w chunk = (lds_u64w)wbase;
    const lds_u64 rd_lane = (lds_u64)(wbase + PA_LDS_WAVE_FIXED) + lane;
    const lds_u32 refs_lane = (lds_u32)(wbase + PA_LDS_WAVE_FIXED + (p.wpr + 1) * 512) + lane * LDS_CLASSES;
    const lds_params pp = (lds_params)smem;
    const lds_u32 dbg = (lds_u32)(wbase + 16);                            // [0..ST_COUNT) iterations, [ST_COUNT..2*ST_COUNT) lanes served
    const lds_u64w dbg_clk = (lds_u64w)(wbase + 16 + 8 * ST_COUNT + 8);   // wall ticks per state (8-byte aligned)
    if (lane < 60) ((lds_u32)wbase)[lane] = 0;
    __syncthreads();

    // static partition of the tiles over the waves (no global work queue: one hot atomic would cap the rate)
    const uint64_t ntiles = (p.n_reads + 63) >> 6;
    uint64_t next = (ntiles * wave / nwaves) << 6;
    uint64_t end = (ntiles * (wave + 1) / nwaves) << 6;
    if (end > p.n_reads) end = p.n_reads;
    if (next > end) next = end;

    const glb_u32 ec = (glb_u32)p.ix.ec;
    const glb_u32w arena_g = (glb_u32w)p.arena, results_g = (glb_u32w)p.results, counts_g = (glb_u32w)p.counts;

    Lane s;
    s.rid = s.lk = s.cm = s.h = s.of = s.rr = s.rm = s.ph = s.nc = 0;   // state ST_EMPTY

    const bool from_list = false;

    // ---- population-scheduled state machine ----------------------------------------------------------------------------
    for (;;) {
        const uint32_t st = l_st(s);
        // population of every state. The cheap finishing states (NONE, LIGHT, COPY) are one section; the common states
        // compete by population; the rare expensive ones (SCAN, COOP, NOVEL) wait until enough of their kind have gathered
        // (or nothing else can run), so that they neither stall cheap lanes nor starve.
        const uint64_t mE = __ballot(st == ST_EMPTY);
        const uint64_t left = end - next;
        const uint32_t nE = __popcll(mE);
        const uint32_t nR = (uint32_t)(left < (uint64_t)nE ? left : (uint64_t)nE);
        const uint32_t nS = __popcll(__ballot(st == ST_SEEK)), nF = __popcll(__ballot(st == ST_FWD)), nL = __popcll(__ballot(st == ST_LEFT));
        const uint32_t nFast = __popcll(__ballot(st == ST_NONE || st == ST_F_LIGHT || st == ST_F_BITS));
        const uint32_t nScan = __popcll(__ballot(st == ST_F_SCAN)), nCoop = __popcll(__ballot(st == ST_F_COOP)), nNovel = __popcll(__ballot(st == ST_F_NOVEL));
        uint32_t best = nR, sel = ST_EMPTY;
        if (nS > best) { best = nS; sel = ST_SEEK; }
        if (nF > best) { best = nF; sel = ST_FWD; }
        if (nFast > best) { best = nFast; sel = ST_F_LIGHT; }
        if (nL > best) { best = nL; sel = ST_LEFT; }
        if (nCoop >= p.thr_coop) { best = nCoop; sel = ST_F_COOP; }
        else if (nScan >= p.thr_scan) { best = nScan; sel = ST_F_SCAN; }
        else if (nNovel >= p.thr_novel) { best = nNovel; sel = ST_F_NOVEL; }
        else if (best < p.thr_idle) {   // little common work left: drain the rare states
            if (nScan > best) { best = nScan; sel = ST_F_SCAN; }
            if (nNovel > best) { best = nNovel; sel = ST_F_NOVEL; }
            if (nCoop > best) { best = nCoop; sel = ST_F_COOP; }
        }
        if (best == 0) break;
        if (p.dbg && lane == 0) {
            dbg[sel] += 1;
            dbg[ST_COUNT + sel] += best;
        }
        const unsigned long long t_sec = p.dbg ? __builtin_readcyclecounter() : 0ull;

        if (sel == ST_EMPTY) {   // ---- REFILL: empty lanes take the next reads of this wave's range (coalesced by rank)
            const uint32_t rank = __popcll(mE & ((1ull << lane) - 1));
            if (st == ST_EMPTY && rank < nR) {
                const uint64_t rid = from_list ? (uint64_t)((glb_u32w)p.slow)[next + rank] : next + rank;
                uint32_t L = p.lens[rid];
                if (L > p.wpr * 32) L = p.wpr * 32;
                const uint64_t* src = p.tiles + ((rid >> 6) * p.wpr) * 64 + (rid & 63);
                for (uint32_t w = 0; w < p.wpr; ++w) rd_lane[w * 64] = src[(uint64_t)w * 64];
                rd_lane[p.wpr * 64] = 0;
                lane_start(s, (uint32_t)rid, L, p.ix.k);
            }
            next += nR;
        } else if (sel == ST_SEEK) {
            if (st == ST_SEEK && (p.ablate & 4u)) { s.nc |= 1; l_set_st(s, (p.ablate & 2u) ? ST_ISECT : ST_FWD); s.h = s.rid & 1023u; l_or_flags(s, F_FRESH); }
            else if (st == ST_SEEK) s = seek_call(s, (glb_u32)p.ix.table, (uint32_t)p.ix.nbuckets, p.ix.kmask, p.ix.k, rd_lane, p.wpr);
        } else if (sel == ST_FWD) {
            if (st == ST_FWD && (p.ablate & 2u)) l_set_st(s, ST_ISECT);
            else if (st == ST_FWD) s = fwd_call<TRACE>(s, (glb_u8)p.ix.blobs, (glb_u32)p.ix.nid_of_handle, p.ix.k, rd_lane, p.wpr, refs_lane, (glb_u32w)p.spill, slot, p.spill_cap, TRACE ? (glb_u32w)p.trace : nullptr, p.allowed);
        } else if (sel == ST_LEFT) {
            if (st == ST_LEFT) s = left_call<TRACE>(s, (glb_u8)p.ix.blobs, (glb_u32)p.ix.ledge, (glb_u32)p.ix.nid_of_handle, p.ix.k, rd_lane, p.wpr, refs_lane, (glb_u32w)p.spill, slot, p.spill_cap, TRACE ? (glb_u32w)p.trace : nullptr, p.allowed);
        } else if (sel == ST_F_COOP) {
            s = fin_coop_call<TRACE>(s, lane, slot, refs_lane, pp, chunk, ec, arena_g, results_g, counts_g);
        } else if (sel == ST_F_NOVEL) {
            s = fin_novel_call(s, pp, ec, arena_g, counts_g);
        } else {   // ST_F_LIGHT (= NONE + BITS + LIGHT) or ST_F_SCAN
            const bool mine = sel == ST_F_SCAN ? st == ST_F_SCAN : (st == ST_NONE || st == ST_F_LIGHT || st == ST_F_BITS);
            Isect is;
            is.count = 0;
            is.base_len = 0xFFFFFFFFu;
            is.base_ref = is.base_colour = 0;
            is.alive = 0;
            is.in_regs = false;
            for (int j = 0; j < 7; ++j) is.ids[j] = 0;
            uint32_t mode = 0;
            if (st == ST_F_BITS && sel == ST_F_LIGHT) {   // window mode: the class is already in LDS as {base, mask, class id}
                const u32x4 w = *(__attribute__((address_space(3))) const u32x4*)refs_lane;
                is.count = (uint32_t)__popc(w.y);
                is.base_len = w.z != NO_CLASS ? is.count : 0xFFFFFFFFu;   // by reference iff the window IS a class that was seen
                is.base_colour = w.z;
                is.alive = w.y;
                is.ids[0] = w.x;
                mode = 2;
            }
            if (!(p.ablate & 1u)) {
                if (st == ST_F_LIGHT && sel == ST_F_LIGHT) is = light_call(s.nc, refs_lane, ec);
                else if (st == ST_F_SCAN && sel == ST_F_SCAN) is = scan_call(s.nc, refs_lane, (glb_u32w)p.spill, slot, p.spill_cap, ec);
            }
            s = emit_call<TRACE>(s, mine, st != ST_NONE, is.count, is.base_len, is.base_colour, is.base_ref, (uint32_t)is.alive, mode == 2 ? 2u : is.in_regs ? 1u : 0u,
                                 is.ids[0], is.ids[1], is.ids[2], is.ids[3], is.ids[4], is.ids[5], is.ids[6], lane, slot, pp, chunk);
        }
        if (l_st(s) == ST_ISECT) {   // the walk just ended: choose how this read's classes will be intersected (LDS only)
            if (!(l_flags(s) & F_LISTS)) l_set_st(s, ST_F_BITS);   // window mode: nothing left to intersect
            else {
                Isect tmp;
                const ColRef cols = make_col_ref(refs_lane, (glb_u32w)p.spill, slot, p.spill_cap, nullptr);
                const uint32_t tier = (p.ablate & 1u) ? 0u : isect_pick(s, cols, tmp);
                l_set_st(s, (tier == 0 || (p.ablate & 64u)) ? ST_F_LIGHT : tier == 1 ? ST_F_SCAN : ST_F_COOP);   // ablate 64: everything LIGHT (wrong)
            }
        }
        if (p.dbg && lane == 0) dbg_clk[sel] += __builtin_readcyclecounter() - t_sec;
    }
    if (p.dbg && lane < 2 * ST_COUNT) atomicAdd(p.dbg + lane, (unsigned long long)dbg[lane]);
    if (p.dbg && lane < ST_COUNT) atomicAdd(p.dbg + 2 * ST_COUNT + lane, dbg_clk[lane]);
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
// WAVES = waves per SIMD the register allocator must leave room for (launch-bounds variant; tuning knob PA_MAP_WAVES)
template <bool TRACE, int WAVES>
static int launch_map_variant(const MapParams& p, uint32_t grid, size_t lds_bytes, hipStream_t stream) {
    if (lds_bytes > 48 * 1024) {   // long-read tiles: opt in to more than the default dynamic LDS limit
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pa_map_kernel<TRACE, WAVES>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((pa_map_kernel<TRACE, WAVES>), dim3(grid), dim3(PA_MAP_BLOCK), lds_bytes, stream, p);
    return (int)hipGetLastError();
}

// Only ONE launch-bounds variant of the production kernel is instantiated: the state functions are shared by every kernel
// that calls them and are register-allocated for the most restrictive caller (an 8-waves variant forces them into 64
// VGPRs and ~150 bytes/lane of scratch, which showed up as 5 GB of HBM writes per 10 M reads).
int launch_map(const MapParams& p, uint32_t grid, size_t lds_bytes, int waves, hipStream_t stream) {
    (void)waves;
    if (p.trace) return launch_map_variant<true, 4>(p, grid, lds_bytes, stream);
    return launch_map_variant<false, PA_DEFAULT_MAP_WAVES>(p, grid, lds_bytes, stream);
}

int map_kernel_occupancy(size_t lds_bytes, int waves, int* blocks_per_cu) {
    (void)waves;
    return (int)hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, reinterpret_cast<const void*>(&pa_map_kernel<false, PA_DEFAULT_MAP_WAVES>),
                                                             PA_MAP_BLOCK, lds_bytes);
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

int launch_count(const pa_read_result* results, const uint32_t* arena, const uint32_t* colour, uint64_t n, const DevIndexView& ix,
                 const uint32_t* class_table, uint64_t class_table_size, unsigned long long* counts, hipStream_t stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(pa_count_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, results, arena, colour, n, ix,
                       class_table, class_table_size, counts);
    return (int)hipGetLastError();
}

}  // namespace pa
