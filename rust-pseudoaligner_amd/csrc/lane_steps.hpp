// Per-lane state machine of the mapping kernel: one lane = one read, one call = one memory-dependent step.
//
// This is the reference's map_read_to_nodes_with_mismatch (src/pseudoaligner.rs:64-319) re-expressed for a
// 64-wide wavefront: the three data-dependent loops of the reference (dictionary scan :91-114, left extension
// :131-203, forward extension :209-301) become the states SEEK / LEFT / FWD, each advancing by one dependent HBM
// fetch per call, so that a wave can run the same state for many reads at once (kernels.hip schedules the states).
// The per-base comparison loops (:151-170, :236-255) are replaced by XOR + popcount/ctz on 32-base windows; the
// nodes Vec (:219) is replaced by the set of DISTINCT colours seen, because nodes_to_eq_class (:323-356) only uses
// the colour lists and intersection is idempotent/commutative.
//
// Compiled for gfx950 by kernels.hip and for the host by tests/emu (CPU parity tests of exactly this text).
#pragma once
#include "device_layout.hpp"

namespace pa {

enum : uint32_t { ST_EMPTY = 0, ST_SEEK = 1, ST_FWD = 2, ST_LEFT = 3, ST_ISECT = 4, ST_NONE = 5 };
enum : uint32_t { F_FRESH = 1u, F_FIRST_SEEK = 2u, F_LEFT_SEED = 4u, F_SPILL_OVERFLOW = 8u, F_PROBE_SHIFT = 8 };

struct Lane {
    uint32_t st, rid, L;
    uint32_t kp;         // kmer_pos (:79)
    uint32_t cov, mism;  // read_coverage, mismatch_count (:71-72)
    uint32_t h, off;     // node_id / kmer_offset of the forward search (:118-121), h = blob handle
    uint32_t ro;         // FWD: ref offset inside the node; LEFT: node bases still to the left (prev_kmer_offset + 1)
    uint32_t rem;        // bases of max_matchable_pos not yet compared in this node visit (:145, :231)
    uint32_t snp;        // seen_snp of this node visit (:150, :235)
    uint32_t ra;         // LEFT: read bases still to the left (last_pos + 1)
    uint32_t ph;         // LEFT: prev_node_id (:128)
    uint32_t ncol;       // distinct colours collected
    uint32_t flags;
    uint32_t ntrace;     // TRACE builds only: nodes.len()
};

struct ReadRef {   // the lane's packed read: word w at p[w * stride]; word ceil(L/32) must be readable
    const uint64_t* p;
    uint32_t stride;
};

struct ColRef {    // the lane's colour list: first `cap` entries at p[i*stride] (LDS), the rest in `spill` (HBM)
    uint32_t* p;
    uint32_t stride, cap;
    uint32_t* spill;
    uint32_t spill_cap;
    uint32_t* trace;   // TRACE builds only: node ids in visit order (map_read_to_nodes, :54-61), capacity spill_cap
};

struct Hdr {
    uint32_t len, exts, colour, nid, e0, e1, e2, e3;
};

// ---------------------------------------------------------------------------------------------- helpers
PA_HD uint64_t pa_mix64(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

PA_HD uint64_t pa_mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

PA_HD uint32_t pa_popc64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__popcll(x);
#else
    return (uint32_t)__builtin_popcountll(x);
#endif
}

PA_HD uint32_t pa_ctz64(uint64_t x) {   // x != 0
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)(__ffsll((unsigned long long)x) - 1);
#else
    return (uint32_t)__builtin_ctzll(x);
#endif
}

PA_HD uint64_t pa_brev64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brevll(x);
#else
    x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0f0f0f0f0f0f0f0full) | ((x & 0x0f0f0f0f0f0f0f0full) << 4);
    return __builtin_bswap64(x);
#endif
}

PA_HD uint32_t pa_min(uint32_t a, uint32_t b) { return a < b ? a : b; }

PA_HD uint64_t funnel(uint64_t lo, uint64_t hi, uint32_t sh) { return sh ? (lo >> sh) | (hi << (64 - sh)) : lo; }

// 32 bases of the read starting at base `pos`
PA_HD uint64_t read_window(ReadRef r, uint32_t pos) {
    const uint32_t w = pos >> 5;
    return funnel(r.p[w * r.stride], r.p[(w + 1) * r.stride], (pos & 31) * 2);
}
// 32 bases ENDING at base p (base p lands in the top 2 bits; missing low bases are zero)
PA_HD uint64_t read_window_end(ReadRef r, uint32_t p) { return p >= 31 ? read_window(r, p - 31) : r.p[0] << (2 * (31 - p)); }

PA_HD uint32_t read_base(ReadRef r, uint32_t pos) { return (uint32_t)(r.p[(pos >> 5) * r.stride] >> ((pos & 31) * 2)) & 3u; }

PA_HD Hdr load_hdr(const DevIndexView& ix, uint32_t h) {
    const U4* p = reinterpret_cast<const U4*>(ix.blobs + (uint64_t)h * BLOB_GRANULE);
    const U4 a = p[0], b = p[1];
    return Hdr{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
}
PA_HD const uint64_t* node_seq(const DevIndexView& ix, uint32_t h) {
    return reinterpret_cast<const uint64_t*>(ix.blobs + (uint64_t)h * BLOB_GRANULE + 32);
}

// mismatch mask of a 32-base XOR: bit 2i set <=> base i differs
PA_HD uint64_t diff_mask(uint64_t x) { return (x | (x >> 1)) & 0x5555555555555555ull; }

// The body of the compare loops (:151-170 / :236-255) over n <= 32 bases given their mismatch mask (bit 2i = i-th base
// compared). Returns matched_bases for the chunk; updates seen_snp / mismatch_count; sets premature.
PA_HD uint32_t compare_chunk(uint64_t m, uint32_t n, uint32_t allowed, uint32_t& snp, uint32_t& mism, bool& premature) {
    if (n < 32) m &= (1ull << (2 * n)) - 1;
    const uint32_t cnt = pa_popc64(m);
    if (snp + cnt <= allowed) {
        snp += cnt;
        mism += cnt;
        return n;
    }
    const uint32_t tolerated = allowed - snp;   // mismatches of this chunk that are still within budget
    for (uint32_t i = 0; i < tolerated; ++i) m &= m - 1;
    mism += tolerated + 1;                      // the breaking base is counted (:158) but not matched (:162-165)
    snp = allowed + 1;
    premature = true;
    return pa_ctz64(m) >> 1;
}

template <bool TRACE>
PA_HD void push_node(Lane& s, ColRef c, uint32_t colour, uint32_t nid) {
    if (TRACE) {
        if (s.ntrace < c.spill_cap) c.trace[s.ntrace] = nid;
        s.ntrace += 1;
    }
    const uint32_t inl = pa_min(s.ncol, c.cap);
    for (uint32_t i = 0; i < inl; ++i)
        if (c.p[i * c.stride] == colour) return;
    if (s.ncol < c.cap) c.p[s.ncol * c.stride] = colour;
    else if (s.ncol - c.cap < c.spill_cap) c.spill[s.ncol - c.cap] = colour;
    else { s.flags |= F_SPILL_OVERFLOW; return; }
    s.ncol += 1;
}
PA_HD uint32_t get_colour(ColRef c, uint32_t i) { return i < c.cap ? c.p[i * c.stride] : c.spill[i - c.cap]; }

PA_HD void lane_start(Lane& s, uint32_t rid, uint32_t L, uint32_t k) {
    s.rid = rid;
    s.L = L;
    s.kp = 0;
    s.cov = 0;
    s.mism = 0;
    s.ncol = 0;
    s.flags = F_FIRST_SEEK;
    s.ntrace = 0;
    s.h = s.off = s.ro = s.rem = s.snp = s.ra = s.ph = 0;
    s.st = L < k ? ST_NONE : ST_SEEK;   // :82-84
}

// ---------------------------------------------------------------------------------------------- SEEK
// One dictionary probe of find_kmer_match (:91-114): dbg_index.get + verification collapse into one bucket fetch.
PA_HD void seek_step(Lane& s, const DevIndexView& ix, ReadRef rd) {
    const uint32_t K = ix.k;
    const uint32_t last = s.L - K;                                  // last_kmer_pos (:86)
    const uint64_t kmer = read_window(rd, s.kp) & ix.kmask;         // read_seq.get_kmer(kmer_pos) (:93)
    const uint32_t probe = s.flags >> F_PROBE_SHIFT;
    uint64_t b = pa_mulhi64(pa_mix64(kmer), ix.nbuckets) + probe;
    if (b >= ix.nbuckets) b -= ix.nbuckets;
    const U4* slot = ix.table + b * SLOTS_PER_BUCKET;
    const U4 s0 = slot[0], s1 = slot[1], s2 = slot[2], s3 = slot[3];
    const uint32_t klo = (uint32_t)kmer, khi = (uint32_t)(kmer >> 32);
    uint32_t h = NO_HANDLE, off = 0;
    if (s0.x == klo && s0.y == khi && s0.z != NO_HANDLE) { h = s0.z; off = s0.w; }
    if (s1.x == klo && s1.y == khi && s1.z != NO_HANDLE) { h = s1.z; off = s1.w; }
    if (s2.x == klo && s2.y == khi && s2.z != NO_HANDLE) { h = s2.z; off = s2.w; }
    if (s3.x == klo && s3.y == khi && s3.z != NO_HANDLE) { h = s3.z; off = s3.w; }
    s.flags &= (1u << F_PROBE_SHIFT) - 1;
    if (h != NO_HANDLE) {                                           // Some((nid, offset)) (:106)
        s.h = h;
        s.off = off;
        const uint32_t thr = s.L / 5;                               // (0.2 * L as f64) as usize (:77) == L/5 for L < 2^31
        if ((s.flags & F_FIRST_SEEK) && s.kp >= thr) {              // :124-126
            s.st = ST_LEFT;
            s.ra = s.kp;                                            // last_pos + 1 (:127)
            s.ph = h;                                               // :128
            s.ro = (off > 0 ? off - 1 : 0) + 1;                     // prev_kmer_offset + 1 (:129, quirk Q1 kept)
            s.flags = (s.flags & ~F_FIRST_SEEK) | F_FRESH | F_LEFT_SEED;
        } else {
            s.st = ST_FWD;
            s.flags = (s.flags & ~F_FIRST_SEEK) | F_FRESH;
        }
        return;
    }
    const bool full = s0.z != NO_HANDLE && s1.z != NO_HANDLE && s2.z != NO_HANDLE && s3.z != NO_HANDLE;
    if (full && probe + 1 < ix.nbuckets) {                          // key may live in the next bucket
        s.flags |= (probe + 1) << F_PROBE_SHIFT;
        return;
    }
    s.kp += PA_SEEK_STRIDE;                                         // :110
    if (s.kp > last) s.st = s.ncol ? ST_ISECT : ST_NONE;            // None (:113) -> :294 break / :305-314
}

// ---------------------------------------------------------------------------------------------- FWD
// Forward search (:209-301): one call = enter/continue one node and compare up to 64 bases.
template <bool TRACE = false>
PA_HD void fwd_step(Lane& s, const DevIndexView& ix, ReadRef rd, ColRef cols, uint32_t allowed) {
    const uint32_t K = ix.k;
    const bool fresh = s.flags & F_FRESH;
    const uint32_t ro0 = fresh ? s.off + K : s.ro;                  // ref_offset (:227)
    uint32_t kp = fresh ? s.kp + K : s.kp;                          // kmer_pos += kmer_length (:215)
    const Hdr hd = load_hdr(ix, s.h);                               // dbg.get_node (:210)
    const uint64_t* sq = node_seq(ix, s.h) + (ro0 >> 5);
    const uint64_t a0 = sq[0], a1 = sq[1], a2 = sq[2];
    uint32_t rem = s.rem, snp = s.snp, ro = ro0;
    if (fresh) {
        s.cov += K;                                                 // :216
        push_node<TRACE>(s, cols, hd.colour, hd.nid);                // nodes.push (:219)
        rem = pa_min(s.L - kp, hd.len - ro);                        // max_matchable_pos (:222-231)
        snp = 0;                                                    // :235
        s.flags &= ~F_FRESH;
    }
    bool premature = false;
    const uint32_t n = pa_min(rem, 64u);
    uint32_t matched = 0;
    if (n > 0) {
        const uint32_t sh = (ro & 31) * 2;
        matched = compare_chunk(diff_mask(read_window(rd, kp) ^ funnel(a0, a1, sh)), pa_min(n, 32u), allowed, snp, s.mism, premature);
        if (!premature && n > 32)
            matched += compare_chunk(diff_mask(read_window(rd, kp + 32) ^ funnel(a1, a2, sh)), n - 32, allowed, snp, s.mism, premature);
    }
    kp += matched;                                                  // :257
    s.cov += matched;                                               // :254
    ro += matched;
    rem -= matched;
    s.kp = kp;
    s.ro = ro;
    s.rem = rem;
    s.snp = snp;
    if (!premature && rem > 0) return;                              // same node, next 64 bases
    if (kp >= s.L) { s.st = ST_ISECT; return; }                     // :259-261
    const uint32_t b = read_base(rd, kp);                           // :265
    if (!premature && ((hd.exts >> b) & 1u)) {                      // :267
        s.h = b == 0 ? hd.e0 : b == 1 ? hd.e1 : b == 2 ? hd.e2 : hd.e3;   // r_edges()[index].0 (:275-278)
        s.off = 0;                                                  // :279
        s.kp = kp - (K - 1);                                        // :282
        s.cov -= K - 1;                                             // :283
        s.flags |= F_FRESH;
    } else if (kp > s.L - K) {                                      // :287-290
        s.st = ST_ISECT;
    } else {
        s.st = ST_SEEK;                                             // find_kmer_match(&mut kmer_pos) (:293)
    }
}

// ---------------------------------------------------------------------------------------------- LEFT
// Left extension (:131-203): one call = enter/continue one node and compare up to 32 bases leftwards.
template <bool TRACE = false>
PA_HD void left_step(Lane& s, const DevIndexView& ix, ReadRef rd, ColRef cols, uint32_t allowed) {
    const uint32_t K = ix.k;
    const Hdr hd = load_hdr(ix, s.ph);                              // dbg.get_node(prev_node_id) (:132)
    uint32_t na = s.ro, rem = s.rem, snp = s.snp;
    if (s.flags & F_FRESH) {
        if (!(s.flags & F_LEFT_SEED)) {
            push_node<TRACE>(s, cols, hd.colour, hd.nid);            // nodes.push(prev_node.node_id) (:199)
            na = hd.len - K + 1;                                    // prev_kmer_offset = len - k (:196)
        }
        rem = pa_min(s.ra, na);                                     // max_matchable_pos (:139-145)
        snp = 0;                                                    // :150
        s.flags &= ~(F_FRESH | F_LEFT_SEED);
    }
    bool premature = false;
    const uint32_t n = pa_min(rem, 32u);
    uint32_t matched = 0;
    if (n > 0) {
        const uint32_t po = na - 1, lp = s.ra - 1;                  // ref_pos / read_offset of idx 0 (:152-153)
        const uint64_t* sq = node_seq(ix, s.ph);
        uint64_t sw;
        if (po >= 31) {
            const uint32_t st = po - 31;
            sw = funnel(sq[st >> 5], sq[(st >> 5) + 1], (st & 31) * 2);
        } else {
            sw = sq[0] << (2 * (31 - po));
        }
        const uint64_t m = diff_mask(read_window_end(rd, lp) ^ sw);
        matched = compare_chunk(pa_brev64(m) >> 1, n, allowed, snp, s.mism, premature);   // base idx 0 = top bits
    }
    s.ra -= matched;                                                // last_pos -= matched_bases (:178)
    na -= matched;
    rem -= matched;
    s.cov += matched;                                               // :169
    s.ro = na;
    s.rem = rem;
    s.snp = snp;
    if (!premature && rem > 0) return;
    bool stop = s.ra == 0 || premature;                             // :173-175
    if (!stop) {
        const uint32_t b = read_base(rd, s.ra - 1);                 // next_base = read_seq.get(last_pos) (:182)
        if ((hd.exts >> (4 + b)) & 1u) {                            // has_ext(Dir::Left, b) (:183)
            s.ph = ix.ledge[4ull * hd.nid + b];                     // l_edges()[index].0 (:191-194)
            s.flags |= F_FRESH;
            return;
        }
        stop = true;                                                // :200-202
    }
    s.st = ST_FWD;                                                  // forward search from the seed (:208)
    s.flags |= F_FRESH;
}

// ---------------------------------------------------------------------------------------------- ISECT
// nodes_to_eq_class (:323-356) + intersect (:389-418): the class is the intersection of the colour lists of every
// visited node. Base list = a shortest one (what the stable sort at :331-334 puts first); survivors are tracked as a
// 64-bit mask over the base list when it has <= 64 ids, else recomputed in the write pass.
struct Isect {
    uint32_t base_start, base_len, base_colour, count;
    uint64_t alive;
};

PA_HD bool list_contains(const uint32_t* v, uint32_t n, uint32_t key) {   // binary_search (:404)
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const uint32_t x = v[mid];
        if (x < key) lo = mid + 1; else hi = mid;
    }
    return lo < n && v[lo] == key;
}

PA_HD bool in_all_lists(const DevIndexView& ix, ColRef cols, uint32_t ncol, uint32_t base_colour, uint32_t v) {
    for (uint32_t i = 0; i < ncol; ++i) {
        const uint32_t c = get_colour(cols, i);
        if (c == base_colour) continue;
        const uint32_t st = ix.ec_off[c], ln = ix.ec_off[c + 1] - st;
        if (!list_contains(ix.ec_ids + st, ln, v)) return false;
    }
    return true;
}

PA_HD Isect isect_count(const Lane& s, const DevIndexView& ix, ColRef cols) {
    Isect r{0, 0xFFFFFFFFu, 0, 0, 0};
    for (uint32_t i = 0; i < s.ncol; ++i) {
        const uint32_t c = get_colour(cols, i);
        const uint32_t st = ix.ec_off[c], ln = ix.ec_off[c + 1] - st;
        if (ln < r.base_len) { r.base_len = ln; r.base_start = st; r.base_colour = c; }
    }
    if (r.base_len <= 64) {
        uint64_t alive = r.base_len == 64 ? ~0ull : ((1ull << r.base_len) - 1);
        for (uint32_t i = 0; i < s.ncol && alive; ++i) {
            const uint32_t c = get_colour(cols, i);
            if (c == r.base_colour) continue;
            const uint32_t st = ix.ec_off[c], ln = ix.ec_off[c + 1] - st;
            for (uint64_t t = alive; t; t &= t - 1) {
                const uint32_t j = pa_ctz64(t);
                if (!list_contains(ix.ec_ids + st, ln, ix.ec_ids[r.base_start + j])) alive &= ~(1ull << j);
            }
        }
        r.alive = alive;
        r.count = pa_popc64(alive);
    } else {
        for (uint32_t j = 0; j < r.base_len; ++j)
            r.count += in_all_lists(ix, cols, s.ncol, r.base_colour, ix.ec_ids[r.base_start + j]) ? 1u : 0u;
    }
    return r;
}

PA_HD void isect_write(const Lane& s, const DevIndexView& ix, ColRef cols, const Isect& r, uint32_t* dst) {
    if (r.base_len <= 64) {
        for (uint64_t t = r.alive; t; t &= t - 1) *dst++ = ix.ec_ids[r.base_start + pa_ctz64(t)];
    } else {
        for (uint32_t j = 0; j < r.base_len; ++j) {
            const uint32_t v = ix.ec_ids[r.base_start + j];
            if (in_all_lists(ix, cols, s.ncol, r.base_colour, v)) *dst++ = v;
        }
    }
}

}  // namespace pa
