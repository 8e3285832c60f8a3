// The text of lane_steps.hpp in two kinds of sections: PA_LS_COMMON (what does not depend on how a lane's state is packed: included once, in
// namespace pa) and PA_LS_LANE (the packed lane state and every step that reads or writes it: included once per packing — PA_LANE_WIDE 0 / 1 —
// in namespace pa::narrow / pa::wide). Never included directly: lane_steps.hpp does.
#if PA_LS_COMMON
// walk states, then the finishing states the kernel schedules separately (ST_ISECT = walk ended, tier not yet chosen)
enum : uint32_t { ST_EMPTY = 0, ST_SEEK = 1, ST_FWD = 2, ST_LEFT = 3, ST_ISECT = 4, ST_NONE = 5,
                  ST_F_LIGHT = 6, ST_F_SCAN = 7, ST_F_COOP = 8, ST_F_BITS = 9, ST_F_MASK = 10, ST_COUNT = 11 };
enum : uint32_t { F_FRESH = 1u, F_FIRST_SEEK = 2u, F_LEFT_SEED = 4u, F_SPILL_OVERFLOW = 8u, F_CAREFUL = 16u, F_LISTS = 32u,
                  F_SMALL_BASE = 64u,   // list mode: the shortest class met has <= 8 ids
                  F_SPEC = 128u };      // the scan for a k-mer is past a miss (or re-seeks behind a broken-off node visit): the next probe will
                                        // probably miss too, so a step probes kmer_pos AND kmer_pos + 3 (evaluated in the scan's order: exact)
constexpr uint32_t NO_CLASS = 0xFFFFFFFFu;
constexpr uint32_t LDS_CLASSES = 4;   // list mode: distinct classes kept inline (one 16-byte vector each of refs, lengths, class ids)
#endif   // PA_LS_COMMON

#if PA_LS_LANE
#if !PA_LANE_WIDE
// Packed state of a read (9 words), the NARROW packing. Limits: read length <= 16383 (14 bits: positions, class and node counters), node length < 2^24.
struct Lane {
    uint32_t rid;
    uint32_t lk;    // L (bits 0..13) | kmer_pos (14..27) | state (28..31)                  (:70, :79)
    uint32_t cm;    // read_coverage (0..15) | mismatch_count (16..31)                      (:71-72)
    uint32_t h;     // forward search: the chain block it is in (node_id + kmer_offset of :118-121 as a place in a chain)
    uint32_t of;    // position in that block's window (0..9): of the k-mer's first base (F_FRESH), else of the next base to compare |
                    // slot of the block that holds the record of the node the lane stands in (10..11), valid when bit 12 is set (else 0) | flags (24..31)
    uint32_t rr;    // LEFT: position + 1 in the block's window of the next base to compare (0..23) | seen_snp (24..31)
    uint32_t rm;    // SEEK: what a probe that goes on remembers (0..3, l_pending) | LEFT: read bases still to the left (16..31)
    uint32_t ph;    // LEFT: the chain block the extension is in                            (:128)
    uint32_t nc;    // classes collected (0..13) | dictionary probe index (14..17) | TRACE: nodes.len() (18..31)
                    // (a read of L <= 16383 bases visits at most L nodes — every visit consumes a base — so 14 bits hold both counts).
                    // Classes collected: list mode = the distinct classes; window mode = bit 0 "a window is held" | pending classes << 1
};
constexpr uint32_t LANE_MAX_READ_LEN = 16383u;
PA_HD uint32_t l_L(const Lane& s) { return s.lk & 0x3FFFu; }
PA_HD uint32_t l_kp(const Lane& s) { return (s.lk >> 14) & 0x3FFFu; }
PA_HD void l_set_kp(Lane& s, uint32_t kp) { s.lk = (s.lk & 0xF0003FFFu) | (kp << 14); }
PA_HD void l_set_lk(Lane& s, uint32_t L, uint32_t kp, uint32_t st) { s.lk = L | (kp << 14) | (st << 28); }
PA_HD uint32_t l_cov(const Lane& s) { return s.cm & 0xFFFFu; }
PA_HD uint32_t l_mism(const Lane& s) { return s.cm >> 16; }
PA_HD void l_set_cm(Lane& s, uint32_t cov, uint32_t mism) { s.cm = cov | (mism << 16); }
constexpr uint32_t NC_COL_MASK = 0x3FFFu, NC_PROBE_SHIFT = 14, NC_TRACE_SHIFT = 18;
PA_HD uint32_t l_ntrace(const Lane& s) { return s.nc >> NC_TRACE_SHIFT; }
PA_HD void l_inc_trace(Lane& s) { s.nc += 1u << NC_TRACE_SHIFT; }
PA_HD uint32_t l_left_ra(const Lane& s) { return s.rm >> 16; }                                   // LEFT: read bases still to the left
PA_HD void l_set_left_ra(Lane& s, uint32_t ra) { s.rm = (s.rm & 0xFFFFu) | (ra << 16); }
PA_HD void l_zero(Lane& s) { s.lk = s.cm = s.h = s.of = s.rr = s.rm = s.ph = s.nc = 0; }
#else
// The WIDE packing (12 words): reads of up to 2^20 - 1 bases (PA_MAX_READ_LEN) that stay in HBM while they are mapped (GREAD kernels). Positions
// have 28 bits, the read's length, coverage, mismatches and the node-trace length a word of their own, the class counter 24 bits.
struct Lane {
    uint32_t rid;
    uint32_t lk;    // kmer_pos (0..27) | state (28..31)                                    (:79)
    uint32_t cm;    // read_coverage                                                        (:71)
    uint32_t h;     // as narrow
    uint32_t of;    // as narrow
    uint32_t rr;    // as narrow
    uint32_t rm;    // SEEK: l_pending (0..3) | LEFT: read bases still to the left (4..31)
    uint32_t ph;    // as narrow
    uint32_t nc;    // classes collected (0..23) | dictionary probe index (24..27)
    uint32_t xl;    // L                                                                    (:70)
    uint32_t xm;    // mismatch_count                                                       (:72)
    uint32_t xt;    // TRACE: nodes.len()
};
constexpr uint32_t LANE_MAX_READ_LEN = 0xFFFFFu;
PA_HD uint32_t l_L(const Lane& s) { return s.xl; }
PA_HD uint32_t l_kp(const Lane& s) { return s.lk & 0x0FFFFFFFu; }
PA_HD void l_set_kp(Lane& s, uint32_t kp) { s.lk = (s.lk & 0xF0000000u) | kp; }
PA_HD void l_set_lk(Lane& s, uint32_t L, uint32_t kp, uint32_t st) { s.xl = L; s.lk = kp | (st << 28); }
PA_HD uint32_t l_cov(const Lane& s) { return s.cm; }
PA_HD uint32_t l_mism(const Lane& s) { return s.xm; }
PA_HD void l_set_cm(Lane& s, uint32_t cov, uint32_t mism) { s.cm = cov; s.xm = mism; }
constexpr uint32_t NC_COL_MASK = 0xFFFFFFu, NC_PROBE_SHIFT = 24;
PA_HD uint32_t l_ntrace(const Lane& s) { return s.xt; }
PA_HD void l_inc_trace(Lane& s) { s.xt += 1u; }
PA_HD uint32_t l_left_ra(const Lane& s) { return s.rm >> 4; }
PA_HD void l_set_left_ra(Lane& s, uint32_t ra) { s.rm = (s.rm & 15u) | (ra << 4); }
PA_HD void l_zero(Lane& s) { s.lk = s.cm = s.h = s.of = s.rr = s.rm = s.ph = s.nc = s.xl = s.xm = s.xt = 0; }
#endif

PA_HD uint32_t l_st(const Lane& s) { return s.lk >> 28; }
PA_HD void l_set_st(Lane& s, uint32_t st) { s.lk = (s.lk & 0x0FFFFFFFu) | (st << 28); }
PA_HD uint32_t l_flags(const Lane& s) { return s.of >> 24; }
PA_HD void l_or_flags(Lane& s, uint32_t f) { s.of |= f << 24; }
PA_HD void l_clr_flags(Lane& s, uint32_t f) { s.of &= ~(f << 24); }
constexpr uint32_t OF_X_MASK = 0x3FFu, OF_CUR_SHIFT = 10, OF_CUR_KNOWN = 1u << 12;
PA_HD uint32_t l_off(const Lane& s) { return s.of & OF_X_MASK; }
PA_HD uint32_t l_cur(const Lane& s) { return (s.of >> OF_CUR_SHIFT) & 3u; }   // 0 when not known
PA_HD uint32_t of_cur(uint32_t slot, bool known) { return known ? (slot << OF_CUR_SHIFT) | OF_CUR_KNOWN : 0u; }
PA_HD uint32_t l_ncol(const Lane& s) { return s.nc & NC_COL_MASK; }
PA_HD uint32_t l_probe(const Lane& s) { return (s.nc >> NC_PROBE_SHIFT) & 15u; }
PA_HD uint32_t l_npend(const Lane& s) { return l_ncol(s) >> 1; }   // window mode: classes without windows met so far
constexpr uint32_t PEND_MAX = NC_COL_MASK >> 1;
#endif   // PA_LS_LANE

#if PA_LS_COMMON
struct ReadRef {   // the lane's packed read: word w < wmax at p[w * stride]; words from wmax on read as zero
    const uint64_t* p;
    uint32_t stride;
    uint32_t wmax;
    bool slack = false;   // the four words after the read's last may be loaded as well (LDS pool: whatever lies there is never looked at)
};

// The lane's record of the classes seen, in one of two modes:
//   window mode (default)  win[0..3] = {base1, mask1, base2, mask2}: the running intersection of the classes of every
//                          node pushed so far as two 32-id windows (bit i of mask w = transcript base w + i; base2 >=
//                          base1 + 32) and wcand[0] = the class id when that intersection IS one of the classes seen,
//                          else NO_CLASS. A class that does NOT fit two windows (cmask == 0: few ids, far apart — a repeat
//                          shared by distant genes) is only noted: (ref, len) appended to pend[], the same HBM row
//                          list mode uses. Such a class cannot BE the result once a window is held (the result fits
//                          the window, the class does not), it can only remove ids from it: mask_pending does that
//                          after the walk, one pass over the class's ids per pending entry (state ST_F_MASK).
//   list mode (F_LISTS)    the classes seen: the first LDS_CLASSES as refs[0..3] / lens[0..3] / cids[0..3] (each one
//                          16-byte vector), the rest as (ref, len, class id, -) quads in `spill` — all of it in HBM and
//                          only written during the walk. What the walk reads back is in LDS: win[0..2] = refs of the
//                          first three classes (exact dedupe while there are <= 3, which the register tier needs),
//                          win[3] / wcand[0] = ref / length of the shortest class so far (the base of the intersection).
// A read starts in window mode. A read that ends its walk with pending classes and NO window (every class it met lacks
// windows), or with more than PEND_MAX pending, is restarted in list mode.
struct ColRef {
    uint32_t* win;    // window mode: one 16-byte vector
    uint32_t* wcand;  // window mode: one word
    uint32_t* refs;
    uint32_t* lens;
    uint32_t* cids;
    uint32_t* spill;
    uint32_t spill_cap;   // u32 words
    uint32_t* pend;       // window mode: (ref, len) pairs of the classes without windows; capacity >= spill_cap words
    uint32_t* trace;      // TRACE builds only: node ids in visit order (map_read_to_nodes, :54-61), capacity spill_cap
};

struct Seg {   // one node record of a chain block (device_layout.hpp), decoded
    uint32_t e;        // the node's end relative to the block (SEG_E_FAR: beyond the window)
    uint32_t flags;    // SEG_WIDE / SEG_LAST / SEG_EDGES (word 0 of the record as it is)
    uint32_t cid, cmin, cmask, cmin2, cmask2;
    uint32_t ec_ref, ec_len;   // ec_ref only when SEG_WIDE (a one-window class has its record looked up when list mode asks for it)
};

// ---------------------------------------------------------------------------------------------- helpers
PA_HD uint64_t pa_mix64(uint64_t x) {   // murmur3 fmix64
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

PA_HD uint32_t pa_popc32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__popc(x);
#else
    return (uint32_t)__builtin_popcount(x);
#endif
}
PA_HD uint32_t pa_popc64(uint64_t x) { return pa_popc32((uint32_t)x) + pa_popc32((uint32_t)(x >> 32)); }

PA_HD uint32_t pa_ctz64(uint64_t x) {   // x != 0
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)(__ffsll((unsigned long long)x) - 1);
#else
    return (uint32_t)__builtin_ctzll(x);
#endif
}
PA_HD uint32_t pa_ctz32(uint32_t x) {   // x != 0
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)(__ffs((int)x) - 1);
#else
    return (uint32_t)__builtin_ctz(x);
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

// 64 bits starting at bit `sh` (0..63) of the 128-bit value hi:lo
PA_HD uint64_t funnel(uint64_t lo, uint64_t hi, uint32_t sh) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t w0 = (uint32_t)lo, w1 = (uint32_t)(lo >> 32), w2 = (uint32_t)hi, w3 = (uint32_t)(hi >> 32);
    const bool up = sh >= 32;                 // v_alignbit_b32 uses sh[4:0]
    const uint32_t t0 = up ? w1 : w0, t1 = up ? w2 : w1, t2 = up ? w3 : w2;
    return (uint64_t)__builtin_amdgcn_alignbit(t1, t0, sh) | ((uint64_t)__builtin_amdgcn_alignbit(t2, t1, sh) << 32);
#else
    return sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
#endif
}

PA_HD uint64_t read_word(ReadRef r, uint32_t w) {
    const uint64_t v = r.p[pa_min(w, r.wmax - 1) * r.stride];
    return w < r.wmax ? v : 0;
}
// 32 bases of the read starting at base `pos`
PA_HD uint64_t read_window(ReadRef r, uint32_t pos) {
    const uint32_t w = pos >> 5;
    return funnel(read_word(r, w), read_word(r, w + 1), (pos & 31) * 2);
}
// the same when the caller only looks at bases INSIDE the read (a k-mer at kp <= L - K): the word after the last one is
// not zeroed but re-read, one multiply instead of two
PA_HD uint64_t read_window_in(ReadRef r, uint32_t pos) {
    const uint32_t i0 = (pos >> 5) * r.stride, ilast = (r.wmax - 1) * r.stride;
    return funnel(r.p[i0], r.p[pa_min(i0 + r.stride, ilast)], (pos & 31) * 2);
}
// 32 bases ENDING at base p (base p lands in the top 2 bits; missing low bases are zero)
PA_HD uint64_t read_window_end(ReadRef r, uint32_t p) { return p >= 31 ? read_window(r, p - 31) : r.p[0] << (2 * (31 - p)); }

PA_HD uint32_t read_base(ReadRef r, uint32_t pos) { return (uint32_t)(r.p[(pos >> 5) * r.stride] >> ((pos & 31) * 2)) & 3u; }

// Streaming loads: every dictionary line, node blob and read word is touched by ONE read of the batch and never again, while
// the count replicas, the window table and the hot class records are shared by all of them. PA_NT (bit 0 read words,
// 1 result stores, 2 dictionary lines, 3 node blobs) marks the former non-temporal, so that they do not push the latter out
// of the L2.
#ifndef PA_NT
#define PA_NT 0
#endif
#if defined(__HIP_DEVICE_COMPILE__)
typedef uint32_t pa_nt_u32x4 __attribute__((ext_vector_type(4)));
PA_HD U4 ld_nt(const U4* p) {
    const pa_nt_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const pa_nt_u32x4*>(p));
    return U4{v.x, v.y, v.z, v.w};
}
PA_HD uint64_t ld_nt(const uint64_t* p) { return __builtin_nontemporal_load(p); }
typedef uint64_t pa_nt_u64x2 __attribute__((ext_vector_type(2), aligned(8)));
PA_HD Q2 ld_nt(const Q2* p) {
    const pa_nt_u64x2 v = __builtin_nontemporal_load(reinterpret_cast<const pa_nt_u64x2*>(p));
    return Q2{v.x, v.y};
}
#else
PA_HD Q2 ld_nt(const Q2* p) { return *p; }
PA_HD U4 ld_nt(const U4* p) { return *p; }
PA_HD uint64_t ld_nt(const uint64_t* p) { return *p; }
#endif
#define PA_LD(bit, ptr) ((PA_NT & (bit)) ? ld_nt(ptr) : *(ptr))
// ... and decided per INDEX at run time (DevIndexView::stream_nt, wave-uniform): dictionary lines and read words of an index whose
// dictionary is larger than the caches are loaded non-temporal, so that they do not push the chain blocks — 0.4 GB that every read
// comes back to — out of the L2 and the Infinity Cache (round 5: config 3 -3 %, config 5 -4...5 % kernel time; an index that fits the
// caches, config 2, is 2 % faster WITHOUT the hint: profiles/r05_nontemporal_ab.txt)
template <class T>
PA_HD T ld_stream(const T* p, bool nt) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (__builtin_amdgcn_readfirstlane((int)nt)) return ld_nt(p);
#else
    (void)nt;
#endif
    return *p;
}

// first byte of chain block `h`
PA_HD const uint8_t* chain_block(const DevIndexView& ix, uint32_t h) { return ix.blobs + (uint64_t)h * CH_BLOCK; }
// the four slots of a block as separate values (never an array: a per-lane index into one would be served from scratch memory)
struct Slots {
    U4 a, b, c, d;
};
// a / b / c / d by i = 0..3 as a tree of two-way selects on the bits of i (a chain `i == 0 ? a : i == 1 ? ...` is turned into a
// switch by the compiler, and that into nested divergent branches)
PA_HD uint32_t sel4(uint32_t i, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    const bool b0 = i & 1u, b1 = i & 2u;
    const uint32_t lo = b0 ? b : a, hi = b0 ? d : c;
    return b1 ? hi : lo;
}
PA_HD uint64_t sel4q(uint32_t i, uint64_t a, uint64_t b, uint64_t c, uint64_t d) {
    const bool b0 = i & 1u, b1 = i & 2u;
    const uint64_t lo = b0 ? b : a, hi = b0 ? d : c;
    return b1 ? hi : lo;
}
PA_HD U4 slot_at(const Slots& sl, uint32_t i) {
    return U4{sel4(i, sl.a.x, sl.b.x, sl.c.x, sl.d.x), sel4(i, sl.a.y, sl.b.y, sl.c.y, sl.d.y), sel4(i, sl.a.z, sl.b.z, sl.c.z, sl.d.z),
              sel4(i, sl.a.w, sl.b.w, sl.c.w, sl.d.w)};
}
// record in slot i (and its extension in slot i + 1)
PA_HD Seg seg_at(const Slots& sl, uint32_t i) {
    const U4 r = slot_at(sl, i), x = slot_at(sl, (i + 1) & 3u);
    const bool wide = (r.x & SEG_WIDE) != 0;
    Seg g;
    g.e = r.x & SEG_E_MASK;
    g.flags = r.x;
    g.cid = r.y; g.cmin = r.z; g.cmask = r.w;
    g.cmin2 = wide ? x.x : 0u;
    g.cmask2 = wide ? x.y : 0u;
    g.ec_ref = wide ? x.z : NO_HANDLE;
    g.ec_len = wide ? x.w : 0u;
    return g;
}
// the first record whose node ends beyond window position y (the node that owns the k-mer ending at y, or the base y + 1)
PA_HD uint32_t seg_find(const Slots& sl, uint32_t y) {
    const uint32_t gt = (uint32_t)((sl.a.x & SEG_E_MASK) > y) | ((uint32_t)((sl.b.x & SEG_E_MASK) > y) << 1) |
                        ((uint32_t)((sl.c.x & SEG_E_MASK) > y) << 2) | ((uint32_t)((sl.d.x & SEG_E_MASK) > y) << 3);
    const uint32_t m = gt & (sl.a.x >> SEG_RECMASK_SHIFT);
    return pa_ctz32(m | 8u);
}
// node id of the node whose k-mers start at global position g = 64 * block handle + position in its window (node traces only)
PA_HD uint32_t trace_nid(const DevIndexView& ix, uint64_t g) {
    uint32_t lo = 0, hi = ix.num_segs - 1;
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo + 1) / 2;
        if (ix.seg_g[mid] <= g) lo = mid; else hi = mid - 1;
    }
    return ix.seg_nid[lo];
}

// mismatch mask of a 32-base XOR restricted to its first n bases (1 <= n <= 32): bit 2i set <=> base i differs.
// (the 32-bit halves can be shifted separately: the bit that would cross lands on an odd position and is masked away)
PA_HD uint64_t diff_mask(uint64_t x, uint32_t n) {
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    lo = (lo | (lo >> 1)) & 0x55555555u;
    hi = (hi | (hi >> 1)) & 0x55555555u;
    const uint64_t keep = n >= 32 ? ~0ull : ((1ull << (2 * n)) - 1);
    return ((uint64_t)lo | ((uint64_t)hi << 32)) & keep;
}

// the same for 1 <= n <= 32 (the callers that skip empty chunks): the keep mask is one 64-bit shift
PA_HD uint64_t diff_mask_nz(uint64_t x, uint32_t n) {
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    lo = (lo | (lo >> 1)) & 0x55555555u;
    hi = (hi | (hi >> 1)) & 0x55555555u;
    return ((uint64_t)lo | ((uint64_t)hi << 32)) & (~0ull >> (64u - 2u * n));
}

// The body of the compare loops (:151-170 / :236-255) over n <= 32 bases given their mismatch mask (bit 2i = i-th base
// compared). Returns matched_bases for the chunk; updates seen_snp / mismatch_count; sets premature. (slow path)
PA_HD uint32_t compare_chunk(uint64_t m, uint32_t n, uint32_t allowed, uint32_t& snp, uint32_t& mism, bool& premature) {
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

// hash of a canonical two-window class (cmin is the smallest id, so bit 0 of m1 is set; b2/m2 = 0 when one window)
PA_HD uint32_t window_hash(uint32_t b1, uint32_t m1, uint32_t b2, uint32_t m2) {
    return (uint32_t)(pa_mix64(((uint64_t)b1 << 32 | m1) ^ pa_mix64((uint64_t)b2 << 32 | m2)) >> 32);
}

// canonical form of the id set {b1 + i : m1 bit i} U {b2 + i : m2 bit i} (b2 >= b1 + 32, not both empty): window 1 starts
// at the smallest id, window 2 at the first id beyond window 1 — what the index stores for a class
PA_HD void window_canon(uint32_t& b1, uint32_t& m1, uint32_t& b2, uint32_t& m2) {
    if (m1 == 0) { b1 = b2; m1 = m2; m2 = 0; }
    const uint32_t z = pa_ctz32(m1);
    b1 += z;
    m1 >>= z;
    if (m2) {
        const uint32_t gap = b2 - b1;                 // ids of window 2 that now fall inside window 1
        if (gap < CLASS_WINDOW) {
            m1 |= m2 << gap;
            m2 = gap ? m2 >> (CLASS_WINDOW - gap) : 0u;
            b2 = b1 + CLASS_WINDOW;
        }
    }
    if (m2) {
        const uint32_t z2 = pa_ctz32(m2);
        b2 += z2;
        m2 >>= z2;
    } else b2 = 0;
}

// class whose windows are exactly (b1, m1, b2, m2) (canonical), or NO_CLASS
PA_HD uint32_t window_class(const DevIndexView& ix, uint32_t b1, uint32_t m1, uint32_t b2, uint32_t m2) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t line = __umulhi(window_hash(b1, m1, b2, m2), ix.wbuckets);
#else
    uint32_t line = (uint32_t)(((uint64_t)window_hash(b1, m1, b2, m2) * ix.wbuckets) >> 32);
#endif
    for (;;) {
        const U4* p = reinterpret_cast<const U4*>(ix.wtable + (uint64_t)line * 16);
        const U4 a = p[0], b = p[1], c = p[2], d = p[3];
        if (a.x == b1 && a.y == m1 && a.z == b2 && a.w == m2 && b.x != NO_CLASS) return b.x;
        if (b.y == b1 && b.z == m1 && b.w == b2 && c.x == m2 && c.y != NO_CLASS) return c.y;
        if (c.z == b1 && c.w == m1 && d.x == b2 && d.y == m2 && d.z != NO_CLASS) return d.z;
        if (b.x == NO_CLASS || c.y == NO_CLASS || d.z == NO_CLASS) return NO_CLASS;   // a line with a free entry ends the probe sequence
        if (++line == ix.wbuckets) line = 0;
    }
}

// the 32 bits of a membership bitmap that start at bit b (transcripts b .. b + 31; the bitmap has two spare words)
PA_HD uint32_t window_bits(const uint32_t* bits, uint32_t b) {
    const uint32_t w = b >> 5, sh = b & 31u;
    const uint64_t v = (uint64_t)bits[w] | ((uint64_t)bits[w + 1] << 32);
    return (uint32_t)(v >> sh);
}

// mask of the window (base c, mask n) expressed relative to base b
PA_HD uint32_t window_at(uint32_t b, uint32_t c, uint32_t n) {
    // one 64-bit shift instead of two branches: n sits in the upper half of a 64-bit value, d = 32 + b - c moves it down;
    // d in [1, 63] covers both directions (c - b < 32: up; b - c < 32: down), anything else leaves nothing in the low half
    const uint32_t d = CLASS_WINDOW + b - c;
    const uint32_t r = (uint32_t)(((uint64_t)n << 32) >> (d & 63u));
    return d - 1u < 63u ? r : 0u;
}
#endif   // PA_LS_COMMON

#if PA_LS_LANE
// nodes.push(node_id) (:199, :219): fold the node's class into the lane's record. `gpos` = global position of one of the node's
// k-mers (trace_nid; TRACE builds only). Returns true when the read has to restart in list mode (the caller resets the lane
// with restart_lists).
template <bool TRACE>
PA_HD bool push_node(Lane& s, ColRef c, const DevIndexView& ix, const Seg& hd, uint64_t gpos) {
    if (TRACE) {
        const uint32_t nt = l_ntrace(s);
        if (nt < c.spill_cap) c.trace[nt] = trace_nid(ix, gpos);
        l_inc_trace(s);
    }
    const uint32_t n = l_ncol(s);
    if (!(l_flags(s) & F_LISTS)) {                                   // window mode: AND of masks
        if (hd.cmask == 0) {                                         // no windows: noted, applied after the walk
            const uint32_t np = n >> 1;
            if (np >= PEND_MAX || 2 * np + 1 >= c.spill_cap) return true;
            c.pend[2 * np] = hd.ec_ref;
            c.pend[2 * np + 1] = hd.ec_len;
            s.nc += 2;
            return false;
        }
        U4 w = *reinterpret_cast<const U4*>(c.win);                  // {base1, mask1, base2, mask2}
        uint32_t cand = hd.cid;
        if (!(n & 1u)) {
            w = U4{hd.cmin, hd.cmask, hd.cmin2, hd.cmask2};
        } else {
            // the class's windows re-based onto each running window (ids outside a running window cannot survive)
            const uint32_t m1 = w.y & (window_at(w.x, hd.cmin, hd.cmask) | window_at(w.x, hd.cmin2, hd.cmask2));
            const uint32_t m2 = w.w & (window_at(w.z, hd.cmin, hd.cmask) | window_at(w.z, hd.cmin2, hd.cmask2));
            if (pa_popc32(m1) + pa_popc32(m2) != pa_popc32(hd.cmask) + pa_popc32(hd.cmask2))   // else: this class is a subset of all before, it IS the result
                cand = (m1 == w.y && m2 == w.w) ? c.wcand[0] : NO_CLASS;   // unchanged, or a strict subset of everything seen so far
            w.y = m1;
            w.w = m2;
        }
        *reinterpret_cast<U4*>(c.win) = w;
        c.wcand[0] = cand;
        s.nc |= 1u;
        return false;
    }
    // list mode: win = {ref of class 0, 1, 2, ref of the shortest class so far}, wcand = the shortest length (both in LDS:
    // the row in HBM is only written during the walk, never read). A one-window class does not carry its record ref in the
    // block: looked up here (a dependent load that only list-mode reads pay)
    const bool wide = (hd.flags & SEG_WIDE) != 0;
    const uint32_t ec_ref = wide ? hd.ec_ref : ix.class_ref[hd.cid], ec_len = wide ? hd.ec_len : pa_popc32(hd.cmask);
    U4 r = *reinterpret_cast<const U4*>(c.win);
    const bool dup = (n > 0 && r.x == ec_ref) | (n > 1 && r.y == ec_ref) | (n > 2 && r.z == ec_ref) | (n > 0 && r.w == ec_ref);
    if (dup) return false;
    if (n == 0) r.x = ec_ref;
    if (n == 1) r.y = ec_ref;
    if (n == 2) r.z = ec_ref;
    if (n == 0 || ec_len < c.wcand[0]) {   // strict: the first of the shortest classes is the base
        r.w = ec_ref;
        c.wcand[0] = ec_len;
        if (ec_len <= 8) l_or_flags(s, F_SMALL_BASE);
    }
    *reinterpret_cast<U4*>(c.win) = r;
    if (n < LDS_CLASSES) {
        c.refs[n] = ec_ref;
        c.lens[n] = ec_len;
        c.cids[n] = hd.cid;
    } else {
        const uint32_t o = 4 * (n - LDS_CLASSES);
        if (o + 3 >= c.spill_cap || n >= NC_COL_MASK) { l_or_flags(s, F_SPILL_OVERFLOW); return false; }   // (unreachable: the row holds 2 L + 3 classes, the counter L)
        c.spill[o] = ec_ref;
        c.spill[o + 1] = ec_len;
        c.spill[o + 2] = hd.cid;
    }
    s.nc += 1;
    return false;
}

PA_HD void lane_start(Lane& s, uint32_t rid, uint32_t L, uint32_t k) {
    s.rid = rid;
    l_zero(s);
    l_set_lk(s, L, 0, L < k ? ST_NONE : ST_SEEK);                        // :82-84
    s.of = F_FIRST_SEEK << 24;
}
// only classes without windows were met (or too many of them): map the read again from its first base, this time collecting class lists
PA_HD void restart_lists(Lane& s, uint32_t k) {
    lane_start(s, s.rid, l_L(s), k);
    l_or_flags(s, F_LISTS);
}
// window mode, the walk has ended (ST_ISECT): what is left to do before the result can be written?
//   0 nothing (no pending classes)   1 mask_pending (ST_F_MASK)   2 restart in list mode (pending classes, no window to mask)
PA_HD uint32_t window_todo(const Lane& s) {
    const uint32_t n = l_ncol(s);
    return (n >> 1) == 0 ? 0u : (n & 1u) ? 1u : 2u;
}
#endif   // PA_LS_LANE
#if PA_LS_COMMON
// ids of the sorted list (record `ref`, `len` ids) that fall into the windows [b1, b1 + 32) and [b2, b2 + 32), as masks.
// (Host form: the kernel does the same with 16-byte chunks per lane and a binary search into long lists — map_pool.hip, ST_F_MASK.)
PA_HD void list_window_mask(const DevIndexView& ix, uint32_t ref, uint32_t len, uint32_t b1, uint32_t b2, uint32_t& m1, uint32_t& m2) {
    const uint32_t* ids = ix.ec + 4ull * ref + 1;
    m1 = m2 = 0;
    for (uint32_t j = 0; j < len; ++j) {
        const uint32_t d1 = ids[j] - b1, d2 = ids[j] - b2;
        if (d1 < CLASS_WINDOW) m1 |= 1u << d1;
        if (d2 < CLASS_WINDOW) m2 |= 1u << d2;
    }
#if !defined(__HIP_DEVICE_COMPILE__)
    // the class's membership bitmap (what the kernel reads instead of the list: class_bitmap) must say the same — the emulator's check of
    // the flattener's bitmaps, on every pending class of every read it maps
    if (ix.bitmap_min && len >= ix.bitmap_min) {
        const uint32_t* bits = ix.ec + class_bitmap(ref, len);
        const uint32_t x1 = window_bits(bits, b1), x2 = window_bits(bits, b2);
        if (x1 != m1 || x2 != m2) { m1 = m2 = 0xDEADBEEFu; }   // (poisons the result: the parity test fails loudly)
    }
#endif
}
#endif   // PA_LS_COMMON
#if PA_LS_LANE
// the pending classes of a window-mode read applied to its window; leaves the read as a plain window-mode result
PA_HD void mask_pending(Lane& s, const DevIndexView& ix, ColRef c) {
    U4 w = *reinterpret_cast<const U4*>(c.win);
    const uint32_t np = l_npend(s);
    uint32_t a1 = w.y, a2 = w.w;
    for (uint32_t i = 0; i < np; ++i) {
        uint32_t m1, m2;
        list_window_mask(ix, c.pend[2 * i], c.pend[2 * i + 1], w.x, w.z, m1, m2);
        a1 &= m1;
        a2 &= m2;
    }
    if (a1 != w.y || a2 != w.w) c.wcand[0] = NO_CLASS;   // a strict subset of the window classes seen, and no class without windows fits a window
    w.y = a1;
    w.w = a2;
    *reinterpret_cast<U4*>(c.win) = w;
    s.nc = (s.nc & ~NC_COL_MASK) | 1u;
}
#endif   // PA_LS_LANE

#if PA_LS_COMMON
// ---------------------------------------------------------------------------------------------- SEEK
// One dictionary probe of find_kmer_match (:91-114), K <= 32 (layout: device_layout.hpp), in the pieces the kernel interleaves
// with other work:
//   seek_issue     k-mer -> bucket and slot; ONE 16-byte load goes out: the key's home slot, or — when an earlier step of this probe found
//                  another key there — the first of the slots the home slot named
//   seek_complete  the answer, or the slots still to look at (the lane then stays in SEEK: one slot per step, so that nothing of a
//                  probe depends on a second load and its state is gone before the forward half of the iteration computes)
struct SeekProbe {
    U4 v;                    // the slot loaded
    uint32_t klo, khi;       // the k-mer
    uint32_t pending;        // l_pending of the lane at issue (0: v is the home slot)
};
struct SeekProbe2 {          // K > 32: the line of the two-word dictionary a probe looks at (two whole entries) and the k-mer's four words
    U4 k0, v0, k1, v1;
    uint32_t w0, w1, w2, w3;
};
// what a probe that goes on remembers (Lane::rm bits 0..3): the named slots still to look at as a mask over i = 0..2 (slot (home + 1 + i) & 3)
// and the home slot's overflow flag
constexpr uint32_t SK_NAMED = 7u, SK_FULL = 8u, SK_MASK = 15u;
// bucket and home slot of a k-mer (one hash)
PA_HD uint32_t pa_bucket_home(uint64_t kmer, uint32_t nbuckets, uint32_t& home) {
    // three 32-bit multiplies (fmix64 is two 64-bit ones = eight multiply instructions and their adds per probe): the dictionary's placement
    // is as good with either on the config-3 keys (measured at load 0.5: home slot 79 % / other slot of the bucket 17 % / next bucket 4 %,
    // furthest key 14 buckets from home; the table is now built at 0.25), and the mapping kernel is bound by instruction issue (config 5 -4 % time, config 3 +-0)
    uint32_t x = (uint32_t)kmer * 0x9E3779B1u + (uint32_t)(kmer >> 32) * 0x85EBCA77u;
    x ^= x >> 15;
    x *= 0x2C1B3C6Du;
    x ^= x >> 13;
    home = x & 3u;
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(x, nbuckets);
#else
    return (uint32_t)(((uint64_t)x * nbuckets) >> 32);
#endif
}
#endif   // PA_LS_COMMON
#if PA_LS_LANE
// does this step also probe kmer_pos + 3? (a lane past a miss, not in the middle of an overflow chain, with a k-mer left there)
PA_HD bool seek_two(const Lane& s, uint32_t K) { return (l_flags(s) & F_SPEC) && l_probe(s) == 0 && l_kp(s) + PA_SEEK_STRIDE <= l_L(s) - K; }
PA_HD uint32_t l_pending(const Lane& s) { return s.rm & SK_MASK; }
// the probe of the k-mer at kmer_pos + ahead (ahead = 0: at the lane's probe index; ahead = 3: the speculative second probe, always a
// home slot). `live` = false (second probe of a lane that does not speculate): the load still goes out — a branch around it would put a
// full wait behind the loads already in flight — but to the first line of the table, which all such lanes share; the probe is ignored
PA_HD void seek_issue(const Lane& s, const DevIndexView& ix, ReadRef rd, SeekProbe& q, uint32_t ahead = 0, bool live = true) {
    const uint32_t at = live ? l_kp(s) + ahead : l_kp(s);
    const uint64_t kmer = read_window_in(rd, at) & ix.kmask;   // read_seq.get_kmer(kmer_pos) (:93); kmer_pos <= L - K
    uint32_t home;
    uint32_t b = pa_bucket_home(kmer, (uint32_t)ix.nbuckets, home) + (ahead ? 0u : l_probe(s));
    if (b >= (uint32_t)ix.nbuckets) b -= (uint32_t)ix.nbuckets;
    q.pending = ahead ? 0u : l_pending(s);
    const uint32_t named = q.pending & SK_NAMED;
    uint32_t slot = named ? (home + 1u + pa_ctz32(named)) & 3u : home;
    if (!live) { b = 0; slot = 0; }
    q.v = ld_stream(reinterpret_cast<const U4*>(ix.table + (uint64_t)b * BUCKET_WORDS + SLOT_WORDS * slot), (PA_NT & 4) || ix.stream_nt);
    q.klo = (uint32_t)kmer;
    q.khi = (uint32_t)(kmer >> 32);
}
#endif   // PA_LS_LANE
#if PA_LS_COMMON
PA_HD bool slot_holds(const U4& v, uint32_t klo, uint32_t khi) { return v.z != NO_HANDLE && v.x == klo && v.y == khi; }
PA_HD uint32_t slot_flags(const U4& v) { return (~v.w >> SLOT_FLAG_SHIFT) & 15u; }
#endif   // PA_LS_COMMON
#if PA_LS_LANE
PA_HD void seek_finish(Lane& s, uint32_t K, uint32_t h, uint32_t off, bool full, uint32_t probe);
#endif   // PA_LS_LANE
#if PA_LS_COMMON
// what the slot of a probe says: a hit (h != NO_HANDLE), or `pending` != 0 — other slots of the bucket may hold the key: the probe goes
// on with them in the next step — or a miss, with `full` = a key of this home went on to the next bucket
PA_HD void seek_eval(const SeekProbe& q, uint32_t& h, uint32_t& off, bool& full, uint32_t& pending) {
    h = NO_HANDLE;
    off = 0;
    const bool hit = slot_holds(q.v, q.klo, q.khi);
    if (hit) { h = q.v.z; off = q.v.w & SLOT_OFF_MASK; }
    const uint32_t named = q.pending & SK_NAMED;
    // the home slot: its flags name the other slots and the overflow; a named slot: the rest of what the home slot named
    const uint32_t next = named ? (named & (named - 1u)) | (q.pending & SK_FULL) : slot_flags(q.v);
    pending = hit ? 0u : (next & SK_NAMED) ? next : 0u;
    full = (next & SK_FULL) != 0;
}
#endif   // PA_LS_COMMON
#if PA_LS_LANE
// The probe q of the k-mer at the lane's kmer_pos settles the lane's next state — a hit, the named slots or the next bucket still to
// look at, the end of the scan — and returns true; or it is a definite miss while another probe of this step covers kmer_pos + 3
// (`more`): then kmer_pos moves on (:110) and the caller evaluates that probe — the scan of :92-111 in its own order
PA_HD bool seek_settle(Lane& s, uint32_t K, const SeekProbe& q, bool more) {
    uint32_t h, off, pending;
    bool full;
    seek_eval(q, h, off, full, pending);
    if (pending) { s.rm = (s.rm & ~SK_MASK) | pending; return true; }
    const uint32_t probe = l_probe(s);
    if (!more || h != NO_HANDLE || (full && probe < DICT_MAX_PROBES)) {
        seek_finish(s, K, h, off, full, probe);
        return true;
    }
    l_set_kp(s, l_kp(s) + PA_SEEK_STRIDE);                          // :110 (kmer_pos + 3 <= L - K: seek_two)
    s.rm &= ~SK_MASK;
    s.nc &= ~(15u << NC_PROBE_SHIFT);
    return false;
}
PA_HD void seek_complete(Lane& s, uint32_t K, const SeekProbe& q) { (void)seek_settle(s, K, q, false); }
// with the speculative second probe (q1: the home slot of the k-mer at kmer_pos + 3, issued when seek_two(s))
PA_HD void seek_complete2(Lane& s, uint32_t K, const SeekProbe& q0, bool two, const SeekProbe& q1) {
    if (seek_settle(s, K, q0, two)) return;
    (void)seek_settle(s, K, q1, false);
}
// what a probe found -> the lane's next state (the tail of find_kmer_match and of :118-129). `h` = the chain block the k-mer
// starts in, `off` = the entry's second word (device_layout.hpp: position in the block, first-k-mer-of-its-node flag, blocks
// before this one in the chain)
PA_HD void seek_finish(Lane& s, uint32_t K, uint32_t h, uint32_t off, bool full, uint32_t probe) {
    const uint32_t L = l_L(s), kp = l_kp(s);
    s.nc &= ~(15u << NC_PROBE_SHIFT);                               // probe index back to 0
    s.rm &= ~SK_MASK;                                               // ... and nothing of a probe pending (l_pending)
    if (h != NO_HANDLE) {                                           // Some((nid, offset)) (:106)
        s.h = h;
        const uint32_t fl = l_flags(s), p = (off & ENT_P_MASK) | of_cur((off >> ENT_CUR_SHIFT) & 3u, true);   // position + the slot of the k-mer's node
        const uint32_t thr = L / 5;                                 // (0.2 * L as f64) as usize (:77) == L/5 for L < 2^31
        if ((fl & F_FIRST_SEEK) && kp >= thr) {                     // :124-126  (F_SPEC ends with the hit: fl is rebuilt below without it)
            const uint32_t back = (off >> ENT_BACK_SHIFT) & CH_BACK_MAX;
            l_set_left_ra(s, kp);                                   // last_pos + 1 (:127)
            s.ph = h - back;                                        // :128 — the block with the most room to the left
            // prev_kmer_offset (:129): one base to the left of the k-mer — or, quirk Q1 kept, the k-mer's own first base when
            // it is the first k-mer of its node (kmer_offset == 0). Stored + 1; snp = 0
            s.rr = (off & ENT_P_MASK) + CH_STRIDE * back + ((off & ENT_NODE_START) ? 1u : 0u);
            s.of = p | (((fl & ~(F_FIRST_SEEK | F_SPEC)) | F_FRESH | F_LEFT_SEED) << 24);
            l_set_st(s, ST_LEFT);
        } else {
            s.of = p | (((fl & ~(F_FIRST_SEEK | F_SPEC)) | F_FRESH) << 24);
            l_set_st(s, ST_FWD);
        }
        return;
    }
    if (full && probe < DICT_MAX_PROBES) {                          // a key of this home slot (K > 32: of this line) went on to the next bucket
        s.nc |= (probe + 1) << NC_PROBE_SHIFT;
        return;
    }
    const uint32_t nkp = kp + PA_SEEK_STRIDE;                       // :110
    l_set_kp(s, nkp);
    l_or_flags(s, F_SPEC);
    if (nkp > L - K) l_set_st(s, l_ncol(s) ? ST_ISECT : ST_NONE);   // None (:113) -> :294 break / :305-314
}
#endif   // PA_LS_LANE

#if PA_LS_COMMON
// hash of a k-mer of more than 32 bases (two words)
PA_HD uint64_t pa_mix128(uint64_t lo, uint64_t hi) { return pa_mix64(lo ^ (pa_mix64(hi) * 0x9e3779b97f4a7c15ull)); }
#endif   // PA_LS_COMMON

#if PA_LS_LANE
// One dictionary probe of find_kmer_match (:91-114): dbg_index.get + verification collapse into one bucket line.
// K > 32 (two-word k-mers: a line holds two whole entries {key word 0..3, handle, off, -, -}) in the same two pieces as seek_issue / seek_complete, so
// that the pooled kernel can let such a probe ride in a forward iteration too (round 6; it used to be a step of its own)
PA_HD void seek_issue_k2(const Lane& s, const DevIndexView& ix, ReadRef rd, SeekProbe2& q) {
    const uint32_t kp = l_kp(s), probe = l_probe(s);
    const uint64_t klo = read_window(rd, kp), khi = read_window(rd, kp + 32) & ix.kmask_hi;   // read_seq.get_kmer(kmer_pos) (:93)
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t b = __umulhi((uint32_t)(pa_mix128(klo, khi) >> 32), (uint32_t)ix.nbuckets) + probe;
#else
    uint32_t b = (uint32_t)(((pa_mix128(klo, khi) >> 32) * (uint32_t)ix.nbuckets) >> 32) + probe;
#endif
    if (b >= (uint32_t)ix.nbuckets) b -= (uint32_t)ix.nbuckets;
    const U4* line = reinterpret_cast<const U4*>(ix.table + (uint64_t)b * BUCKET_WORDS);
    q.k0 = line[0]; q.v0 = line[1]; q.k1 = line[2]; q.v1 = line[3];   // (never non-temporal: four loads of one line, config 3 at K = 64 +9 % time with the hint)
    q.w0 = (uint32_t)klo; q.w1 = (uint32_t)(klo >> 32); q.w2 = (uint32_t)khi; q.w3 = (uint32_t)(khi >> 32);
}
PA_HD void seek_complete_k2(Lane& s, uint32_t K, const SeekProbe2& q) {
    const bool h0 = q.k0.x == q.w0 && q.k0.y == q.w1 && q.k0.z == q.w2 && q.k0.w == q.w3 && q.v0.x != NO_HANDLE,
               h1 = q.k1.x == q.w0 && q.k1.y == q.w1 && q.k1.z == q.w2 && q.k1.w == q.w3 && q.v1.x != NO_HANDLE;
    seek_finish(s, K, h0 ? q.v0.x : h1 ? q.v1.x : NO_HANDLE, h0 ? q.v0.y : q.v1.y, q.v0.x != NO_HANDLE && q.v1.x != NO_HANDLE, l_probe(s));
}
PA_HD void seek_step(Lane& s, const DevIndexView& ix, ReadRef rd) {
    const uint32_t K = ix.k;
    if (K > 32) {
        SeekProbe2 q2;
        seek_issue_k2(s, ix, rd, q2);
        seek_complete_k2(s, K, q2);
        return;
    }
    SeekProbe q, q1;
    const bool two = seek_two(s, K);
    seek_issue(s, ix, rd, q);
    seek_issue(s, ix, rd, q1, PA_SEEK_STRIDE, two);
    seek_complete2(s, K, q, two, q1);
}
#endif   // PA_LS_LANE

#if PA_LS_COMMON
// ---------------------------------------------------------------------------------------------- FWD
// Forward search (:209-301): one call = one chain block = one dependent fetch (the block's four slots and the sequence words
// this step can need, issued together). A step compares up to 128 bases and walks through as many NODES of the chain as lie
// in them: inside a node it counts mismatches (the compare loop :236-255 without a per-base loop); at a node's end e the
// chain's base at e is the one right extension the node has, so `has_ext(Right, read[kmer_pos])` (:267) is "that base equals
// the read's" and the hop (:275-283: kmer_pos and coverage move on by ONE base net, nodes.push, seen_snp = 0) needs no fetch.
// At the chain's end the edge slot names the next chain. When a node would exceed its mismatch budget nothing of it is
// consumed and the lane switches to careful mode, which walks it 32 bases per call and locates the breaking base exactly as
// the reference's loop does.
struct FwdLoad {     // what fwd_issue leaves in flight
    U4 s0, s1, s2, s3;
    Q2 s01, s23, s45;
};
#endif   // PA_LS_COMMON
#if PA_LS_LANE
PA_HD void fwd_issue(const Lane& s, const DevIndexView& ix, FwdLoad& f) {
    const uint32_t K = ix.k, L = l_L(s);
    const bool fresh = l_flags(s) & F_FRESH;
    const uint32_t xs = l_off(s) + (fresh ? K : 0u);                  // first position to compare: ref_offset (:227) as a window position
    const uint32_t kp0 = l_kp(s) + (fresh ? K : 0u);                  // kmer_pos += kmer_length (:215)
    const uint8_t* blk = chain_block(ix, s.h);                        // dbg.get_node (:210)
    const U4* sp = reinterpret_cast<const U4*>(blk);
    // the four slots ROTATED so that s0 is the record of the node the lane stands in, when the lane knows which slot that is (from the
    // dictionary entry, a link, or the step before in the same block); else as they lie (rot = 0) and the step looks for the record
    const uint32_t rot = l_cur(s);
    f.s0 = PA_LD(8, sp + rot); f.s1 = PA_LD(8, sp + ((rot + 1) & 3u)); f.s2 = PA_LD(8, sp + ((rot + 2) & 3u)); f.s3 = PA_LD(8, sp + ((rot + 3) & 3u));
    // sequence words this step can need, known before the block arrives: at most the rest of the read, 128 bases per step
    // (the second and third 16-byte load only go out for the lanes that can need them: every load is an access of the vector L1)
    const uint32_t most = pa_min(L - kp0, 128u), nwords = ((xs & 31) + most + 31) >> 5;
    const Q2* sq2 = reinterpret_cast<const Q2*>(blk + CH_SEQ_BYTES + 8u * (xs >> 5));
    f.s01 = PA_LD(8, sq2);
    f.s23 = f.s45 = Q2{0ull, 0ull};
    if (nwords > 2) f.s23 = PA_LD(8, sq2 + 1);
    if (nwords > 4) f.s45 = PA_LD(8, sq2 + 2);
}
#endif   // PA_LS_LANE

#if PA_LS_COMMON
// mismatches among the first z bases (0..128) of a step's compare masks m (bit 2i of word w = base 32 w + i differs), pc = their
// running popcounts
struct DiffMasks {   // (separate values, not arrays: see Slots)
    uint64_t m0, m1, m2, m3;
    uint32_t p0, p1, p2, p3;
};
PA_HD uint64_t diff_word(const DiffMasks& d, uint32_t w) { return sel4q(w, d.m0, d.m1, d.m2, d.m3); }
PA_HD uint32_t mism_prefix(const DiffMasks& d, uint32_t z) {
    const uint32_t w = z >> 5, r = z & 31u;
    const uint32_t b03 = sel4(w, 0u, d.p0, d.p1, d.p2), base = (w & 4u) ? d.p3 : b03;
    return base + pa_popc64(diff_word(d, w) & ((1ull << (2 * r)) - 1));          // (z == 128: r == 0, nothing of a fifth word)
}
#endif   // PA_LS_COMMON

#if PA_LS_LANE
// The general form of the step: any number of nodes per step (a loop), careful mode, list mode, node traces. The kernel runs
// it for the lanes that need one of these (fwd_finish below); the host emulator also runs it for every read of a traced batch.
template <bool TRACE = false>
PA_HD void fwd_finish_general(Lane& s, const DevIndexView& ix, ReadRef rd, ColRef cols, uint32_t allowed, const FwdLoad& f) {
    const uint32_t K = ix.k, L = l_L(s);
    const uint32_t fl = l_flags(s);
    const bool fresh = fl & F_FRESH, careful = fl & F_CAREFUL;
    uint32_t x = l_off(s) + (fresh ? K : 0u);                         // ref_offset (:227)
    uint32_t kp = l_kp(s) + (fresh ? K : 0u);                         // kmer_pos += kmer_length (:215)
    const uint32_t xs = x, kp0 = kp;
    uint32_t snp = s.rr >> 24, cov = l_cov(s), mism = l_mism(s);
    const uint32_t most = pa_min(L - kp0, 128u), nwords = ((xs & 31) + most + 31) >> 5;   // as in fwd_issue
    const uint64_t a[5] = {f.s01.a, f.s01.b, nwords > 2 ? f.s23.a : 0ull, nwords > 2 ? f.s23.b : 0ull, nwords > 4 ? f.s45.a : 0ull};
    const Slots sl{f.s0, f.s1, f.s2, f.s3};                           // (rotated by l_cur(s): fwd_issue)
    const bool known = s.of & OF_CUR_KNOWN;
    const uint32_t rot = l_cur(s);
    uint32_t cur = known ? 0u : seg_find(sl, x - 1);                  // the node of the k-mer ending at x - 1 / of the base x
    Seg g = seg_at(sl, cur);
    if (fresh) {
        cov += K;                                                     // :216
        if (push_node<TRACE>(s, cols, ix, g, 64ull * s.h + l_off(s))) {   // nodes.push (:219)
            restart_lists(s, K);
            return;
        }
        snp = 0;                                                      // :235
    }
    // five read words, all LDS reads in flight together. Words beyond the read's last are NOT zeroed here (they re-read the last
    // word): every compare below is limited to bases inside the read (n <= L - kp), so what lies beyond is never looked at
    uint64_t r[5];
    {
        const uint32_t i0 = (kp0 >> 5) * rd.stride, ilast = (rd.wmax - 1) * rd.stride;
#pragma unroll
        for (uint32_t i = 0; i < 5; ++i) r[i] = rd.p[pa_min(i0 + i * rd.stride, ilast)];
    }
    const uint32_t sh_a = (xs & 31) * 2, sh_r = (kp0 & 31) * 2;
    DiffMasks dm;
    dm.m0 = diff_mask(funnel(r[0], r[1], sh_r) ^ funnel(a[0], a[1], sh_a), 32u);
    dm.m1 = most > 32u ? diff_mask(funnel(r[1], r[2], sh_r) ^ funnel(a[1], a[2], sh_a), 32u) : 0ull;
    dm.m2 = most > 64u ? diff_mask(funnel(r[2], r[3], sh_r) ^ funnel(a[2], a[3], sh_a), 32u) : 0ull;
    dm.m3 = most > 96u ? diff_mask(funnel(r[3], r[4], sh_r) ^ funnel(a[3], a[4], sh_a), 32u) : 0ull;
    dm.p0 = pa_popc64(dm.m0); dm.p1 = dm.p0 + pa_popc64(dm.m1); dm.p2 = dm.p1 + pa_popc64(dm.m2); dm.p3 = dm.p2 + pa_popc64(dm.m3);
    const uint32_t lim = careful ? 32u : 128u;
    uint32_t consumed = 0, st = ST_FWD, h = s.h, nfl = fl & ~F_FRESH;
    uint32_t ncur = 0xFFFFFFFFu;                                      // `of` bits of the next step's slot when a hop decides them
    bool premature = false, hopped = false;
    for (;;) {
        const uint32_t n = pa_min(pa_min(g.e - x, L - kp), lim - consumed);   // max_matchable_pos (:222-231), as far as this step goes
        uint32_t matched;
        if (!careful) {
            const uint32_t cnt = mism_prefix(dm, consumed + n) - mism_prefix(dm, consumed);
            if (snp + cnt > allowed) { nfl |= F_CAREFUL; break; }    // over budget somewhere in these bases: redo this node carefully
            snp += cnt;
            mism += cnt;
            matched = n;
        } else {                                                      // (careful steps start at consumed == 0 and do one piece)
            matched = compare_chunk(n ? dm.m0 & (~0ull >> (64u - 2u * n)) : 0ull, n, allowed, snp, mism, premature);
        }
        x += matched; kp += matched; cov += matched; consumed += matched;   // :254, :257
        if (premature) { nfl &= ~F_CAREFUL; st = kp > L - K ? ST_ISECT : ST_SEEK; break; }   // :287-293 (a breaking base is left: kp < L)
        if (kp >= L) { nfl &= ~F_CAREFUL; st = ST_ISECT; break; }     // :259-261
        if (x < g.e) break;                                           // more of this node in the next step
        if (g.flags & SEG_LINK) {                                     // the chain's copy of this node ends here: on in the node's own chain
            const U4 lk = slot_at(sl, cur + 1 + ((g.flags & SEG_WIDE) ? 1u : 0u));
            h = lk.x;
            x = lk.y;
            ncur = lk.z;                                              // (slot of the node there, OF_CUR_KNOWN set: device_flatten.cpp)
            hopped = true;
            break;
        }
        nfl &= ~F_CAREFUL;                                            // node visit finished
        if (g.flags & SEG_LAST) {                                     // the chain's last node: its right edges (:265-283)
            uint32_t nh = NO_HANDLE;
            if (g.flags & SEG_EDGES) {
                const U4 ed = slot_at(sl, cur + 1 + ((g.flags & SEG_WIDE) ? 1u : 0u));
                const uint32_t b = read_base(rd, kp);                 // :265
                nh = sel4(b, ed.x, ed.y, ed.z, ed.w);                 // r_edges()[index].0 (:275-278); NO_HANDLE: !has_ext (:267)
            }
            if (nh != NO_HANDLE) {
                h = nh;
                x = 0;                                                // :279
                kp -= K - 1;                                          // :282
                cov -= K - 1;                                         // :283
                nfl |= F_FRESH;
                ncur = of_cur(0u, true);                              // a chain's first node is its first block's first record
                hopped = true;
            } else st = kp > L - K ? ST_ISECT : ST_SEEK;              // :287-293
            break;
        }
        if (consumed >= 128u) break;                                  // the base of the extension test is not in this step's masks
        const bool branch = (g.flags & SEG_EDGES) != 0;               // (not LAST:) several right extensions, the favoured one follows in the block
        {
            if ((diff_word(dm, consumed >> 5) >> (2 * (consumed & 31u))) & 1ull) {              // the chain's next base is not the read's
                uint32_t nh = NO_HANDLE;
                if (branch) {                                         // another right extension? the record's edge slot (:267-278)
                    const U4 ed = slot_at(sl, cur + 1 + ((g.flags & SEG_WIDE) ? 1u : 0u));
                    nh = sel4(read_base(rd, kp), ed.x, ed.y, ed.z, ed.w);
                }
                if (nh != NO_HANDLE) {                                // over the edge, as at a chain's end
                    h = nh;
                    x = 0;                                            // :279
                    kp -= K - 1;                                      // :282
                    cov -= K - 1;                                     // :283
                    nfl |= F_FRESH;
                    ncur = of_cur(0u, true);
                    hopped = true;
                } else st = kp > L - K ? ST_ISECT : ST_SEEK;          // !has_ext (:267), :287-293
                break;
            }
        }
        // the next node of the chain (:267-283 and the top of the loop :215-219): one base net, nothing of the K-1 overlap re-verified
        x += 1; kp += 1; cov += 1; consumed += 1;
        cur += 1 + ((g.flags & SEG_WIDE) ? 1u : 0u) + (branch ? 1u : 0u);
        g = seg_at(sl, cur);
        if (push_node<TRACE>(s, cols, ix, g, 64ull * s.h + x - K)) {  // nodes.push (:219); the node's first k-mer starts at x - K
            restart_lists(s, K);
            return;
        }
        snp = 0;                                                      // :235
        if (careful) break;
    }
    nfl |= l_flags(s) & (F_SMALL_BASE | F_SPILL_OVERFLOW);           // (what push_node may have set)
    if (st == ST_SEEK) nfl |= F_SPEC;                                 // a re-seek starts at the base the visit broke off at: a probable miss
    if (st == ST_FWD && !hopped) {                                    // goes on in this chain: the block whose window starts at most 64 bases before x
        const uint32_t adv = (x - 1) >> CH_STRIDE_LOG2;               // (x >= 1; a position that is a multiple of 64 stays the 64th of the block before:
        h += adv;                                                     //  a node that ends exactly there is still on that block's list)
        x -= adv << CH_STRIDE_LOG2;
        ncur = of_cur((rot + cur) & 3u, adv == 0);                    // the same block: the node's slot is known; another block: the next step looks for it
    }
    s.h = h;
    l_set_lk(s, L, kp, st);
    l_set_cm(s, cov, mism);
    s.rr = snp << 24;
    s.of = x | (st == ST_FWD ? ncur : 0u) | (nfl << 24);
}
#endif   // PA_LS_LANE

#if PA_LS_COMMON
// nodes.push in window mode on values held in registers (push_node's window branch without its memory traffic): w / cand /
// have = the running windows, the class id the running intersection is known to be, whether a window is held yet
PA_HD void push_window(bool on, U4& w, uint32_t& cand, bool& have, uint32_t cid, uint32_t cmin, uint32_t cmask, uint32_t cmin2, uint32_t cmask2) {
    const uint32_t m1 = w.y & (window_at(w.x, cmin, cmask) | window_at(w.x, cmin2, cmask2));
    const uint32_t m2 = w.w & (window_at(w.z, cmin, cmask) | window_at(w.z, cmin2, cmask2));
    const bool full = pa_popc32(m1) + pa_popc32(m2) == pa_popc32(cmask) + pa_popc32(cmask2);   // this class is a subset of all before: it IS the result
    const bool same = m1 == w.y && m2 == w.w;                                                   // nothing removed
    const uint32_t cand_h = full ? cid : same ? cand : NO_CLASS;
    const bool first = on && !have, later = on && have;
    w.x = first ? cmin : w.x;
    w.y = first ? cmask : later ? m1 : w.y;
    w.z = first ? cmin2 : w.z;
    w.w = first ? cmask2 : later ? m2 : w.w;
    cand = first ? cid : later ? cand_h : cand;
    have = have || on;
}
#endif   // PA_LS_COMMON

#if PA_LS_LANE
// The step as the kernel's common lanes take it — window mode, not careful, no node trace — in straight-line code: the node the
// step starts in (A) and, when the read runs over A's end into the chain's next node (B), B as well; a third node of the same
// block is left to the next step, which finds the lane standing at B's end. Every decision is a select, the class windows are
// read from and written to LDS once. Lanes in careful or list mode (and traced batches) take fwd_finish_general.
template <bool TRACE = false>
PA_HD void fwd_finish(Lane& s, const DevIndexView& ix, ReadRef rd, ColRef cols, uint32_t allowed, const FwdLoad& f) {
    const uint32_t fl = l_flags(s);
#ifndef PA_PROBE_FAST_ONLY   // (tools: instruction count of the straight-line part alone)
    if (TRACE || (fl & (F_LISTS | F_CAREFUL))) {
        fwd_finish_general<TRACE>(s, ix, rd, cols, allowed, f);
        return;
    }
#endif
    const uint32_t K = ix.k, L = l_L(s);
    const bool fresh = fl & F_FRESH;
    const uint32_t kadd = fresh ? K : 0u;
    const uint32_t x0 = l_off(s) + kadd, kp0 = l_kp(s) + kadd;       // ref_offset (:227), kmer_pos += kmer_length (:215)
    const uint32_t snp0 = fresh ? 0u : s.rr >> 24;                    // :235
    const uint64_t a0 = f.s01.a, a1 = f.s01.b, a2 = f.s23.a, a3 = f.s23.b, a4 = f.s45.a;   // (words fwd_issue did not load are zero there; never looked at)
    // t0 = the record of the node this step starts in. A lane that knows its slot had the block's slots loaded rotated by it
    // (fwd_issue); one that does not (it moved on to another block of its chain) finds the record — the first one that ends beyond
    // x0 - 1 — and rotates what it loaded
#ifdef PA_PROBE_KNOWN   // (tools/isa_probe.hip: the common lane's text alone)
    const bool known = true;
#else
    const bool known = s.of & OF_CUR_KNOWN;
#endif
    U4 t0 = f.s0, t1 = f.s1, t2 = f.s2, t3 = f.s3;
    uint32_t curp = l_cur(s);                                        // the slot of t0 in the block
    if (!known) {
        const Slots sl{f.s0, f.s1, f.s2, f.s3};
        const uint32_t cur = seg_find(sl, x0 - 1);
        const bool c1 = cur & 1u, c2 = cur & 2u;
#define PA_ROT1(fld) const uint32_t u0##fld = c1 ? f.s1.fld : f.s0.fld, u1##fld = c1 ? f.s2.fld : f.s1.fld, u2##fld = c1 ? f.s3.fld : f.s2.fld, u3##fld = c1 ? f.s0.fld : f.s3.fld;
        PA_ROT1(x) PA_ROT1(y) PA_ROT1(z) PA_ROT1(w)
#undef PA_ROT1
        t0 = U4{c2 ? u2x : u0x, c2 ? u2y : u0y, c2 ? u2z : u0z, c2 ? u2w : u0w}; t1 = U4{c2 ? u3x : u1x, c2 ? u3y : u1y, c2 ? u3z : u1z, c2 ? u3w : u1w};
        t2 = U4{c2 ? u0x : u2x, c2 ? u0y : u2y, c2 ? u0z : u2z, c2 ? u0w : u2w}; t3 = U4{c2 ? u1x : u3x, c2 ? u1y : u3y, c2 ? u1z : u3z, c2 ? u1w : u3w};
        curp = cur;
    }
    const bool wideA = t0.x & SEG_WIDE, lastA = t0.x & SEG_LAST;
    const bool branchA = (t0.x & (SEG_EDGES | SEG_LAST)) == SEG_EDGES;   // several right extensions, a copy of the favoured one follows: record [ext] edges B [ext]
    const U4 ed{wideA ? t2.x : t1.x, wideA ? t2.y : t1.y, wideA ? t2.z : t1.z, wideA ? t2.w : t1.w};   // the slot behind A's record: A's right edges / link — or B's record
    const U4 e2{wideA ? t3.x : t2.x, wideA ? t3.y : t2.y, wideA ? t3.z : t2.z, wideA ? t3.w : t2.w};   // ... and the one behind that
    const U4 nx{branchA ? e2.x : ed.x, branchA ? e2.y : ed.y, branchA ? e2.z : ed.z, branchA ? e2.w : ed.w};   // B's record
    // B's extension slot (a wide B behind a wide branch record would be a fifth slot: the flattener never lays that out)
    const U4 xb{branchA ? t3.x : e2.x, branchA ? t3.y : e2.y, branchA ? t3.z : e2.z, branchA ? t3.w : e2.w};
    const bool wideB = nx.x & SEG_WIDE;
    // five read words, all LDS reads in flight together (words beyond the read's last re-read the last one: never looked at)
    uint64_t r0, r1, r2, r3, r4;
    {
        const uint32_t i0 = (kp0 >> 5) * rd.stride, ilast = rd.slack ? 0xFFFFFFFFu : (rd.wmax - 1) * rd.stride;
        r0 = rd.p[pa_min(i0, ilast)]; r1 = rd.p[pa_min(i0 + rd.stride, ilast)]; r2 = rd.p[pa_min(i0 + 2 * rd.stride, ilast)];
        r3 = rd.p[pa_min(i0 + 3 * rd.stride, ilast)]; r4 = rd.p[pa_min(i0 + 4 * rd.stride, ilast)];
    }
    const uint32_t sh_a = (x0 & 31) * 2, sh_r = (kp0 & 31) * 2;
    DiffMasks dm;   // (bases beyond the read's end are never counted: every n below is limited to L - kp)
    dm.m0 = diff_mask(funnel(r0, r1, sh_r) ^ funnel(a0, a1, sh_a), 32u);
    dm.m1 = diff_mask(funnel(r1, r2, sh_r) ^ funnel(a1, a2, sh_a), 32u);
    dm.m2 = diff_mask(funnel(r2, r3, sh_r) ^ funnel(a2, a3, sh_a), 32u);
    dm.m3 = diff_mask(funnel(r3, r4, sh_r) ^ funnel(a3, a4, sh_a), 32u);
    dm.p0 = pa_popc64(dm.m0); dm.p1 = dm.p0 + pa_popc64(dm.m1); dm.p2 = dm.p1 + pa_popc64(dm.m2); dm.p3 = dm.p2 + pa_popc64(dm.m3);
    // ---- node A: the compare loop (:236-255) as a count
    const uint32_t eA = t0.x & SEG_E_MASK;
    const uint32_t nA = pa_min(pa_min(eA - x0, L - kp0), 128u);       // max_matchable_pos (:222-231), as far as this step goes
    const uint32_t pA = mism_prefix(dm, nA), cntA = pA;
    const bool okA = snp0 + cntA <= allowed;                          // else: the node visit breaks off inside these bases (:243-249)
    uint32_t brk = 0;                                                 // ... at this base: the (allowed - snp0 + 1)-th mismatch of the visit
    if (!okA) {
        const uint32_t tol = allowed - snp0;                          // mismatches still within budget (seen_snp never exceeds allowed between steps)
        const uint32_t w = (uint32_t)(tol >= dm.p0) + (uint32_t)(tol >= dm.p1) + (uint32_t)(tol >= dm.p2);   // the word that holds it (it lies among the first nA bases)
        uint32_t skip = tol - sel4(w, 0u, dm.p0, dm.p1, dm.p2);
        uint64_t mw = diff_word(dm, w);
        for (; skip; --skip) mw &= mw - 1;
        brk = 32u * w + (pa_ctz64(mw) >> 1);
    }
    const uint32_t kpA = kp0 + nA, xA = x0 + nA;                      // :257
    const bool endA = okA && xA == eA && kpA < L;                     // node visit finished, read not (:259-261)
    // ---- A's end: the chain's next node, or the chain's right edges
    const bool bitA = (diff_word(dm, nA >> 5) >> (2 * (nA & 31u))) & 1ull;   // the chain's next base differs from the read's (nA < 128)
    const uint32_t bA = read_base(rd, pa_min(kpA, L - 1));            // :265
    const bool linkA = t0.x & SEG_LINK;                               // a copy cut short: the same node goes on in its own chain
    const uint32_t edge = (t0.x & SEG_EDGES) ? sel4(bA, ed.x, ed.y, ed.z, ed.w) : linkA ? ed.x : NO_HANDLE;   // r_edges()[index].0 (:275-278)
    const bool in_masks = nA < 128u;
    const bool other = endA && !lastA && in_masks && bitA;            // the chain's next base is not the read's: at a branch record another right extension may be
    const bool hop_chain = endA && (lastA || (other && branchA)) && edge != NO_HANDLE, hop_edge = hop_chain && !linkA;
    const bool hopB = endA && !lastA && in_masks && !bitA;            // has_ext(Right, b) (:267): the chain's next node
    const bool dead = endA && (lastA ? edge == NO_HANDLE : (other && !hop_chain));   // :287-293
    // ---- node B
    const uint32_t c1n = nA + 1, x1 = xA + 1, kp1 = kpA + 1;          // the hop: one base net (:282-283, :215-216)
    const uint32_t eB = nx.x & SEG_E_MASK;
    const uint32_t nB = hopB ? pa_min(pa_min(eB - x1, L - kp1), 128u - c1n) : 0u;
    const uint32_t cntB = mism_prefix(dm, c1n + nB) - pA;            // (bit nA is clear when B is entered)
    const bool okB = cntB <= allowed;
    const bool useB = hopB && okB;
    const uint32_t kpB = kp1 + nB, xB = x1 + nB;
    // ---- classes: nodes.push (:219) of A (a fresh entry) and of B, on the windows held in LDS
    const uint32_t ncol = l_ncol(s);
    U4 w = *reinterpret_cast<const U4*>(cols.win);
    uint32_t cand = cols.wcand[0];
    bool have = ncol & 1u;
    const bool pushA = fresh, pendA = pushA && t0.w == 0, pendB = hopB && nx.w == 0;
    push_window(pushA && !pendA, w, cand, have, t0.y, t0.z, t0.w, wideA ? t1.x : 0u, wideA ? t1.y : 0u);
    push_window(hopB && !pendB, w, cand, have, nx.y, nx.z, nx.w, wideB ? xb.x : 0u, wideB ? xb.y : 0u);
    uint32_t np = ncol >> 1;
    bool restart = false;
    if (pendA | pendB) {                                              // classes without windows: noted, applied after the walk (push_node)
        if (pendA) {
            if (np >= PEND_MAX || 2 * np + 1 >= cols.spill_cap) restart = true;
            else { cols.pend[2 * np] = t1.z; cols.pend[2 * np + 1] = t1.w; ++np; }
        }
        if (pendB && !restart) {
            if (np >= PEND_MAX || 2 * np + 1 >= cols.spill_cap) restart = true;
            else { cols.pend[2 * np] = xb.z; cols.pend[2 * np + 1] = xb.w; ++np; }
        }
    }
    if (restart) { restart_lists(s, K); return; }
    if (pushA | hopB) {
        *reinterpret_cast<U4*>(cols.win) = w;
        cols.wcand[0] = cand;
    }
    s.nc = (s.nc & ~NC_COL_MASK) | (np << 1) | (have ? 1u : 0u);
    // ---- the lane's next state
    // A broken off: the breaking base is counted as a mismatch (:244) but not as matched (:247-249), the walk re-seeks from it or
    // ends (:287-293). B over budget: nothing of B is consumed, the next step starts in it and breaks it off the same way
    const bool ended = useB ? kpB >= L : (okA && kpA >= L);           // :259-261
    const uint32_t kp_dead = okA ? kpA : kp0 + brk;
    const bool deadA = dead || !okA;
    uint32_t st = (ended || (deadA && kp_dead > L - K)) ? (uint32_t)ST_ISECT : deadA ? (uint32_t)ST_SEEK : (uint32_t)ST_FWD;
    uint32_t kp = !okA ? kp_dead : hop_edge ? kpA - (K - 1) : hopB ? (okB ? kpB : kp1) : kpA;
    uint32_t x = !okA ? x0 : hop_chain ? (linkA ? ed.y : 0u) : hopB ? (okB ? xB : x1) : xA;
    const uint32_t cov = l_cov(s) + kadd + (okA ? nA : brk) + (hopB ? 1u : 0u) + (useB ? nB : 0u) - (hop_edge ? K - 1 : 0u);   // :216, :254, :283
    const uint32_t mism = l_mism(s) + (okA ? cntA : allowed - snp0 + 1) + (useB ? cntB : 0u);
    const uint32_t snp = hopB ? (okB ? cntB : 0u) : snp0 + cntA;
    const uint32_t nfl = (fl & ~(F_FRESH | F_CAREFUL)) | (hop_edge ? F_FRESH : 0u) | (st == ST_SEEK ? F_SPEC : 0u);
    uint32_t h = hop_chain ? edge : s.h;
    // the slot of the node the next step starts in: a chain entered over an edge starts with its first record; a link names it
    // (nx.z); in this block it is A's or B's slot; in another block of this chain the next step has to look for it
    uint32_t ncur = linkA ? ed.z : of_cur(0u, true);
    if (st == ST_FWD && !hop_chain) {                                 // goes on in this chain (fwd_finish_general)
        const uint32_t adv = (x - 1) >> CH_STRIDE_LOG2;
        h += adv;
        x -= adv << CH_STRIDE_LOG2;
        ncur = of_cur((curp + (hopB ? 1u + (wideA ? 1u : 0u) + (branchA ? 1u : 0u) : 0u)) & 3u, adv == 0);
    }
    s.h = h;
    l_set_lk(s, L, kp, st);
    l_set_cm(s, cov, mism);
    s.rr = snp << 24;
    s.of = x | (st == ST_FWD ? ncur : 0u) | (nfl << 24);
}

// Forward search (:209-301): one call = one chain block
template <bool TRACE = false>
PA_HD void fwd_step(Lane& s, const DevIndexView& ix, ReadRef rd, ColRef cols, uint32_t allowed) {
    FwdLoad f;
    fwd_issue(s, ix, f);
    fwd_finish<TRACE>(s, ix, rd, cols, allowed, f);
}
#endif   // PA_LS_LANE

#if PA_LS_COMMON
// ---------------------------------------------------------------------------------------------- LEFT
// Left extension (:131-203): one call = one chain block (its slots and the two sequence words around the bases to compare, one
// round trip), up to 32 bases leftwards. In chain coordinates the reference's hop to a left neighbour (:183-199: has_ext(Left,
// read[last_pos]), l_edges, prev_kmer_offset = len - k) is: the base just left of the node's first base must equal the read's,
// and the compare goes on AT that base with a fresh mismatch budget. At the chain's first base the left-edge table (by chain
// handle) names the block and position to go on in.
struct LeftLoad {
    U4 s0, s1, s2, s3;
    Q2 sq;           // the two sequence words around the bases to compare
};
#endif   // PA_LS_COMMON
#if PA_LS_LANE
PA_HD uint32_t left_y1(const Lane& s) { return s.rr & 0xFFFFFFu; }   // window position + 1 of the next base to compare
PA_HD void left_issue(const Lane& s, const DevIndexView& ix, LeftLoad& f) {
    const uint8_t* blk = chain_block(ix, s.ph);                      // dbg.get_node(prev_node_id) (:132)
    const U4* sp = reinterpret_cast<const U4*>(blk);
    f.s0 = sp[0]; f.s1 = sp[1]; f.s2 = sp[2]; f.s3 = sp[3];
    const uint32_t y1 = left_y1(s), st = y1 > 32 ? y1 - 33 : 0;      // 33 bases ending at y = y1 - 1: the compare and the extension test
    f.sq = *reinterpret_cast<const Q2*>(blk + CH_SEQ_BYTES + 8u * (st >> 5));
}
template <bool TRACE = false>
PA_HD void left_finish(Lane& s, const DevIndexView& ix, ReadRef rd, ColRef cols, uint32_t allowed, const LeftLoad& f) {
    const uint32_t K = ix.k;
    const Slots sl{f.s0, f.s1, f.s2, f.s3};
    uint32_t y1 = left_y1(s), snp = s.rr >> 24, ra = l_left_ra(s), cov = l_cov(s), mism = l_mism(s);
    const uint32_t y = y1 - 1;                                       // ref_pos of idx 0 (:152) as a window position
    const uint32_t ws = (y1 > 32 ? y1 - 33 : 0) >> 5;                // first sequence word loaded
    const uint32_t fl = l_flags(s), recmask = f.s0.x >> SEG_RECMASK_SHIFT, back = (f.s0.x >> SEG_BACK_SHIFT) & CH_BACK_MAX;
    const uint32_t cur = seg_find(sl, y + K - 1);                    // prev_node_id: the node whose k-mer starts at y
    const uint32_t below = recmask & ((1u << cur) - 1);              // records before it in this block
    const uint32_t prev = (below & 4u) ? 2u : (below >> 1) & 1u;
    // window position of the node's first base: the node before ends K - 1 bases later (device_layout.hpp); -1: left of the window
    int32_t s_lo = below ? (int32_t)(slot_at(sl, prev).x & SEG_E_MASK) - (int32_t)(K - 1) : (back == 0 ? 0 : -1);
    if (s_lo < 0) s_lo = -1;
    const uint32_t lo = s_lo < 0 ? 0u : (uint32_t)s_lo;
    // The node starts exactly at window position 0 of a block that is not its chain's first (the node before it still has its
    // record here: below != 0, back > 0), and this step could reach that base: the base the extension test (:183) looks at lies
    // one position to the left of this window. Step back FIRST, consuming nothing — in the earlier block the same node starts at
    // 64 * back > 0 and its in-chain hop is the ordinary case below. (Window position 0 with records before it is never the
    // chain's first base: only `!below && back == 0` is.)
    if (below && s_lo == 0 && y1 <= 32u) {
        s.ph -= back;
        s.rr = (y1 + CH_STRIDE * back) | (snp << 24);
        return;
    }
    if (fl & F_FRESH) {
        if (!(fl & F_LEFT_SEED)) {
            if (push_node<TRACE>(s, cols, ix, seg_at(sl, cur), 64ull * s.ph + y)) {   // nodes.push(prev_node.node_id) (:199)
                restart_lists(s, K);
                return;
            }
        }
        snp = 0;                                                    // :150
        l_clr_flags(s, F_FRESH | F_LEFT_SEED);
    }
    bool premature = false;
    const uint32_t n = pa_min(pa_min(ra, y1 - lo), 32u);            // max_matchable_pos (:139-145), 32 bases per step
    uint32_t matched = 0;
    if (n > 0) {
        const uint32_t lp = ra - 1, pin = y - 32u * ws;              // read_offset of idx 0 (:153); y within the two words
        const uint64_t sw = pin >= 63 ? f.sq.b : pin >= 31 ? funnel(f.sq.a, f.sq.b, (pin - 31) * 2) : f.sq.a << (2 * (31 - pin));   // 32 bases ending at y
        // base idx 0 sits in the top bits: fold each base's two XOR bits onto its odd bit, then bit-reverse so that
        // bit 2i = i-th base compared
        const uint64_t xr = read_window_end(rd, lp) ^ sw;
        const uint64_t mm = pa_brev64((xr | (xr << 1)) & 0xAAAAAAAAAAAAAAAAull);
        const uint64_t keep = n >= 32 ? ~0ull : ((1ull << (2 * n)) - 1);
        matched = compare_chunk(mm & keep, n, allowed, snp, mism, premature);
    }
    ra -= matched;                                                  // last_pos -= matched_bases (:178)
    y1 -= matched;
    cov += matched;                                                 // :169
    l_set_cm(s, cov, mism);
    s.rr = y1 | (snp << 24);
    l_set_left_ra(s, ra);
    if (!(ra == 0 || premature)) {                                  // :173-175
        if (y1 > lo) return;                                        // more of this node in this block
        if (s_lo < 0) {                                             // the node goes on to the left of the window: an earlier block
            if (back) {
                s.ph -= back;
                s.rr = (y1 + CH_STRIDE * back) | (snp << 24);
                return;
            }
        } else {
            const uint32_t b = read_base(rd, ra - 1);               // next_base = read_seq.get(last_pos) (:182)
            if (s_lo > 0) {                                         // inside the chain: the node's one left extension is the base before it
                const uint32_t q = (uint32_t)s_lo - 1 - 32u * ws;
                const uint32_t cb = (uint32_t)((q < 32 ? f.sq.a : f.sq.b) >> (2 * (q & 31u))) & 3u;
                if (cb == b) {                                      // has_ext(Dir::Left, b) (:183): on into that node at its last k-mer (:196)
                    if (push_node<TRACE>(s, cols, ix, seg_at(sl, prev), 64ull * s.ph + (uint32_t)s_lo - 1)) {   // :199
                        restart_lists(s, K);
                        return;
                    }
                    s.rr = y1;                                      // snp = 0 (:150)
                    return;
                }
            } else {                                                // the chain's first base: l_edges()[index].0 (:191-194) by chain handle
                const uint64_t e = *reinterpret_cast<const uint64_t*>(ix.ledge + 8ull * s.ph + 2 * b);
                if ((uint32_t)e != NO_HANDLE) {
                    s.ph = (uint32_t)e;
                    s.rr = (uint32_t)(e >> 32);                     // the neighbour's last k-mer (:196)
                    l_or_flags(s, F_FRESH);
                    return;
                }
            }                                                       // else :200-202
        }
    }
    l_set_st(s, ST_FWD);                                            // forward search from the seed (:208)
    l_or_flags(s, F_FRESH);
}
template <bool TRACE = false>
PA_HD void left_step(Lane& s, const DevIndexView& ix, ReadRef rd, ColRef cols, uint32_t allowed) {
    LeftLoad f;
    left_issue(s, ix, f);
    left_finish<TRACE>(s, ix, rd, cols, allowed, f);
}
#endif   // PA_LS_LANE

#if PA_LS_COMMON
// ---------------------------------------------------------------------------------------------- ISECT
// nodes_to_eq_class (:323-356) + intersect (:389-418): the class is the intersection of the id lists of every visited
// node's class. Base list = a shortest one (what the stable sort at :331-334 puts first). Regimes:
//   register tier  <= 4 classes, base <= 7 ids: base ids in registers (two 16-byte loads of its record); other lists of
//                  <= 7 ids are compared all-pairs in registers, longer ones by binary search
//   generic tier   base <= 64 ids: survivors tracked as a 64-bit mask, membership by binary search; longer bases are
//                  counted, then recomputed in the write pass
struct Isect {
    uint32_t base_ref, base_len, base_colour, count;
    uint32_t base_slot;   // which of the lane's LDS class slots holds the base list (tier 0)
    uint64_t alive;       // survivors as a mask over the base list (base_len <= 64)
    uint32_t ids[7];      // register tier only: the base list itself (ids beyond base_len are the 0xFFFFFFFF padding)
    bool in_regs;
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

PA_HD const uint32_t* class_ids(const DevIndexView& ix, uint32_t ec_ref) { return ix.ec + 4ull * ec_ref + 1; }

PA_HD void get_class(ColRef c, uint32_t i, uint32_t& ec_ref, uint32_t& ec_len) {
    if (i < LDS_CLASSES) {
        ec_ref = c.refs[i];
        ec_len = c.lens[i];
    } else {
        ec_ref = c.spill[4 * (i - LDS_CLASSES)];
        ec_len = c.spill[4 * (i - LDS_CLASSES) + 1];
    }
}
PA_HD uint32_t get_class_id(ColRef c, uint32_t i) { return i < LDS_CLASSES ? c.cids[i] : c.spill[4 * (i - LDS_CLASSES) + 2]; }

PA_HD bool in_all_lists(const DevIndexView& ix, ColRef cols, uint32_t ncol, uint32_t base_ref, uint32_t v) {
    for (uint32_t i = 0; i < ncol; ++i) {
        uint32_t ref, len;
        get_class(cols, i, ref, len);
        if (ref == base_ref) continue;
        if (!list_contains(class_ids(ix, ref), len, v)) return false;
    }
    return true;
}

// 1 iff v is one of the seven ids (unused slots hold the 0xFFFFFFFF record padding, which no transcript id equals)
PA_HD uint32_t any_eq7(uint32_t v, const uint32_t (&o)[7]) {
    return (uint32_t)((o[0] == v) | (o[1] == v) | (o[2] == v) | (o[3] == v) | (o[4] == v) | (o[5] == v) | (o[6] == v));
}
#endif   // PA_LS_COMMON

#if PA_LS_LANE
// Step 1: the base list (a shortest one) and the tier that will intersect it:
//   0  one class (the result IS that class: nothing to load), or <= 3 classes with every list <= 7 ids: registers only,
//      no dependent loads                                                                       (isect_light)
//   1  base <= 8 ids, other lists long and/or more than 4 classes: base in registers, the other lists are scanned
//      with 16-byte loads whose addresses are all known up front                                (isect_scan)
//   2  base > 8 ids and at least two classes: the whole wave works on one read                   (kernel, cooperative)
// The host emulator treats tier 2 with per-lane binary searches (isect_count).
PA_HD uint32_t isect_pick(const Lane& s, ColRef cols, Isect& r) {
    r.alive = 0;
    r.count = 0;
    r.in_regs = false;
    const uint32_t ncol = l_ncol(s);
    if (ncol > 3) {   // the walk kept the shortest class: nothing to load. base_colour = word 0 of the record (class_colour)
        r.base_ref = cols.win[3];
        r.base_len = cols.wcand[0];
        r.base_colour = NO_CLASS;
        r.base_slot = 0;
        return r.base_len <= 8 ? 1u : 2u;
    }
    const U4 refs = *reinterpret_cast<const U4*>(cols.refs), lens = *reinterpret_cast<const U4*>(cols.lens),
             cids = *reinterpret_cast<const U4*>(cols.cids);
    const uint32_t ln1 = ncol > 1 ? lens.y : lens.x, ln2 = ncol > 2 ? lens.z : lens.x;
    r.base_len = lens.x;
    r.base_ref = refs.x;
    r.base_colour = cids.x;
    r.base_slot = 0;
    if (ln1 < r.base_len) { r.base_len = ln1; r.base_ref = refs.y; r.base_colour = cids.y; r.base_slot = 1; }
    if (ln2 < r.base_len) { r.base_len = ln2; r.base_ref = refs.z; r.base_colour = cids.z; r.base_slot = 2; }
    uint32_t maxlen = lens.x > ln1 ? lens.x : ln1;
    maxlen = maxlen > ln2 ? maxlen : ln2;
    if (ncol == 1) {                                                // eq_class = eq_classes[colour] (:346-350), no intersection
        r.count = r.base_len;
        return 0;
    }
    if (maxlen <= 7) return 0;
    return r.base_len <= 8 ? 1u : 2u;
}
#endif   // PA_LS_LANE
#if PA_LS_COMMON
// the class id of a class record (its first word)
PA_HD uint32_t class_colour(const DevIndexView& ix, uint32_t ec_ref) { return ix.ec[4ull * ec_ref]; }

PA_HD uint32_t match7(const U4& o0, const U4& o1, const uint32_t (&b)[7]) {
    const uint32_t o[7] = {o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
    uint32_t m = 0;
#pragma unroll
    for (int j = 0; j < 7; ++j) m |= any_eq7(b[j], o) << j;
    return m;
}
#endif   // PA_LS_COMMON

#if PA_LS_LANE
// Tier 0: the records of the (<= 3) classes are fetched together (two 16-byte loads each, one round trip); the base ids
// are compared all-pairs with the other lists in registers; survivors are a 7-bit mask over the base list.
PA_HD void isect_light(const Lane& s, const DevIndexView& ix, ColRef cols, Isect& r) {
    const uint32_t ncol = l_ncol(s);
    if (ncol == 1) return;                                          // the class itself, returned by reference
    const U4 refs = *reinterpret_cast<const U4*>(cols.refs);
    const U4* p0 = reinterpret_cast<const U4*>(ix.ec + 4ull * refs.x);
    const U4* p1 = reinterpret_cast<const U4*>(ix.ec + 4ull * refs.y);
    const U4* p2 = reinterpret_cast<const U4*>(ix.ec + 4ull * (ncol > 2 ? refs.z : refs.x));   // unused slot: re-read slot 0
    const U4 a0 = p0[0], a1 = p0[1], b0 = p1[0], b1 = p1[1], c0 = p2[0], c1 = p2[1];           // six loads in flight together
    const uint32_t bs = r.base_slot;
    const U4 q0 = bs == 0 ? a0 : bs == 1 ? b0 : c0, q1 = bs == 0 ? a1 : bs == 1 ? b1 : c1;
    r.in_regs = true;
    r.ids[0] = q0.y; r.ids[1] = q0.z; r.ids[2] = q0.w; r.ids[3] = q1.x; r.ids[4] = q1.y; r.ids[5] = q1.z; r.ids[6] = q1.w;
    uint32_t alive = (1u << r.base_len) - 1;
    if (bs != 0) alive &= match7(a0, a1, r.ids);
    if (bs != 1) alive &= match7(b0, b1, r.ids);
    if (bs != 2 && ncol > 2) alive &= match7(c0, c1, r.ids);
    r.alive = alive;
    r.count = pa_popc32(alive);
}
#endif   // PA_LS_LANE

#if PA_LS_COMMON
PA_HD uint32_t eq_mask8(uint32_t v, const uint32_t (&b)[8]) {
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) m |= (b[i] == v ? 1u : 0u) << i;
    return m;
}

// Tier 1: base list of <= 8 ids in registers; every other list is streamed through 16-byte loads (record words
// {class id, id0, id1, id2}, {id3..id6}, ..., 0xFFFFFFFF padded) and each word is compared with the eight base ids.
// The first 11 ids of the classes held in LDS are fetched two classes at a time (six loads in flight); longer lists and
// classes spilled to HBM take the sequential tail loop.
PA_HD uint32_t scan_words(const U4& w, bool first, const uint32_t (&b)[8]) {
    return (first ? 0u : eq_mask8(w.x, b)) | eq_mask8(w.y, b) | eq_mask8(w.z, b) | eq_mask8(w.w, b);   // word 0 of a record is its class id
}

PA_HD uint32_t scan_tail(const DevIndexView& ix, uint32_t ref, uint32_t len, uint32_t from_chunk, uint32_t alive, const uint32_t (&b)[8]) {
    uint32_t m = 0;
    if (len <= 64) {
        const U4* rec = reinterpret_cast<const U4*>(ix.ec + 4ull * ref);
        const uint32_t nchunks = (len + 4) >> 2;
#pragma unroll 1
        for (uint32_t q0 = from_chunk; q0 < nchunks; q0 += 2) {      // two 16-byte loads in flight per round trip
            const U4 w0 = rec[q0], w1 = rec[q0 + 1 < nchunks ? q0 + 1 : q0];
            m |= scan_words(w0, q0 == 0, b);
            if (q0 + 1 < nchunks) m |= scan_words(w1, false, b);
        }
    } else {                                                         // long list: binary_search (:404) per surviving base id
        const uint32_t* ids = class_ids(ix, ref);
#pragma unroll 1
        for (uint32_t t = alive; t; t &= t - 1) {
            const uint32_t j = pa_ctz32(t);
            const uint32_t v = j == 0 ? b[0] : j == 1 ? b[1] : j == 2 ? b[2] : j == 3 ? b[3] : j == 4 ? b[4] : j == 5 ? b[5] : j == 6 ? b[6] : b[7];
            if (list_contains(ids, len, v)) m |= 1u << j;
        }
    }
    return m;
}

// membership of the base ids in up to four other lists: the first three chunks (11 ids) of each are fetched together —
// twelve loads, one round trip — longer lists continue in scan_tail
PA_HD void scan_quad(const DevIndexView& ix, const uint32_t (&ref)[4], const uint32_t (&len)[4], const bool (&use)[4], uint32_t base_ref,
                     const uint32_t (&b)[8], uint32_t& alive) {
    const U4* brec = reinterpret_cast<const U4*>(ix.ec + 4ull * base_ref);
    const U4* rec[4];
    uint32_t n[4];
    U4 w[4][3];
#pragma unroll
    for (uint32_t t = 0; t < 4; ++t) {
        rec[t] = use[t] ? reinterpret_cast<const U4*>(ix.ec + 4ull * ref[t]) : brec;   // an unused slot re-reads the base record
        n[t] = use[t] && len[t] <= 64 ? (len[t] + 4) >> 2 : 0;
#pragma unroll
        for (uint32_t q = 0; q < 3; ++q) w[t][q] = rec[t][q < n[t] ? q : 0];
    }
#pragma unroll
    for (uint32_t t = 0; t < 4; ++t) {
        uint32_t m = 0;
#pragma unroll
        for (uint32_t q = 0; q < 3; ++q)
            if (q < n[t]) m |= scan_words(w[t][q], q == 0, b);
        if (use[t]) {
            if (n[t] == 0 || n[t] > 3) m |= scan_tail(ix, ref[t], len[t], n[t] == 0 ? 0 : 3, alive, b);
            alive &= m;
        }
    }
}
#endif   // PA_LS_COMMON

#if PA_LS_LANE
PA_HD void isect_scan(const Lane& s, const DevIndexView& ix, ColRef cols, Isect& r) {
    const uint32_t ncol = l_ncol(s);
    const U4 refs = *reinterpret_cast<const U4*>(cols.refs), lens = *reinterpret_cast<const U4*>(cols.lens);
    const U4* brec = reinterpret_cast<const U4*>(ix.ec + 4ull * r.base_ref);
    const U4 q0 = brec[0], q1 = brec[1], q2 = brec[2];
    const uint32_t b[8] = {q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, r.base_len > 7 ? q2.x : 0xFFFFFFFFu};
    uint32_t alive = (1u << r.base_len) - 1;
    {
        const uint32_t ref[4] = {refs.x, refs.y, refs.z, refs.w}, len[4] = {lens.x, lens.y, lens.z, lens.w};
        const bool use[4] = {refs.x != r.base_ref, ncol > 1 && refs.y != r.base_ref, ncol > 2 && refs.z != r.base_ref,
                             ncol > 3 && refs.w != r.base_ref};
        scan_quad(ix, ref, len, use, r.base_ref, b, alive);
    }
#pragma unroll 1
    for (uint32_t i = LDS_CLASSES; i < ncol && alive; i += 4) {      // classes spilled to HBM, four at a time
        uint32_t ref[4], len[4];
        bool use[4];
#pragma unroll
        for (uint32_t t = 0; t < 4; ++t) {
            const U4 qd = *reinterpret_cast<const U4*>(cols.spill + 4 * (i + t < ncol ? i + t - LDS_CLASSES : i - LDS_CLASSES));
            ref[t] = qd.x;
            len[t] = qd.y;
            use[t] = i + t < ncol && qd.x != r.base_ref;
        }
        scan_quad(ix, ref, len, use, r.base_ref, b, alive);
    }
    r.alive = alive;
    r.count = pa_popc32(alive);
}

// Whole intersection by one lane (host emulator; tiers 1 and 2 by per-lane binary search)
PA_HD Isect isect_count(const Lane& s, const DevIndexView& ix, ColRef cols) {
    Isect r;
    const uint32_t tier = isect_pick(s, cols, r);
    if (tier == 0) {
        isect_light(s, ix, cols, r);
        return r;
    }
    const uint32_t ncol = l_ncol(s);
    if (ncol > 3) r.base_colour = class_colour(ix, r.base_ref);
    if (tier == 1) {
        isect_scan(s, ix, cols, r);
        return r;
    }
    const uint32_t* bids = class_ids(ix, r.base_ref);
    if (r.base_len <= 64) {
        uint64_t alive = r.base_len == 64 ? ~0ull : ((1ull << r.base_len) - 1);
        for (uint32_t i = 0; i < ncol && alive; ++i) {
            uint32_t ref, len;
            get_class(cols, i, ref, len);
            if (ref == r.base_ref) continue;
            const uint32_t* ids = class_ids(ix, ref);
            for (uint64_t t = alive; t; t &= t - 1) {
                const uint32_t j = pa_ctz64(t);
                if (!list_contains(ids, len, bids[j])) alive &= ~(1ull << j);
            }
        }
        r.alive = alive;
        r.count = pa_popc64(alive);
    } else {
        for (uint32_t j = 0; j < r.base_len; ++j) r.count += in_all_lists(ix, cols, ncol, r.base_ref, bids[j]) ? 1u : 0u;
    }
    return r;
}

PA_HD void isect_write(const Lane& s, const DevIndexView& ix, ColRef cols, const Isect& r, uint32_t* dst) {
    if (r.in_regs) {                                                // survivors straight from registers, no loads
        const uint32_t alive = (uint32_t)r.alive;
#pragma unroll
        for (int j = 0; j < 7; ++j)
            if ((alive >> j) & 1u) dst[pa_popc32(alive & ((1u << j) - 1))] = r.ids[j];
        return;
    }
    const uint32_t* bids = class_ids(ix, r.base_ref);
    if (r.base_len <= 64) {
        for (uint64_t t = r.alive; t; t &= t - 1) *dst++ = bids[pa_ctz64(t)];
    } else {
        const uint32_t ncol = l_ncol(s);
        for (uint32_t j = 0; j < r.base_len; ++j) {
            const uint32_t v = bids[j];
            if (in_all_lists(ix, cols, ncol, r.base_ref, v)) *dst++ = v;
        }
    }
}
#endif   // PA_LS_LANE

