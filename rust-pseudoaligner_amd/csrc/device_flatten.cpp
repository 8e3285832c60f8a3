// pa_flat_index -> GPU layout (device_layout.hpp). Pure host code; runs once per pa_index_create.
//
// Replaces make_dbg_index (src/build_index.rs:182-221: boomphf MPHF + (node, offset) scatter) by an exact bucketed
// dictionary, and Node::r_edges()/l_edges() (debruijn crate: hash lookups at every hop) by edge handles stored in
// the node header.
#include "device_flatten.hpp"

#include <algorithm>
#include <atomic>
#include <thread>

#include "dict_slots.hpp"
#include "lane_steps.hpp"
#include "pa_common.hpp"

#ifndef PA_BRANCH_TAILS   // A/B builds: -DPA_BRANCH_TAILS=0 (tails only behind nodes with ONE right extension, as in round 4)
#define PA_BRANCH_TAILS 1
#endif

namespace pa {

DevIndexView FlatDevice::host_view() const {
    DevIndexView v;
    v.table = table.data();
    v.nbuckets = nbuckets;
    v.blobs = blobs.data();
    v.ledge = ledge.data();
    v.seg_g = seg_g.data();
    v.seg_nid = seg_nid.data();
    v.ec = ec.data();
    v.class_ref = class_ref.data();
    v.class_len = class_len.data();
    v.wtable = wtable.data();
    v.wbuckets = wbuckets;
    v.kmask = kmer_mask(k);
    v.kmask_hi = k > 32 ? kmer_mask(k - 32) : 0;
    v.k = k;
    v.num_nodes = num_nodes;
    v.num_classes = num_classes;
    v.num_segs = (uint32_t)seg_g.size();
    v.stream_nt = 0;
    v.bitmap_min = bitmap_min;
    v.bitmap_words = bitmap_words;
    return v;
}

namespace {

template <class F>
void par_ranges(int threads, uint64_t n, F f) {   // static contiguous ranges
    if (threads <= 1 || n < 4096) { f(0, n, 0); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t) th.emplace_back([=] { f(n * t / threads, n * (t + 1) / threads, t); });
    for (auto& x : th) x.join();
}

template <class KT> struct Dict;

template <>
struct Dict<u128> {   // k > 32: two whole entries {key word 0..3, handle, off, -, -} per line
    static constexpr double LOAD = 1.0 / 3.0;
    static constexpr uint32_t SLOTS = 2;
    uint32_t* words;
    uint64_t nbuckets;
    uint64_t bucket_of(u128 kmer) const { return ((pa_mix128((uint64_t)kmer, (uint64_t)(kmer >> 64)) >> 32) * (uint64_t)(uint32_t)nbuckets) >> 32; }
    void insert_mt(u128 kmer, uint32_t handle, uint32_t off) {
        uint64_t b = bucket_of(kmer);
        for (;;) {
            uint32_t* line = words + b * BUCKET_WORDS;
            for (uint32_t i = 0; i < SLOTS; ++i) {
                uint32_t expect = NO_HANDLE;
                if (__atomic_compare_exchange_n(&line[8 * i + 4], &expect, handle, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
                    line[8 * i] = (uint32_t)kmer; line[8 * i + 1] = (uint32_t)(kmer >> 32);
                    line[8 * i + 2] = (uint32_t)(kmer >> 64); line[8 * i + 3] = (uint32_t)(kmer >> 96);
                    line[8 * i + 5] = off;
                    return;
                }
            }
            if (++b == nbuckets) b = 0;
        }
    }
    void insert_rest_mt(u128, uint32_t, uint32_t) {}   // (one pass)
    bool find(u128 kmer, uint32_t& handle, uint32_t& off, uint32_t* probes_out = nullptr) const {
        uint64_t b = bucket_of(kmer);
        for (uint64_t probes = 0; probes < nbuckets; ++probes) {
            const uint32_t* line = words + b * BUCKET_WORDS;
            bool full = true;
            for (uint32_t i = 0; i < SLOTS; ++i) {
                if (line[8 * i + 4] == NO_HANDLE) { full = false; continue; }
                if (line[8 * i] == (uint32_t)kmer && line[8 * i + 1] == (uint32_t)(kmer >> 32) && line[8 * i + 2] == (uint32_t)(kmer >> 64) &&
                    line[8 * i + 3] == (uint32_t)(kmer >> 96)) {
                    handle = line[8 * i + 4];
                    off = line[8 * i + 5];
                    if (probes_out) *probes_out = (uint32_t)probes;
                    return true;
                }
            }
            if (!full) return false;
            if (++b == nbuckets) b = 0;
        }
        return false;
    }
};

struct HostAtomics {
    static bool cas(uint32_t* p, uint32_t expect, uint32_t v) { return __atomic_compare_exchange_n(p, &expect, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED); }
    static void and_(uint32_t* p, uint32_t m) { __atomic_fetch_and(p, m, __ATOMIC_RELAXED); }
};

template <>
struct Dict<uint64_t> {   // host-side builder/reader of the 16-byte slots described in device_layout.hpp (dict_slots.hpp)
    static constexpr double LOAD = DICT_LOAD;
    static constexpr uint32_t SLOTS = SLOTS_PER_BUCKET;
    uint32_t* words;
    uint64_t nbuckets;
    void insert_mt(uint64_t kmer, uint32_t handle, uint32_t off) { dict_insert_home<HostAtomics>(words, (uint32_t)nbuckets, kmer, handle, off); }          // pass 1
    void insert_rest_mt(uint64_t kmer, uint32_t handle, uint32_t off) { dict_insert_rest<HostAtomics>(words, (uint32_t)nbuckets, kmer, handle, off); }     // pass 2
    bool find(uint64_t kmer, uint32_t& handle, uint32_t& off, uint32_t* probes_out = nullptr) const {
        uint32_t probes = 0;
        const bool found = dict_find64(words, (uint32_t)nbuckets, kmer, handle, off, probes);
        if (probes_out) *probes_out = probes;
        return found;
    }
};

}  // namespace

// first / last k-mer of a node -> node id (open addressing, built once by one thread, read by many)
template <class KT>
struct KmerMap {
    std::vector<KT> keys;
    std::vector<uint32_t> vals;
    uint64_t mask;
    explicit KmerMap(uint32_t n) {
        uint64_t cap = 16;
        while (cap < 2ull * n + 2) cap <<= 1;
        keys.assign(cap, 0);
        vals.assign(cap, NO_HANDLE);
        mask = cap - 1;
    }
    void insert(KT km, uint32_t v) {
        for (uint64_t j = KmerOps<KT>::hash(km) & mask;; j = (j + 1) & mask) {
            if (vals[j] == NO_HANDLE) { keys[j] = km; vals[j] = v; return; }
            if (keys[j] == km) return;   // (a k-mer that occurs twice is reported by the dictionary's self-check)
        }
    }
    uint32_t find(KT km) const {
        for (uint64_t j = KmerOps<KT>::hash(km) & mask;; j = (j + 1) & mask) {
            if (vals[j] == NO_HANDLE) return NO_HANDLE;
            if (keys[j] == km) return vals[j];
        }
    }
};

template <class KT>
static int flatten_t(const pa_flat_index& f, int threads, FlatDevice& out, bool device_dict) {
    if (threads < 1) threads = 1;
    const uint32_t k = f.k, N = f.num_nodes;
    out = FlatDevice();
    out.k = k;
    out.num_nodes = N;
    out.num_classes = f.num_classes;
    const KT mask = KmerOps<KT>::mask(k);
    const uint32_t topshift = 2 * (k - 1);

    // ---- classes: their windows first (ids in [cmin, cmin + 32) U [cmin2, cmin2 + 32), cmin2 = first id beyond window 1; cmask = 0 if it does not fit) ----
    if (f.num_transcripts >= 0xFFFFFFFFu) return fail(PA_ERR_UNSUPPORTED, "too many transcripts");   // 0xFFFFFFFF pads the records
    std::vector<U4> cwin(f.num_classes, U4{0, 0, 0, 0});
    uint64_t nwin = 0;
    for (uint32_t c = 0; c < f.num_classes; ++c) {
        const uint32_t* ids = f.ec_ids + f.ec_offset[c];
        const uint64_t len = f.ec_offset[c + 1] - f.ec_offset[c];
        if (len == 0) continue;
        U4 w{ids[0], 0, 0, 0};
        uint64_t j = 0;
        for (; j < len && ids[j] - w.x < CLASS_WINDOW; ++j) w.y |= 1u << (ids[j] - w.x);
        if (j < len) {
            w.z = ids[j];
            for (; j < len && ids[j] - w.z < CLASS_WINDOW; ++j) w.w |= 1u << (ids[j] - w.z);
        }
        if (j < len) continue;   // three or more windows: list mode only
        cwin[c] = w;
        ++nwin;
    }
    // membership bitmaps for the window-less classes of at least bitmap_min ids (device_layout.hpp, class_bitmap): 16 ids and up (a shorter
    // list is one round of four 16-byte loads anyway) unless that would take more than 4 GB — then only the longer ones, or none
    uint32_t id_span = f.num_transcripts;   // (an exporter that does not know tx_names.len() may pass less than the ids say: the largest id decides)
    for (uint64_t j = 0; j < f.ec_offset[f.num_classes]; ++j) id_span = std::max(id_span, f.ec_ids[j] + 1u);
    out.bitmap_words = ((id_span + 31) / 32 + 2 + 3) & ~3u;
    out.bitmap_min = 16;
    for (;;) {
        uint64_t n = 0;
        for (uint32_t c = 0; c < f.num_classes; ++c)
            n += cwin[c].y == 0 && f.ec_offset[c + 1] - f.ec_offset[c] >= out.bitmap_min;
        out.num_bitmaps = n;
        if (n * out.bitmap_words * 4ull <= (4ull << 30)) break;
        if (out.bitmap_min >= (1u << 24)) { out.bitmap_min = 0; out.num_bitmaps = 0; break; }
        out.bitmap_min *= 2;
    }
    // ---- 16-byte aligned records {class id, id0, id1, ...} ----
    out.class_ref.resize(f.num_classes);
    out.class_len.resize(f.num_classes);
    for (uint32_t c = 0; c < f.num_classes; ++c) {
        const uint64_t len = f.ec_offset[c + 1] - f.ec_offset[c];
        if (len >= 0xFFFFFFFFull || out.ec.size() / 4 >= 0xFFFFFFFFull) return fail(PA_ERR_UNSUPPORTED, "class id lists too large");
        out.class_ref[c] = (uint32_t)(out.ec.size() / 4);
        out.class_len[c] = (uint32_t)len;
        out.max_class_len = std::max(out.max_class_len, (uint32_t)len);
        out.ec.push_back(c);
        out.ec.insert(out.ec.end(), f.ec_ids + f.ec_offset[c], f.ec_ids + f.ec_offset[c + 1]);
        while (out.ec.size() % 4 || out.ec.size() - 4ull * out.class_ref[c] < 8) out.ec.push_back(0xFFFFFFFFu);   // >= 8 words, 0xFFFFFFFF padded
        if (out.bitmap_min && cwin[c].y == 0 && len >= out.bitmap_min) {
            const size_t at = out.ec.size();
            if (at != class_bitmap(out.class_ref[c], (uint32_t)len)) return fail(PA_ERR_INTERNAL, "class record of %llu ids is not %u chunks", (unsigned long long)len, class_record_chunks((uint32_t)len));
            out.ec.resize(at + out.bitmap_words, 0u);
            for (uint64_t j = f.ec_offset[c]; j < f.ec_offset[c + 1]; ++j) {
                out.ec[at + (f.ec_ids[j] >> 5)] |= 1u << (f.ec_ids[j] & 31u);
            }
        }
    }
    out.ec.resize(out.ec.size() + 8, 0xFFFFFFFFu);   // tail pad: records are read two 16-byte words at a time
    // window table: canonical windows -> class id
    out.wbuckets = (uint32_t)std::max<uint64_t>(1, (uint64_t)((double)nwin / (WT_ENTRIES * 0.5)) + 1);
    out.wtable.assign((size_t)out.wbuckets * 16, 0);
    for (uint64_t l = 0; l < out.wbuckets; ++l) out.wtable[l * 16 + 4] = out.wtable[l * 16 + 9] = out.wtable[l * 16 + 14] = NO_CLASS;
    for (uint32_t c = 0; c < f.num_classes; ++c) {
        const U4 w = cwin[c];
        if (w.y == 0) continue;
        uint64_t line = ((uint64_t)window_hash(w.x, w.y, w.z, w.w) * out.wbuckets) >> 32;
        for (bool done = false; !done;) {
            uint32_t* e = out.wtable.data() + line * 16;
            for (uint32_t t = 0; t < WT_ENTRIES && !done; ++t)
                if (e[5 * t + 4] == NO_CLASS) {
                    e[5 * t] = w.x; e[5 * t + 1] = w.y; e[5 * t + 2] = w.z; e[5 * t + 3] = w.w; e[5 * t + 4] = c;
                    done = true;
                }
            if (++line == out.wbuckets) line = 0;
        }
    }

    // ---- neighbours: the node whose FIRST k-mer is last(i).extend_right(b) / whose LAST k-mer is first(i).extend_left(b)
    // (Node::r_edges / l_edges of the debruijn crate resolve them by hashing at every hop; taken from the flat
    // index when it supplies them) ----
    std::vector<uint64_t> seq_pad(f.node_seq, f.node_seq + (f.seq_bases + 31) / 32);   // read up to three words past the end: a padded copy
    seq_pad.resize(seq_pad.size() + 3, 0);
    const uint64_t* node_seq = seq_pad.data();
    uint64_t nk = 0;
    for (uint32_t i = 0; i < N; ++i) {
        if (f.node_len[i] < k) return fail(PA_ERR_FORMAT, "node %u shorter than k", i);
        if (f.node_len[i] >= (1u << 24)) return fail(PA_ERR_UNSUPPORTED, "node %u longer than 2^24 bases", i);
        if (f.node_colour[i] >= f.num_classes) return fail(PA_ERR_FORMAT, "node %u: colour out of range", i);
        nk += f.node_len[i] - k + 1;
    }
    out.num_kmers = nk;
    std::vector<uint32_t> redge(4ull * N, NO_HANDLE), ledge_n(4ull * N, NO_HANDLE);   // node ids
    {
        KmerMap<KT> first_of(N), last_of(N);
        if (!f.node_redge || !f.node_ledge)
            for (uint32_t i = 0; i < N; ++i) {
                first_of.insert(KmerOps<KT>::get(node_seq, f.node_start[i], k), i);
                last_of.insert(KmerOps<KT>::get(node_seq, f.node_start[i] + f.node_len[i] - k, k), i);
            }
        std::atomic<uint32_t> dangling{NO_HANDLE};
        par_ranges(threads, N, [&](uint64_t a, uint64_t b, int) {
            for (uint64_t i = a; i < b; ++i) {
                const uint64_t s = f.node_start[i];
                const KT first = KmerOps<KT>::get(node_seq, s, k), last = KmerOps<KT>::get(node_seq, s + f.node_len[i] - k, k);
                for (uint32_t base = 0; base < 4; ++base) {
                    if (f.node_exts[i] & (1u << base)) {
                        const uint32_t t = f.node_redge ? f.node_redge[4 * i + base] : first_of.find((last >> 2) | ((KT)base << topshift));
                        if (t >= N) dangling.store((uint32_t)i);
                        else redge[4 * i + base] = t;
                    }
                    if (f.node_exts[i] & (1u << (4 + base))) {
                        const uint32_t t = f.node_ledge ? f.node_ledge[4 * i + base] : last_of.find(((first << 2) | (KT)base) & mask);
                        if (t >= N) dangling.store((uint32_t)i);
                        else ledge_n[4 * i + base] = t;
                    }
                }
            }
        });
        if (dangling.load() != NO_HANDLE)
            return fail(PA_ERR_FORMAT, "node %u has an extension bit without a terminal neighbour k-mer (missing link)", dangling.load());
    }

    // ---- chains (device_layout.hpp): B follows A in a chain when A's only right extension leads to B, B's only left extension
    // to A, and every block the pair shares still fits its four slots ----
    auto is_wide = [&](uint32_t i) { const U4 cw = cwin[f.node_colour[i]]; return cw.y == 0 || cw.w != 0; };
    auto has_redge = [&](uint32_t i) { return (f.node_exts[i] & 15u) != 0; };
    std::vector<uint32_t> succ(N, NO_HANDLE);
    std::vector<uint8_t> has_pred(N, 0);
    for (uint32_t a = 0; a < N; ++a) {
        const uint32_t re = f.node_exts[a] & 15u;
        if (re == 0 || (re & (re - 1))) continue;
        const uint32_t bnode = redge[4 * a + (uint32_t)__builtin_ctz(re)];
        const uint32_t le = (f.node_exts[bnode] >> 4) & 15u;
        if (bnode == a || le == 0 || (le & (le - 1)) || ledge_n[4 * bnode + (uint32_t)__builtin_ctz(le)] != a) continue;
        succ[a] = bnode;
        has_pred[bnode] = 1;
    }
    struct SegI {
        uint32_t node, s, e;   // e: where the node ends in the chain — or, link set, where the chain's copy of it ends
        bool link;             // a tail copy cut short: the walk goes on in the node's own chain
        bool branch = false;   // the node has SEVERAL right extensions and what follows it here is a copy of the FAVOURED one: its record
                               // is followed by its edge slot (the other extensions) although it is not the chain's last
    };
    std::vector<SegI> segs;             // every node, chain after chain
    segs.reserve(N);
    std::vector<uint32_t> chain_first;  // index into segs of every chain's first node (+ end sentinel)
    // slots block j of the chain v[c0..) needs (v's last entry is the chain's last record)
    auto block_slots = [&](const std::vector<SegI>& v, size_t c0, uint32_t j) {
        const uint64_t base = (uint64_t)j << CH_STRIDE_LOG2;
        uint32_t cnt = 0;
        for (size_t t = v.size(); t-- > c0;) {
            const SegI& g = v[t];
            if (g.e <= base) break;
            if (g.s >= base + CH_WINDOW) continue;
            cnt += 1 + (is_wide(g.node) ? 1u : 0u);
            if ((g.branch || (t + 1 == v.size() && (g.link || has_redge(g.node)))) && g.e - base < CH_WINDOW) ++cnt;   // the edge / link slot, where a step can reach the node's end
        }
        return cnt;
    };
    // may `g` be appended to the chain v[c0..)? (every block its bases lie in, or the record before it can be seen from, must fit)
    auto fits_after = [&](std::vector<SegI>& v, size_t c0, const SegI& g) {
        const uint32_t prev_e = v.back().e;
        v.push_back(g);
        const uint32_t jlo = g.s < CH_WINDOW ? 0u : ((g.s - CH_WINDOW) >> CH_STRIDE_LOG2) + 1;
        const uint32_t jhi = std::min<uint32_t>((g.e - 1) >> CH_STRIDE_LOG2, (prev_e >> CH_STRIDE_LOG2) + 1);
        bool ok = true;
        for (uint32_t j = jlo; j <= jhi && ok; ++j) ok = block_slots(v, c0, j) <= CH_SLOTS;
        if (!ok) v.pop_back();
        return ok;
    };
    out.handle.assign(N, 0);
    out.node_s.assign(N, 0);
    std::vector<uint8_t> visited(N, 0);
    auto run = [&](uint32_t start) {
        uint32_t i = start;
        size_t c0 = segs.size();
        chain_first.push_back((uint32_t)c0);
        segs.push_back(SegI{i, 0u, f.node_len[i], false});
        for (;;) {
            visited[i] = 1;
            const uint32_t bn = succ[i];
            if (bn == NO_HANDLE || visited[bn]) break;
            const uint64_t sb = (uint64_t)segs.back().e - (k - 1), eb = sb + f.node_len[bn];
            if (!(eb < (1ull << 31) && fits_after(segs, c0, SegI{bn, (uint32_t)sb, (uint32_t)eb, false}))) {   // bn starts a chain of its own
                c0 = segs.size();
                chain_first.push_back((uint32_t)c0);
                segs.push_back(SegI{bn, 0u, f.node_len[bn], false});
            }
            i = bn;
        }
    };
    for (uint32_t i = 0; i < N; ++i)
        if (!has_pred[i] && !visited[i]) run(i);
    for (uint32_t i = 0; i < N; ++i)
        if (!visited[i]) run(i);   // closed loops of nodes cut only by colour
    const uint32_t nchains = (uint32_t)chain_first.size();
    chain_first.push_back((uint32_t)segs.size());
    out.num_chains = nchains;
    for (uint32_t c = 0; c < nchains; ++c)
        for (uint32_t t = chain_first[c]; t < chain_first[c + 1]; ++t) {
            out.handle[segs[t].node] = c;   // (the chain's number for now)
            out.node_s[segs[t].node] = segs[t].s;
        }

    // transcripts two classes share (sorted id lists)
    auto shared_ids = [&](uint32_t ca, uint32_t cb) {
        const uint32_t *a = f.ec_ids + f.ec_offset[ca], *ae = f.ec_ids + f.ec_offset[ca + 1], *b = f.ec_ids + f.ec_offset[cb], *be = f.ec_ids + f.ec_offset[cb + 1];
        uint64_t n = 0;
        while (a < ae && b < be) {
            if (*a < *b) ++a; else if (*b < *a) ++b; else { ++n; ++a; ++b; }
        }
        return n;
    };
    // ---- tails (device_layout.hpp): after a chain's last node Z, COPIES of the nodes a read can only go on to — Z's one
    // right extension, that node's one right extension, ... — for up to CH_TAIL bases, so that the step that runs over Z's end
    // finds them in the block it already holds. A copy that is cut short ends in a link to the same base of the node's own chain
    {
        std::vector<SegI> fin;
        std::vector<uint32_t> fin_first;
        fin.reserve(segs.size() + nchains);
        for (uint32_t c = 0; c < nchains; ++c) {
            const size_t c0 = fin.size();
            fin_first.push_back((uint32_t)c0);
            fin.insert(fin.end(), segs.begin() + chain_first[c], segs.begin() + chain_first[c + 1]);
            const uint64_t limit = (uint64_t)fin.back().e + CH_TAIL;
            for (uint32_t cur = fin.back().node; limit < (1ull << 31);) {
                const uint32_t re = f.node_exts[cur] & 15u;
                if (re == 0) break;
                // several right extensions: the copy is of the FAVOURED one — the successor most transcripts of this node go on to
                // (ties: the smaller base) — and the node's record keeps its edge slot for the others (SegI::branch)
                const bool branch = (re & (re - 1)) != 0;
                uint32_t bn = NO_HANDLE;
                if (!branch) bn = redge[4 * cur + (uint32_t)__builtin_ctz(re)];
                else if (PA_BRANCH_TAILS) {
                    uint64_t best = 0;
                    for (uint32_t b = 0; b < 4; ++b) {
                        if (!(re & (1u << b))) continue;
                        const uint32_t t = redge[4 * cur + b];
                        const uint64_t c = 1 + shared_ids(f.node_colour[cur], f.node_colour[t]);
                        if (c > best) { best = c; bn = t; }
                    }
                }
                if (bn == NO_HANDLE) break;
                const uint64_t sb = (uint64_t)fin.back().e - (k - 1), full = sb + f.node_len[bn];
                if ((uint64_t)fin.back().e + 1 >= limit) break;          // no room for a base beyond the one the extension test looks at
                const bool cut = full > limit;
                fin.back().branch = branch;
                if (!fits_after(fin, c0, SegI{bn, (uint32_t)sb, (uint32_t)(cut ? limit : full), cut})) { fin.back().branch = false; break; }
                if (cut) break;
                cur = bn;
            }
            // a copy the tail ends with WHOLE hands the walk to its right edges, and those enter a node at its chain's first
            // k-mer: fine for a node with several right extensions (none of them was merged with it), not for a node whose one
            // successor follows it inside its own chain. Such a copy gives up its last base to a link — or, too short for that, goes
            for (const size_t c1 = c0 + (chain_first[c + 1] - chain_first[c]); fin.size() > c1 && !fin.back().link;) {
                const uint32_t re = f.node_exts[fin.back().node] & 15u;
                if (re == 0 || (re & (re - 1)) || out.node_s[redge[4 * fin.back().node + (uint32_t)__builtin_ctz(re)]] == 0) break;
                bool done = false;
                if (fin.back().e >= fin[fin.size() - 2].e + 2) {
                    SegI g = fin.back();
                    g.e -= 1;
                    g.link = true;
                    fin.pop_back();
                    done = fits_after(fin, c0, g);                   // (one base shorter: its end may now be in reach of one more block)
                    if (!done) { g.e += 1; g.link = false; fin.push_back(g); }
                }
                if (!done) { fin.pop_back(); fin.back().branch = false; }   // (what is the last record now has no copy behind it)
            }
        }
        fin_first.push_back((uint32_t)fin.size());
        segs.swap(fin);
        chain_first.swap(fin_first);
    }
    const size_t nsegs = segs.size();

    // ---- placement: chain handle = 128-byte blocks before it ----
    std::vector<uint32_t> chain_handle(nchains);
    out.seg_g.resize(nsegs);
    out.seg_nid.resize(nsegs);
    uint64_t cursor = 0;
    for (uint32_t c = 0; c < nchains; ++c) {
        if (cursor >= NO_HANDLE - 8) return fail(PA_ERR_UNSUPPORTED, "graph exceeds the 512 GiB chain address space");
        chain_handle[c] = (uint32_t)cursor;
        for (uint32_t t = chain_first[c]; t < chain_first[c + 1]; ++t) {
            out.seg_g[t] = ((uint64_t)cursor << CH_STRIDE_LOG2) + segs[t].s;
            out.seg_nid[t] = segs[t].node;
        }
        cursor += (segs[chain_first[c + 1] - 1].e + CH_STRIDE - 1) >> CH_STRIDE_LOG2;
    }
    for (uint32_t i = 0; i < N; ++i) out.handle[i] = chain_handle[out.handle[i]];   // chain number -> chain handle
    const uint64_t nblocks = cursor;
    out.blobs.assign(nblocks * CH_BLOCK + 64, 0);   // tail pad: a step's last sequence load may reach one word past its block

    // ---- blocks ----
    std::atomic<uint32_t> bad_edge{NO_HANDLE}, bad_slots{NO_HANDLE};
    par_ranges(threads, nchains, [&](uint64_t ca, uint64_t cb, int) {
        std::vector<uint64_t> cs;   // the chain's sequence
        for (uint64_t c = ca; c < cb; ++c) {
            const uint32_t t0 = chain_first[c], t1 = chain_first[c + 1], clen = segs[t1 - 1].e;
            cs.assign((clen + 31) / 32 + 9, 0);
            for (uint32_t t = t0; t < t1; ++t) {
                const SegI& g = segs[t];
                const uint64_t src = f.node_start[g.node];
                for (uint32_t o = t == t0 ? 0 : k - 1; g.s + o < g.e; ++o) set_base(cs.data(), g.s + o, get_base(node_seq, src + o));
            }
            const uint32_t nblk = (clen + CH_STRIDE - 1) >> CH_STRIDE_LOG2;
            uint32_t tlo = t0;   // first node that can still overlap the current block
            for (uint32_t j = 0; j < nblk; ++j) {
                const uint32_t base = j << CH_STRIDE_LOG2;
                uint8_t* blk = out.blobs.data() + ((uint64_t)chain_handle[c] + j) * CH_BLOCK;
                uint32_t* sl = reinterpret_cast<uint32_t*>(blk);
                uint64_t* sq = reinterpret_cast<uint64_t*>(blk + CH_SEQ_BYTES);
                while (tlo < t1 && segs[tlo].e <= base) ++tlo;
                uint32_t slot = 0, recmask = 0;
                for (uint32_t t = tlo; t < t1 && segs[t].s < base + CH_WINDOW; ++t) {
                    const SegI& g = segs[t];
                    const U4 cw = cwin[f.node_colour[g.node]];
                    const bool wide = is_wide(g.node), last = t + 1 == t1, reach = g.e - base < CH_WINDOW,
                               edges = ((last && !g.link && has_redge(g.node)) || g.branch) && reach, link = last && g.link && reach;
                    if (slot + 1 + (wide ? 1 : 0) + (edges || link ? 1 : 0) > CH_SLOTS) { bad_slots.store(g.node); break; }   // (the merge rule keeps every block within its slots)
                    recmask |= 1u << slot;
                    uint32_t* r = sl + 4 * slot++;
                    r[0] = std::min<uint32_t>(g.e - base, SEG_E_FAR) | (wide ? SEG_WIDE : 0u) | (last ? SEG_LAST : 0u) | (edges ? SEG_EDGES : 0u) | (link ? SEG_LINK : 0u);
                    r[1] = f.node_colour[g.node]; r[2] = cw.x; r[3] = cw.y;
                    if (wide) {
                        uint32_t* x = sl + 4 * slot++;
                        x[0] = cw.z; x[1] = cw.w; x[2] = out.class_ref[f.node_colour[g.node]]; x[3] = out.class_len[f.node_colour[g.node]];
                    }
                    if (edges) {
                        uint32_t* x = sl + 4 * slot++;
                        for (uint32_t b = 0; b < 4; ++b) {
                            const uint32_t tn = redge[4 * g.node + b];
                            if (tn != NO_HANDLE && out.node_s[tn] != 0) bad_edge.store(g.node);   // a right edge enters a node at its first k-mer, which starts its chain
                            x[b] = tn == NO_HANDLE ? NO_HANDLE : out.handle[tn];
                        }
                    }
                    if (link) {   // the base after the copy's last one, in the node's own chain, as a continuing step addresses it (fwd_finish)
                        uint32_t* x = sl + 4 * slot++;
                        const uint32_t xt = out.node_s[g.node] + (g.e - g.s), adv = (xt - 1) >> CH_STRIDE_LOG2;
                        x[0] = out.handle[g.node] + adv;
                        x[1] = xt - (adv << CH_STRIDE_LOG2);
                        x[2] = x[3] = 0;
                    }
                }
                sl[0] |= (std::min(j, CH_BACK_MAX) << SEG_BACK_SHIFT) | (recmask << SEG_RECMASK_SHIFT);
                for (uint32_t w = 0; w < CH_WINDOW / 32; ++w) sq[w] = window32(cs.data(), base + 32ull * w);   // (cs is zero beyond the chain)
            }
        }
    });
    if (bad_slots.load() != NO_HANDLE) return fail(PA_ERR_INTERNAL, "node %u: a chain block needs more than its four slots", bad_slots.load());
    // a link also names the SLOT of the node it leads to in the block it leads to (word 2, as the lane state holds it: lane_steps.hpp,
    // of_cur), so that the step behind it can load that block's slots rotated: a second pass, every block is in place now
    par_ranges(threads, nblocks, [&](uint64_t ba, uint64_t bb, int) {
        for (uint64_t b = ba; b < bb; ++b) {
            uint32_t* sl = reinterpret_cast<uint32_t*>(out.blobs.data() + b * CH_BLOCK);
            const uint32_t recmask = sl[0] >> SEG_RECMASK_SHIFT;
            for (uint32_t t = 0; t < CH_SLOTS; ++t) {
                if (!((recmask >> t) & 1u) || !(sl[4 * t] & SEG_LINK)) continue;
                uint32_t* x = sl + 4 * (t + 1 + ((sl[4 * t] & SEG_WIDE) ? 1u : 0u));
                x[2] = of_cur(block_slot_of(out.blobs.data() + (uint64_t)x[0] * CH_BLOCK, x[1] - 1), true);
            }
        }
    });
    if (bad_edge.load() != NO_HANDLE) return fail(PA_ERR_FORMAT, "node %u: a right edge does not lead to the first k-mer of a chain", bad_edge.load());

    // ---- left edges by chain handle: where the extension goes on — the neighbour's last k-mer, in the block with the most
    // room to the left of it ----
    out.ledge.assign(8ull * nblocks + 8, NO_HANDLE);
    for (uint32_t c = 0; c < nchains; ++c) {
        const uint32_t a = segs[chain_first[c]].node;
        for (uint32_t b = 0; b < 4; ++b) {
            const uint32_t tn = ledge_n[4 * a + b];
            uint32_t* e = out.ledge.data() + 8ull * chain_handle[c] + 2 * b;
            e[1] = 0;
            if (tn == NO_HANDLE) continue;
            const uint32_t y = out.node_s[tn] + f.node_len[tn] - k;   // its last k-mer (prev_kmer_offset = len - k, :196)
            const uint32_t jy = y >> CH_STRIDE_LOG2, j = jy > CH_BACK_MAX ? jy - CH_BACK_MAX : 0;
            e[0] = out.handle[tn] + j;
            e[1] = y - (j << CH_STRIDE_LOG2) + 1;
        }
    }
    // (a left edge must enter its node at the node's LAST k-mer, which ends its chain: the neighbour tables above were built
    // from last k-mers, so a left extension that leads anywhere else was reported as a missing link)

    // ---- dictionary: every k-mer of every node -> (block, position); the kernel follows at most 15 overflow buckets, so
    // the table is rebuilt larger in the (practically impossible) case that some key sits further from its home ----
    auto node_kmers = [&](uint32_t i, auto&& fn) {
        const uint64_t s = f.node_start[i];
        const uint32_t n = f.node_len[i] - k + 1;
        KT km = KmerOps<KT>::get(node_seq, s, k);
        for (uint32_t o = 0; o < n; ++o) {
            if (o) km = (km >> 2) | ((KT)get_base(node_seq, s + o + k - 1) << topshift);
            const uint32_t c = out.node_s[i] + o, blk = out.handle[i] + (c >> CH_STRIDE_LOG2);
            fn(km, blk, dict_entry_off(c, o == 0, block_slot_of(out.blobs.data() + (uint64_t)blk * CH_BLOCK, (c & (CH_STRIDE - 1)) + k - 1)));
        }
    };
    Dict<KT> dict{nullptr, 0};
    if (device_dict) {   // the GPU fills the dictionary from the uploaded blocks (index_fill.hip)
        out.node_kcum.resize((size_t)N + 1);
        out.node_kcum[0] = 0;
        for (uint32_t i = 0; i < N; ++i) out.node_kcum[i + 1] = out.node_kcum[i] + (f.node_len[i] - k + 1);
    }
    double load0 = Dict<KT>::LOAD;
    if (const char* v = knob_str("PA_DICT_LOAD")) { const double x = atof(v); if (x > 0.01 && x <= 0.95) load0 = x; }   // (knobs builds and the test emulator: dense tables exercise the probe's rare paths)
    for (double load = load0; !device_dict; load *= 0.75) {
        out.nbuckets = std::max<uint64_t>(1, (uint64_t)((double)nk / (Dict<KT>::SLOTS * load)) + 1);
        if (out.nbuckets >= 0xFFFFFFFFull) return fail(PA_ERR_UNSUPPORTED, "dictionary exceeds 2^32 buckets");
        out.table.assign(out.nbuckets * BUCKET_WORDS, 0xFFFFFFFFu);   // empty slots, no flags
        dict = Dict<KT>{out.table.data(), out.nbuckets};
        par_ranges(threads, N, [&](uint64_t a, uint64_t b, int) {
            for (uint64_t i = a; i < b; ++i) node_kmers((uint32_t)i, [&](KT km, uint32_t h, uint32_t o) { dict.insert_mt(km, h, o); });
        });
        par_ranges(threads, N, [&](uint64_t a, uint64_t b, int) {   // k <= 32: keys that did not get their home slot (dict_slots.hpp)
            for (uint64_t i = a; i < b; ++i) node_kmers((uint32_t)i, [&](KT km, uint32_t h, uint32_t o) { dict.insert_rest_mt(km, h, o); });
        });
        std::atomic<uint32_t> bad{NO_HANDLE}, far{0};
        par_ranges(threads, N, [&](uint64_t a, uint64_t b, int) {
            for (uint64_t i = a; i < b; ++i)
                node_kmers((uint32_t)i, [&](KT km, uint32_t h, uint32_t o) {
                    uint32_t fh, fo, probes = 0;
                    if (!dict.find(km, fh, fo, &probes) || fh != h || fo != o) bad.store((uint32_t)i);
                    if (probes > DICT_MAX_PROBES) far.store(1);
                });
        });
        if (bad.load() != NO_HANDLE) return fail(PA_ERR_FORMAT, "a k-mer of node %u occurs twice in the graph", bad.load());
        if (!far.load()) break;
    }
    return PA_OK;
}

int flatten_for_device(const pa_flat_index& f, int threads, FlatDevice& out, bool device_dict) {
    if (f.k < PA_MIN_K || f.k > PA_MAX_K) return fail(PA_ERR_UNSUPPORTED, "k=%u outside [%u,%u]", f.k, PA_MIN_K, PA_MAX_K);
    return f.k <= 32 ? flatten_t<uint64_t>(f, threads, out, device_dict) : flatten_t<u128>(f, threads, out, device_dict);
}

}  // namespace pa
