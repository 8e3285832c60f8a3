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

namespace pa {

DevIndexView FlatDevice::host_view() const {
    DevIndexView v;
    v.table = table.data();
    v.nbuckets = nbuckets;
    v.blobs = blobs.data();
    v.ledge = ledge.data();
    v.nid_of_handle = nid_of_handle.data();
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
    static constexpr double LOAD = 0.5;
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

    // ---- classes: 16-byte aligned records {class id, id0, id1, ...} ----
    if (f.num_transcripts >= 0xFFFFFFFFu) return fail(PA_ERR_UNSUPPORTED, "too many transcripts");   // 0xFFFFFFFF pads the records
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
    }
    out.ec.resize(out.ec.size() + 8, 0xFFFFFFFFu);   // tail pad: records are read two 16-byte words at a time
    // class windows: ids in [cmin, cmin + 32) U [cmin2, cmin2 + 32), cmin2 = first id beyond window 1; cmask = 0 if it does not fit
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

    // ---- blob placement: every blob starts on a 128-byte block (handles stay in 64-byte units) ----
    out.handle.resize(N);
    uint64_t cursor = 0, nk = 0;
    for (uint32_t i = 0; i < N; ++i) {
        if (f.node_len[i] < k) return fail(PA_ERR_FORMAT, "node %u shorter than k", i);
        if (f.node_len[i] >= (1u << 24)) return fail(PA_ERR_UNSUPPORTED, "node %u longer than 2^24 bases", i);
        if (f.node_colour[i] >= f.num_classes) return fail(PA_ERR_FORMAT, "node %u: colour out of range", i);
        const uint64_t size = (BLOB_HDR_BYTES + 8ull * ((f.node_len[i] + 31) / 32) + BLOB_ALIGN - 1) / BLOB_ALIGN * BLOB_ALIGN;
        if (cursor / BLOB_GRANULE >= NO_HANDLE - 1) return fail(PA_ERR_UNSUPPORTED, "graph exceeds the 256 GiB blob address space");
        // bit 0 (blobs start on 128-byte blocks: always clear in the address) = the header's third vector is needed: the class has a
        // second window, or no windows at all
        const U4 cwi = cwin[f.node_colour[i]];
        out.handle[i] = (uint32_t)(cursor / BLOB_GRANULE) | ((cwi.y == 0 || cwi.w != 0) ? HANDLE_WIDE : 0u);
        cursor += size;
        nk += f.node_len[i] - k + 1;
    }
    out.num_kmers = nk;
    out.blobs.assign(cursor + 64, 0);   // tail pad: fwd_step reads up to 5 words past a node's last word

    // ---- dictionary: every k-mer of every node -> (handle, offset); the kernel follows at most 15 overflow buckets, so
    // the table is rebuilt larger in the (practically impossible) case that some key sits further from its home ----
    // the node sequence is read up to two words past its end: work on a padded copy
    std::vector<uint64_t> seq_pad(f.node_seq, f.node_seq + (f.seq_bases + 31) / 32);
    seq_pad.resize(seq_pad.size() + 3, 0);
    const uint64_t* node_seq = seq_pad.data();
    auto node_kmers = [&](uint32_t i, auto&& fn) {
        const uint64_t s = f.node_start[i];
        const uint32_t n = f.node_len[i] - k + 1;
        KT km = KmerOps<KT>::get(node_seq, s, k);
        for (uint32_t o = 0; o < n; ++o) {
            if (o) km = (km >> 2) | ((KT)get_base(node_seq, s + o + k - 1) << topshift);
            fn(km, o);
        }
    };
    Dict<KT> dict{nullptr, 0};
    if (device_dict) {   // the GPU fills the dictionary and derives the edges (index_fill.hip)
        out.node_kcum.resize((size_t)N + 1);
        out.node_kcum[0] = 0;
        for (uint32_t i = 0; i < N; ++i) out.node_kcum[i + 1] = out.node_kcum[i] + (f.node_len[i] - k + 1);
        out.have_redge = f.node_redge != nullptr;
        out.have_ledge = f.node_ledge != nullptr;
    }
    for (double load = Dict<KT>::LOAD; !device_dict; load *= 0.75) {
        out.nbuckets = std::max<uint64_t>(1, (uint64_t)((double)nk / (Dict<KT>::SLOTS * load)) + 1);
        if (out.nbuckets >= 0xFFFFFFFFull) return fail(PA_ERR_UNSUPPORTED, "dictionary exceeds 2^32 buckets");
        out.table.assign(out.nbuckets * BUCKET_WORDS, 0xFFFFFFFFu);   // empty slots, no flags
        dict = Dict<KT>{out.table.data(), out.nbuckets};
        par_ranges(threads, N, [&](uint64_t a, uint64_t b, int) {
            for (uint64_t i = a; i < b; ++i) node_kmers((uint32_t)i, [&](KT km, uint32_t o) { dict.insert_mt(km, out.handle[i], o); });
        });
        par_ranges(threads, N, [&](uint64_t a, uint64_t b, int) {   // k <= 32: keys that did not get their home slot (dict_slots.hpp)
            for (uint64_t i = a; i < b; ++i) node_kmers((uint32_t)i, [&](KT km, uint32_t o) { dict.insert_rest_mt(km, out.handle[i], o); });
        });
        std::atomic<uint32_t> bad{NO_HANDLE}, far{0};
        par_ranges(threads, N, [&](uint64_t a, uint64_t b, int) {
            for (uint64_t i = a; i < b; ++i)
                node_kmers((uint32_t)i, [&](KT km, uint32_t o) {
                    uint32_t h, off, probes = 0;
                    if (!dict.find(km, h, off, &probes) || h != out.handle[i] || off != o) bad.store((uint32_t)i);
                    if (probes > DICT_MAX_PROBES) far.store(1);
                });
        });
        if (bad.load() != NO_HANDLE) return fail(PA_ERR_FORMAT, "a k-mer of node %u occurs twice in the graph", bad.load());
        if (!far.load()) break;
    }

    // ---- blobs + edges ----
    const uint64_t granules = out.blobs.size() / BLOB_GRANULE;
    out.ledge.assign(8ull * granules + 8, NO_HANDLE);   // {handle, length}[4] per granule (lengths filled in below)
    out.nid_of_handle.assign(granules + 1, 0xFFFFFFFFu);
    std::atomic<uint32_t> dangling{NO_HANDLE};
    par_ranges(threads, N, [&](uint64_t a, uint64_t b, int) {
        for (uint64_t i = a; i < b; ++i) {
            uint8_t* blob = out.blobs.data() + (uint64_t)(out.handle[i] & ~HANDLE_WIDE) * BLOB_GRANULE;
            uint32_t* hd = reinterpret_cast<uint32_t*>(blob);
            uint64_t* sq = reinterpret_cast<uint64_t*>(blob + BLOB_HDR_BYTES);
            const uint32_t len = f.node_len[i];
            const uint64_t s = f.node_start[i];
            hd[0] = len | ((uint32_t)f.node_exts[i] << 24);
            hd[1] = f.node_colour[i];
            out.nid_of_handle[out.handle[i]] = (uint32_t)i;
            const U4 cw = cwin[f.node_colour[i]];
            hd[2] = cw.x; hd[3] = cw.y;
            hd[8] = cw.z; hd[9] = cw.w;
            hd[10] = out.class_ref[f.node_colour[i]];
            hd[11] = out.class_len[f.node_colour[i]];
            for (uint32_t w = 0; w < (len + 31) / 32; ++w) {
                uint64_t v = window32(node_seq, s + 32ull * w);
                const uint32_t rem = len - 32 * w;
                if (rem < 32) v &= (1ull << (2 * rem)) - 1;
                sq[w] = v;
            }
            const KT first = KmerOps<KT>::get(node_seq, s, k), last = KmerOps<KT>::get(node_seq, s + len - k, k);
            for (uint32_t base = 0; base < 4; ++base) {
                uint32_t re = NO_HANDLE, le = NO_HANDLE;
                if (f.node_exts[i] & (1u << base)) {
                    if (f.node_redge) {
                        const uint32_t t = f.node_redge[4 * i + base];
                        if (t < N) re = out.handle[t];
                    } else if (!device_dict) {
                        // find_link(last.extend_right(b), Dir::Right): node whose FIRST k-mer it is (offset 0)
                        uint32_t h, off;
                        if (dict.find((last >> 2) | ((KT)base << topshift), h, off) && off == 0) re = h;
                    }
                    if (re == NO_HANDLE && (f.node_redge || !device_dict)) dangling.store((uint32_t)i);
                }
                if (f.node_exts[i] & (1u << (4 + base))) {
                    if (f.node_ledge) {
                        const uint32_t t = f.node_ledge[4 * i + base];
                        if (t < N) le = out.handle[t];
                    } else if (!device_dict) {
                        // find_link(first.extend_left(b), Dir::Left): node whose LAST k-mer it is
                        // (that it IS the last k-mer is verified below, once every header has been written)
                        uint32_t h, off;
                        if (dict.find(((first << 2) | base) & mask, h, off)) le = h;
                    }
                    if (le == NO_HANDLE && (f.node_ledge || !device_dict)) dangling.store((uint32_t)i);
                }
                hd[4 + base] = re;
                out.ledge[8ull * out.handle[i] + 2 * base] = le;
            }
        }
    });
    if (dangling.load() != NO_HANDLE)
        return fail(PA_ERR_FORMAT, "node %u has an extension bit without a terminal neighbour k-mer (missing link)", dangling.load());
    // every left edge carries the length of the node it leads to (every header exists now)
    par_ranges(threads, N, [&](uint64_t a, uint64_t b, int) {
        for (uint64_t i = a; i < b; ++i)
            for (uint32_t base = 0; base < 4; ++base) {
                uint32_t* e = out.ledge.data() + 8ull * out.handle[i] + 2 * base;
                e[1] = e[0] == NO_HANDLE ? 0u : (*reinterpret_cast<const uint32_t*>(out.blobs.data() + (uint64_t)(e[0] & ~HANDLE_WIDE) * BLOB_GRANULE) & 0xFFFFFFu);
            }
    });
    // left-edge targets must be entered at their LAST k-mer (offset len-k): check now that every header exists
    if (!f.node_ledge && !device_dict) {
        par_ranges(threads, N, [&](uint64_t a, uint64_t b, int) {
            for (uint64_t i = a; i < b; ++i) {
                const KT first = KmerOps<KT>::get(node_seq, f.node_start[i], k);
                for (uint32_t base = 0; base < 4; ++base) {
                    if (!(f.node_exts[i] & (1u << (4 + base)))) continue;
                    uint32_t h = 0, off = 0;
                    dict.find(((first << 2) | base) & mask, h, off);
                    const uint32_t tlen = *reinterpret_cast<const uint32_t*>(out.blobs.data() + (uint64_t)(h & ~HANDLE_WIDE) * BLOB_GRANULE) & 0xFFFFFFu;
                    if (off != tlen - k) dangling.store((uint32_t)i);
                }
            }
        });
        if (dangling.load() != NO_HANDLE)
            return fail(PA_ERR_FORMAT, "node %u: left neighbour k-mer is not the last k-mer of its node", dangling.load());
    }
    return PA_OK;
}

int flatten_for_device(const pa_flat_index& f, int threads, FlatDevice& out, bool device_dict) {
    if (f.k < PA_MIN_K || f.k > PA_MAX_K) return fail(PA_ERR_UNSUPPORTED, "k=%u outside [%u,%u]", f.k, PA_MIN_K, PA_MAX_K);
    return f.k <= 32 ? flatten_t<uint64_t>(f, threads, out, device_dict) : flatten_t<u128>(f, threads, out, device_dict);
}

}  // namespace pa
