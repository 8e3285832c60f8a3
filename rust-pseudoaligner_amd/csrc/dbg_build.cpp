// CPU index construction: transcripts -> stranded, coloured, compacted De Bruijn graph + equivalence classes.
//
// Semantics follow the reference's build_index (src/build_index.rs:27-91) without sharing its mechanics:
//   * every k-mer of every transcript, forward strand only (STRANDED = true, src/config.rs:14); transcripts
//     shorter than k contribute nothing (src/build_index.rs:134,148-150);
//   * k-mer colour = sorted, dedup'd list of transcript ids containing it, interned to an equivalence-class id
//     (CountFilterEqClass::summarize, src/equiv_classes.rs:62-91);
//   * k-mer extensions = union of the neighbouring bases observed inside transcripts
//     (Exts::from_dna_string per MSP slice, src/build_index.rs:144; all_exts.add, src/equiv_classes.rs:72-76);
//   * nodes = maximal paths whose consecutive k-mers are each other's unique extension AND carry the same colour
//     (ScmapCompress, src/build_index.rs:171,178).
// The reference reaches this through MSP sharding (:127-151), per-shard compression (:153-172) and a merge pass
// (:174-179); here it is one partition-sort-scan + one hash-table walk, multi-threaded with std::thread.
// Not pinned by any reference test: node/colour numbering (unobservable through map_read) and the break point of
// a pure cycle of joinable k-mers (no start k-mer exists; we break at the first k-mer in partition order).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <unordered_map>

#include "pa_common.hpp"

namespace pa {
namespace {

template <class KT>
struct Rec {   // trivially default-constructible: arrays of it are left uninitialised
    KT kmer;
    uint32_t tx;
    uint32_t exts;
};

constexpr uint32_t EMPTY = 0xFFFFFFFFu;

template <class KT>
struct KEntry {
    KT kmer;
    uint32_t colour;   // EMPTY = free slot
    uint8_t exts;
    uint8_t visited;
    uint16_t pad;
};
static_assert(sizeof(KEntry<uint64_t>) == 16, "KEntry");

template <class KT>
struct KTable {
    typedef struct KEntry<KT> KEntry;
    uint64_t cap = 0;
    std::unique_ptr<KEntry[]> e;
    void init(uint64_t n, int threads) {   // first touch in parallel: page faults dominate a serial fill
        cap = (uint64_t)((double)n / 0.55) + 64;
        e.reset(new KEntry[cap]);
        std::vector<std::thread> th;
        for (int t = 0; t < threads; ++t)
            th.emplace_back([this, t, threads] {
                for (uint64_t i = cap * t / threads; i < cap * (t + 1) / threads; ++i) e[i] = KEntry{0, EMPTY, 0, 0, 0};
            });
        for (auto& x : th) x.join();
    }
    uint64_t home(KT kmer) const { return (uint64_t)(((unsigned __int128)KmerOps<KT>::hash(kmer) * cap) >> 64); }
    void insert_mt(KT kmer, uint32_t colour, uint8_t exts) {
        uint64_t i = home(kmer);
        for (;;) {
            uint32_t expect = EMPTY;
            if (__atomic_compare_exchange_n(&e[i].colour, &expect, colour, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
                e[i].kmer = kmer;
                e[i].exts = exts;
                return;
            }
            if (++i == cap) i = 0;
        }
    }
    KEntry* find(KT kmer) {
        uint64_t i = home(kmer);
        for (;;) {
            KEntry& s = e[i];
            if (s.colour == EMPTY) return nullptr;
            if (s.kmer == kmer) return &s;
            if (++i == cap) i = 0;
        }
    }
};

// Sharded interning of transcript-id lists (the DashMap of src/equiv_classes.rs:17-19, 81-87).
struct Interner {
    static constexpr int SHARDS = 256;
    struct Shard {
        std::mutex mu;
        std::vector<uint32_t> arena;
        std::vector<uint64_t> off;   // list i = arena[off[i], off[i+1])
        std::unordered_multimap<uint64_t, uint32_t> map;
        Shard() { off.push_back(0); }
    };
    Shard shards[SHARDS];
    static uint64_t hash_list(const uint32_t* v, uint32_t n) {
        uint64_t h = 0x243f6a8885a308d3ull ^ n;
        for (uint32_t i = 0; i < n; ++i) h = mix64(h ^ v[i]) + 0x9e3779b97f4a7c15ull;
        return h;
    }
    // returns temp id = (local << 8) | shard
    uint32_t intern(const uint32_t* v, uint32_t n) {
        const uint64_t h = hash_list(v, n);
        Shard& s = shards[h & (SHARDS - 1)];
        std::lock_guard<std::mutex> g(s.mu);
        auto range = s.map.equal_range(h);
        for (auto it = range.first; it != range.second; ++it) {
            const uint32_t li = it->second;
            const uint64_t o = s.off[li];
            if (s.off[li + 1] - o == n && std::memcmp(&s.arena[o], v, n * sizeof(uint32_t)) == 0)
                return (li << 8) | (uint32_t)(h & (SHARDS - 1));
        }
        const uint32_t li = (uint32_t)(s.off.size() - 1);
        s.arena.insert(s.arena.end(), v, v + n);
        s.off.push_back(s.arena.size());
        s.map.emplace(h, li);
        return (li << 8) | (uint32_t)(h & (SHARDS - 1));
    }
};

template <class KT>
struct DK {   // distinct k-mer
    KT kmer;
    uint32_t colour;
    uint32_t exts;
};

template <class F>
void parallel_for(int threads, uint64_t n, F f) {   // dynamic scheduling over [0,n)
    std::atomic<uint64_t> next{0};
    auto worker = [&](int tid) {
        for (;;) {
            const uint64_t i = next.fetch_add(1);
            if (i >= n) break;
            f(i, tid);
        }
    };
    if (threads <= 1) { worker(0); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t) th.emplace_back(worker, t);
    for (auto& t : th) t.join();
}

struct NodeOut {   // nodes produced by one partition's start k-mers
    std::vector<uint64_t> seq;   // packed, private bit cursor
    uint64_t bases = 0;
    std::vector<uint32_t> len, colour;
    std::vector<uint8_t> exts;
    void push_base(uint32_t b) {
        if ((bases & 31) == 0) seq.push_back(0);
        seq.back() |= (uint64_t)b << ((bases & 31) * 2);
        ++bases;
    }
};

}  // namespace

template <class KT>
static int build_graph_t(const uint64_t* packed_in, const uint64_t* tx_start, uint32_t num_tx, uint32_t k, int threads, HostIndex& out) {
    typedef struct Rec<KT> Rec;
    typedef struct KEntry<KT> KEntry;
    typedef struct DK<KT> DK;
    if (threads < 1) threads = 1;
    const KT mask = KmerOps<KT>::mask(k);
    const uint32_t topshift = 2 * (k - 1);
    // k-mer extraction reads up to two words past the last base: work on a padded copy
    const uint64_t nwords = (tx_start[num_tx] + 31) / 32;
    std::vector<uint64_t> padded(packed_in, packed_in + nwords);
    padded.resize(nwords + 3, 0);
    const uint64_t* packed = padded.data();

    const bool verbose = std::getenv("PA_VERBOSE") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto stage = [&](const char* what) {
        if (!verbose) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[pa build] %-28s %.2f s\n", what, std::chrono::duration<double>(now - t_last).count());
        t_last = now;
    };
    // ---- 1. count k-mers, split transcripts over threads ----
    std::vector<uint64_t> kcum(num_tx + 1, 0);
    for (uint32_t t = 0; t < num_tx; ++t) {
        const uint64_t len = tx_start[t + 1] - tx_start[t];
        kcum[t + 1] = kcum[t] + (len >= k ? len - k + 1 : 0);
    }
    const uint64_t total = kcum[num_tx];
    out = HostIndex();
    out.k = k;
    out.num_transcripts = num_tx;
    out.node_start.push_back(0);
    out.ec_offset.push_back(0);
    if (total == 0) { out.node_seq.assign(2, 0); return PA_OK; }

    uint32_t logp = 4;
    while (logp < 12 && (total >> logp) > (1u << 20)) ++logp;
    const uint32_t P = 1u << logp;
    const int T = threads;
    std::vector<uint32_t> tx_split(T + 1, num_tx);
    tx_split[0] = 0;
    for (int t = 1; t < T; ++t) {
        const uint64_t want = total * t / T;
        tx_split[t] = (uint32_t)(std::lower_bound(kcum.begin(), kcum.end(), want) - kcum.begin());
        if (tx_split[t] > num_tx) tx_split[t] = num_tx;
    }
    for (int t = 1; t <= T; ++t) tx_split[t] = std::max(tx_split[t], tx_split[t - 1]);

    auto for_each_kmer = [&](uint32_t t0, uint32_t t1, auto&& fn) {
        for (uint32_t t = t0; t < t1; ++t) {
            const uint64_t s = tx_start[t], len = tx_start[t + 1] - s;
            if (len < k) continue;
            const uint64_t nk = len - k + 1;
            for (uint64_t p = 0; p < nk; ++p) {
                const KT km = KmerOps<KT>::get(packed, s + p, k);
                uint32_t ex = 0;
                if (p > 0) ex |= 1u << (4 + get_base(packed, s + p - 1));          // left ext
                if (p + k < len) ex |= 1u << get_base(packed, s + p + k);          // right ext
                fn(km, t, ex);
            }
        }
    };

    // ---- 2. partition records by hash prefix (count, prefix, scatter) ----
    std::vector<std::vector<uint64_t>> cnt(T, std::vector<uint64_t>(P, 0));
    parallel_for(T, T, [&](uint64_t ti, int) {
        auto& c = cnt[ti];
        for_each_kmer(tx_split[ti], tx_split[ti + 1], [&](KT km, uint32_t, uint32_t) { ++c[KmerOps<KT>::hash(km) >> (64 - logp)]; });
    });
    stage("  count pass");
    std::vector<uint64_t> pstart(P + 1, 0);
    {
        uint64_t acc = 0;
        for (uint32_t p = 0; p < P; ++p) {
            pstart[p] = acc;
            for (int t = 0; t < T; ++t) { const uint64_t c = cnt[t][p]; cnt[t][p] = acc; acc += c; }
        }
        pstart[P] = acc;
    }
    // uninitialised on purpose: zero-filling 16 B x (#k-mers) on one thread costs more than the whole scatter pass
    std::unique_ptr<Rec[]> recs(new (std::nothrow) Rec[total]);
    if (!recs) return fail(PA_ERR_OOM, "out of memory for %llu k-mer records", (unsigned long long)total);
    parallel_for(T, T, [&](uint64_t ti, int) {
        auto& c = cnt[ti];
        for_each_kmer(tx_split[ti], tx_split[ti + 1], [&](KT km, uint32_t t, uint32_t ex) {
            recs[c[KmerOps<KT>::hash(km) >> (64 - logp)]++] = Rec{km, t, ex};
        });
    });
    cnt.clear();
    stage("partition k-mers");

    // ---- 3. per partition: sort, group, intern colour lists ----
    Interner interner;
    std::vector<std::vector<DK>> dks(P);
    parallel_for(T, P, [&](uint64_t p, int) {
        Rec* b = recs.get() + pstart[p];
        Rec* e = recs.get() + pstart[p + 1];
        std::sort(b, e, [](const Rec& a, const Rec& c) { return a.kmer != c.kmer ? a.kmer < c.kmer : a.tx < c.tx; });
        auto& dk = dks[p];
        std::vector<uint32_t> list;
        for (Rec* i = b; i < e;) {
            Rec* j = i;
            uint32_t ex = 0;
            list.clear();
            for (; j < e && j->kmer == i->kmer; ++j) {
                ex |= j->exts;
                if (list.empty() || list.back() != j->tx) list.push_back(j->tx);
            }
            dk.push_back(DK{i->kmer, interner.intern(list.data(), (uint32_t)list.size()), ex});
            i = j;
        }
    });
    recs.reset();
    stage("sort + colour lists");

    // ---- 4. deterministic class numbering: lexicographic order of the id lists ----
    struct LRef { const uint32_t* p; uint32_t n; uint32_t temp; };
    std::vector<LRef> lrefs;
    for (int s = 0; s < Interner::SHARDS; ++s) {
        auto& sh = interner.shards[s];
        for (uint32_t li = 0; li + 1 < sh.off.size(); ++li)
            lrefs.push_back(LRef{sh.arena.data() + sh.off[li], (uint32_t)(sh.off[li + 1] - sh.off[li]), (li << 8) | (uint32_t)s});
    }
    std::sort(lrefs.begin(), lrefs.end(), [](const LRef& a, const LRef& b) {
        return std::lexicographical_compare(a.p, a.p + a.n, b.p, b.p + b.n);
    });
    if (lrefs.size() >= EMPTY) return fail(PA_ERR_UNSUPPORTED, "too many equivalence classes");
    std::vector<std::vector<uint32_t>> remap(Interner::SHARDS);
    for (int s = 0; s < Interner::SHARDS; ++s) remap[s].resize(interner.shards[s].off.size() - 1);
    for (uint32_t c = 0; c < lrefs.size(); ++c) {
        remap[lrefs[c].temp & 255][lrefs[c].temp >> 8] = c;
        out.ec_ids.insert(out.ec_ids.end(), lrefs[c].p, lrefs[c].p + lrefs[c].n);
        out.ec_offset.push_back(out.ec_ids.size());
    }

    stage("class numbering");
    // ---- 5. k-mer table ----
    uint64_t ndistinct = 0;
    for (auto& d : dks) ndistinct += d.size();
    KTable<KT> tab;
    tab.init(ndistinct, T);
    parallel_for(T, P, [&](uint64_t p, int) {
        for (auto& d : dks[p]) {
            d.colour = remap[d.colour & 255][d.colour >> 8];
            tab.insert_mt(d.kmer, d.colour, (uint8_t)d.exts);
        }
    });

    stage("k-mer table");
    // ---- 6. unitigs: walk right from every start k-mer (no joinable predecessor) ----
    auto popc4 = [](uint32_t x) { return __builtin_popcount(x & 15u); };
    // joinable successor of x, or nullptr
    auto right_join = [&](const KEntry& x) -> KEntry* {
        const uint32_t r = x.exts & 15u;
        if (popc4(r) != 1) return nullptr;
        const KT b = (KT)__builtin_ctz(r);
        const KT y = (x.kmer >> 2) | (b << topshift);
        if (y == x.kmer) return nullptr;
        KEntry* ey = tab.find(y);
        if (!ey) return nullptr;   // cannot happen for a consistent transcript set
        if (popc4(ey->exts >> 4) != 1 || ey->colour != x.colour) return nullptr;
        return ey;
    };
    auto left_joinable = [&](const KEntry& x) -> bool {
        const uint32_t l = (x.exts >> 4) & 15u;
        if (popc4(l) != 1) return false;
        const KT b = (KT)__builtin_ctz(l);
        const KT z = ((x.kmer << 2) | b) & mask;
        if (z == x.kmer) return false;
        KEntry* ez = tab.find(z);
        if (!ez) return false;
        return popc4(ez->exts) == 1 && ez->colour == x.colour;
    };
    auto walk = [&](KEntry* st, NodeOut& no) {
        for (uint32_t i = 0; i < k; ++i) no.push_base((uint32_t)(st->kmer >> (2 * i)) & 3u);
        uint32_t len = k;
        KEntry* cur = st;
        cur->visited = 1;
        for (;;) {
            KEntry* nx = right_join(*cur);
            if (!nx || nx->visited) break;
            nx->visited = 1;
            no.push_base((uint32_t)(nx->kmer >> topshift) & 3u);
            ++len;
            cur = nx;
        }
        no.len.push_back(len);
        no.colour.push_back(st->colour);
        no.exts.push_back((uint8_t)((st->exts & 0xF0u) | (cur->exts & 0x0Fu)));
    };
    std::vector<NodeOut> nouts(P + 1);
    parallel_for(T, P, [&](uint64_t p, int) {
        for (auto& d : dks[p]) {
            KEntry* e = tab.find(d.kmer);
            if (!left_joinable(*e)) walk(e, nouts[p]);
        }
    });
    // pure cycles of joinable k-mers: no start exists; break at the first k-mer in partition order
    for (uint32_t p = 0; p < P; ++p)
        for (auto& d : dks[p]) {
            KEntry* e = tab.find(d.kmer);
            if (!e->visited) walk(e, nouts[P]);
        }

    stage("unitig walks");
    // ---- 7. concatenate in partition order ----
    uint64_t nnodes = 0, nbases = 0;
    for (auto& no : nouts) { nnodes += no.len.size(); nbases += no.bases; }
    if (nnodes >= EMPTY) return fail(PA_ERR_UNSUPPORTED, "too many nodes");
    out.node_seq.assign((nbases + 31) / 32 + 2, 0);
    out.node_len.reserve(nnodes);
    out.node_colour.reserve(nnodes);
    out.node_exts.reserve(nnodes);
    out.node_start.reserve(nnodes + 1);
    uint64_t cursor = 0;
    for (auto& no : nouts) {
        uint64_t local = 0;
        for (size_t i = 0; i < no.len.size(); ++i) {
            const uint32_t len = no.len[i];
            for (uint32_t j = 0; j < len; ++j) set_base(out.node_seq.data(), cursor + j, get_base(no.seq.data(), local + j));
            cursor += len;
            local += len;
            out.node_start.push_back(cursor);
            out.node_len.push_back(len);
            out.node_colour.push_back(no.colour[i]);
            out.node_exts.push_back(no.exts[i]);
        }
        NodeOut().seq.swap(no.seq);
    }
    stage("concatenate");
    return PA_OK;
}

int build_graph(const uint64_t* packed, const uint64_t* tx_start, uint32_t num_tx, uint32_t k, int threads, HostIndex& out) {
    if (k < PA_MIN_K || k > PA_MAX_K) return fail(PA_ERR_UNSUPPORTED, "k=%u outside [%u,%u]", k, PA_MIN_K, PA_MAX_K);
    return k <= 32 ? build_graph_t<uint64_t>(packed, tx_start, num_tx, k, threads, out)
                   : build_graph_t<u128>(packed, tx_start, num_tx, k, threads, out);
}

}  // namespace pa
