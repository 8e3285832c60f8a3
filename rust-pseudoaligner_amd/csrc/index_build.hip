// Index construction on the GPU: transcripts -> stranded, coloured, compacted De Bruijn graph +
// equivalence classes, the same graph the CPU builder (dbg_build.cpp) produces — node for node, class for class: the tests
// compare the two flat indexes array by array.
//
// The reference reaches this graph through MSP sharding, per-shard compression and a merge pass on CPU threads
// (src/build_index.rs:127-179, src/equiv_classes.rs:62-91). Here every stage is a data-parallel pass over arrays in HBM:
//   1. every k-mer occurrence of every transcript -> (k-mer, transcript << 8 | extension bits)            pa_ib_enum_kernel
//   2. stable radix sort by k-mer (rocPRIM): the occurrences of a k-mer become a segment, transcripts ascending
//   3. per segment: OR of the extension bits, number of distinct transcripts, order-free hash of that set  pa_ib_segment_kernel
//   4. colours: segments sorted by set hash; a run of equal hashes is one class once a content comparison against the
//      run's first member has confirmed it (a mismatch re-runs the stage with another hash seed)            pa_ib_verify_kernel
//      the distinct lists go to the host, which numbers them in lexicographic order (CountFilterEqClass interns them in
//      arrival order, src/equiv_classes.rs:81-87: numbering is unobservable; ours is deterministic)
//   5. joins: k-mer x joins its successor y when y is x's only right extension, x is y's only left extension and both
//      carry the same colour (ScmapCompress, src/build_index.rs:171,178); successors are found by binary search in the
//      sorted distinct k-mers                                                                               pa_ib_links_kernel
//   6. unitigs by pointer jumping over the predecessor links: every k-mer learns its unitig's first k-mer and its offset
//      in O(log longest unitig) passes                                                                      pa_ib_jump_kernel
//   7. nodes ordered as the CPU builder orders them (hash partition of the first k-mer, then k-mer), lengths, extension
//      bits, colours, packed sequences (every k-mer writes its last base)                                   pa_ib_tails_kernel, pa_ib_seq_kernel
// Pure cycles of joinable k-mers (no first k-mer exists) are what is left unresolved after step 6; they are rare and
// small and are walked on the host exactly as dbg_build.cpp does.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <numeric>
#include <unordered_map>

#include "lane_steps.hpp"
#include "pa_common.hpp"

namespace pa {
namespace {

constexpr uint32_t NONE32 = 0xFFFFFFFFu;

struct DBuf {   // device allocation that frees itself
    void* p = nullptr;
    size_t bytes = 0;
    DBuf() = default;
    DBuf(const DBuf&) = delete;
    DBuf& operator=(const DBuf&) = delete;
    ~DBuf() { release(); }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    hipError_t alloc(size_t n) {
        release();
        if (n == 0) n = 16;
        const hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

#define IB_HIP(expr) do { const hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(PA_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)

// ---- k-mers of one or two words on the device ----
template <class KT> struct DKmer;
template <> struct DKmer<uint64_t> {
    __host__ __device__ static uint64_t mask(uint32_t k) { return k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1); }
    __device__ static uint64_t win(const uint64_t* w, uint64_t pos) {
        const uint64_t i = pos >> 5;
        const uint32_t s = (uint32_t)(pos & 31) * 2;
        return s ? (w[i] >> s) | (w[i + 1] << (64 - s)) : w[i];
    }
    __device__ static uint64_t get(const uint64_t* w, uint64_t pos, uint32_t k) { return win(w, pos) & mask(k); }
    __host__ __device__ static uint64_t hash(uint64_t km) { return pa_mix64(km); }
};
template <> struct DKmer<u128> {
    __host__ __device__ static u128 mask(uint32_t k) { return k >= 64 ? ~(u128)0 : (((u128)1 << (2 * k)) - 1); }
    __device__ static u128 get(const uint64_t* w, uint64_t pos, uint32_t k) {
        return ((u128)(DKmer<uint64_t>::win(w, pos + 32) & DKmer<uint64_t>::mask(k - 32)) << 64) | DKmer<uint64_t>::win(w, pos);
    }
    __host__ __device__ static uint64_t hash(u128 km) { return pa_mix64((uint64_t)km ^ (pa_mix64((uint64_t)(km >> 64)) * 0x9e3779b97f4a7c15ull)); }
};

__device__ __forceinline__ uint32_t base_at(const uint64_t* w, uint64_t pos) { return (uint32_t)(w[pos >> 5] >> ((pos & 31) * 2)) & 3u; }

// ---- 1. one record per k-mer occurrence; thread = base position of the concatenated transcripts ----
template <class KT>
__global__ __launch_bounds__(256) void pa_ib_enum_kernel(const uint64_t* __restrict__ packed, const uint64_t* __restrict__ tx_start,
                                                         const uint64_t* __restrict__ kcum, uint32_t num_tx, uint64_t total_bases, uint32_t k,
                                                         KT* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total_bases) return;
    uint32_t lo = 0, hi = num_tx - 1;   // the last transcript whose start is <= g (empty transcripts share a start with their successor)
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo + 1) / 2;
        if (tx_start[mid] <= g) lo = mid; else hi = mid - 1;
    }
    const uint64_t s = tx_start[lo], len = tx_start[lo + 1] - s, pos = g - s;
    if (len < k || pos > len - k) return;
    uint32_t ex = 0;
    if (pos > 0) ex |= 1u << (4 + base_at(packed, g - 1));            // Exts::from_dna_string (src/build_index.rs:144)
    if (pos + k < len) ex |= 1u << base_at(packed, g + k);
    const uint64_t at = kcum[lo] + pos;
    keys[at] = DKmer<KT>::get(packed, g, k);
    vals[at] = (lo << 8) | ex;
}

// ---- 3. segment heads / per-segment reduction ----
template <class KT>
__global__ __launch_bounds__(256) void pa_ib_heads_kernel(const KT* __restrict__ keys, uint64_t n, uint32_t* __restrict__ head) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

// seg[i] = 1-based index of record i's distinct k-mer. CountFilterEqClass::summarize (src/equiv_classes.rs:62-91): colour =
// sorted dedup'd transcript list, extensions = union
template <class KT>
__global__ __launch_bounds__(256) void pa_ib_segment_kernel(const KT* __restrict__ keys, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ seg,
                                                            uint64_t n, uint64_t seed, KT* __restrict__ dkmer, uint32_t* __restrict__ dfirst,
                                                            uint32_t* __restrict__ dexts, uint32_t* __restrict__ dcnt,
                                                            unsigned long long* __restrict__ dhash) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t d = seg[i] - 1;
    const bool head = i == 0 || seg[i - 1] != seg[i];
    const uint32_t v = vals[i];
    if (head) { dkmer[d] = keys[i]; dfirst[d] = (uint32_t)i; }
    if (dexts) atomicOr(dexts + d, v & 0xFFu);
    if (head || (vals[i - 1] >> 8) != (v >> 8)) {
        if (dcnt) atomicAdd(dcnt + d, 1u);
        atomicAdd(dhash + d, (unsigned long long)pa_mix64((uint64_t)(v >> 8) + seed));
    }
}

// ---- 4. colours ----
__global__ __launch_bounds__(256) void pa_ib_iota_kernel(uint32_t* out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (uint32_t)i;
}
__global__ __launch_bounds__(256) void pa_ib_runheads_kernel(const unsigned long long* __restrict__ hs, uint64_t n, uint32_t* __restrict__ rh) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    rh[j] = (j == 0 || hs[j] != hs[j - 1]) ? 1u : 0u;
}
// rid[j] = 1-based run of sorted position j; the run's first member (the smallest k-mer index: the sort is stable) represents it
__global__ __launch_bounds__(256) void pa_ib_runs_kernel(const uint32_t* __restrict__ ds, const uint32_t* __restrict__ rid, uint64_t n,
                                                         uint32_t* __restrict__ run_first, uint32_t* __restrict__ coltmp) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t r = rid[j] - 1;
    if (j == 0 || rid[j - 1] != rid[j]) run_first[r] = ds[j];
    coltmp[ds[j]] = r;
}
// distinct transcripts of segment d, in order: fn(tx) for each
template <class F>
__device__ __forceinline__ void for_each_tx(const uint32_t* vals, uint32_t first, uint32_t end, F fn) {
    uint32_t prev = NONE32;
    for (uint32_t i = first; i < end; ++i) {
        const uint32_t tx = vals[i] >> 8;
        if (tx != prev) { fn(tx); prev = tx; }
    }
}
// every member of a run must carry the id list of the run's first member (equal set hashes do not prove it)
__global__ __launch_bounds__(256) void pa_ib_verify_kernel(const uint32_t* __restrict__ vals, const uint32_t* __restrict__ dfirst, const uint32_t* __restrict__ dcnt,
                                                           const uint32_t* __restrict__ coltmp, const uint32_t* __restrict__ run_first, uint64_t D, uint64_t n,
                                                           uint32_t* __restrict__ mismatch) {
    const uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    const uint32_t rep = run_first[coltmp[d]];
    if (rep == d) return;
    if (dcnt[d] != dcnt[rep]) { atomicAdd(mismatch, 1u); return; }
    const uint32_t ea = d + 1 < D ? dfirst[d + 1] : (uint32_t)n, eb = (uint64_t)rep + 1 < D ? dfirst[rep + 1] : (uint32_t)n;
    uint32_t b = dfirst[rep], prevb = NONE32;
    bool same = true;
    for_each_tx(vals, dfirst[d], ea, [&](uint32_t tx) {
        uint32_t txb = NONE32;
        for (; b < eb; ++b) {
            const uint32_t t = vals[b] >> 8;
            if (t != prevb) { txb = t; prevb = t; ++b; break; }
        }
        same = same && txb == tx;
    });
    if (!same) atomicAdd(mismatch, 1u);
}
__global__ __launch_bounds__(256) void pa_ib_listlen_kernel(const uint32_t* __restrict__ run_first, const uint32_t* __restrict__ dcnt, uint64_t C,
                                                            unsigned long long* __restrict__ len) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < C) len[r] = dcnt[run_first[r]];
}
__global__ __launch_bounds__(256) void pa_ib_listwrite_kernel(const uint32_t* __restrict__ vals, const uint32_t* __restrict__ dfirst, const uint32_t* __restrict__ run_first,
                                                              const unsigned long long* __restrict__ off, uint64_t C, uint64_t D, uint64_t n,
                                                              uint32_t* __restrict__ ids) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= C) return;
    const uint32_t d = run_first[r];
    const uint32_t e = (uint64_t)d + 1 < D ? dfirst[d + 1] : (uint32_t)n;
    uint32_t* dst = ids + off[r];
    for_each_tx(vals, dfirst[d], e, [&](uint32_t tx) { *dst++ = tx; });
}
__global__ __launch_bounds__(256) void pa_ib_colour_kernel(const uint32_t* __restrict__ coltmp, const uint32_t* __restrict__ remap, uint64_t D, uint32_t* __restrict__ dcol) {
    const uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d < D) dcol[d] = remap[coltmp[d]];
}

// ---- 5. joins ----
template <class KT>
__device__ __forceinline__ uint32_t find_kmer(const KT* __restrict__ dkmer, uint32_t D, KT key) {
    uint32_t lo = 0, hi = D;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (dkmer[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo < D && dkmer[lo] == key ? lo : NONE32;
}
// pd[d] = (pointer << 32 | distance): a first k-mer points to itself at distance 0, every other k-mer to its predecessor at
// distance 1 (right_join / left_joinable of dbg_build.cpp, i.e. ScmapCompress: unique extension both ways, same colour)
template <class KT>
__global__ __launch_bounds__(256) void pa_ib_links_kernel(const KT* __restrict__ dkmer, const uint32_t* __restrict__ dexts, const uint32_t* __restrict__ dcol,
                                                          uint32_t D, uint32_t k, uint32_t* __restrict__ succ, unsigned long long* __restrict__ pd) {
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    const KT x = dkmer[d], mask = DKmer<KT>::mask(k);
    const uint32_t e = dexts[d], col = dcol[d], topshift = 2 * (k - 1);
    uint32_t sc = NONE32, pr = d;
    const uint32_t r = e & 15u;
    if (__popc(r) == 1) {
        const KT y = (x >> 2) | ((KT)(__ffs((int)r) - 1) << topshift);
        if (y != x) {
            const uint32_t j = find_kmer(dkmer, D, y);
            if (j != NONE32 && __popc((dexts[j] >> 4) & 15u) == 1 && dcol[j] == col) sc = j;
        }
    }
    const uint32_t l = (e >> 4) & 15u;
    if (__popc(l) == 1) {
        const KT z = ((x << 2) | (KT)(__ffs((int)l) - 1)) & mask;
        if (z != x) {
            const uint32_t j = find_kmer(dkmer, D, z);
            if (j != NONE32 && __popc(dexts[j] & 15u) == 1 && dcol[j] == col) pr = j;
        }
    }
    succ[d] = sc;
    pd[d] = ((unsigned long long)pr << 32) | (pr == d ? 0u : 1u);
}
// ---- 6. pointer jumping. In place: a (pointer, distance) pair is read and written as one 64-bit word, and any pair a
// thread can observe is a true statement ("distance steps back from here is that k-mer"), so racing updates only speed it up
__global__ __launch_bounds__(256) void pa_ib_jump_kernel(unsigned long long* __restrict__ pd, uint32_t D, uint32_t* __restrict__ unresolved) {
    // a FIRST k-mer is (itself, distance 0). On a pure cycle the jumps can bring a pointer back to its own k-mer, but then at a
    // distance that is not 0 (it saturates instead of wrapping), so a cycle member is never taken for a first k-mer.
    // Grid-stride, one atomic per WORKGROUP: a counter hit once per wave (1.6 M times per pass at config 3) is one hot word
    // and was the whole cost of a pass (15 ms)
    uint32_t mine = 0;
    for (uint64_t d64 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; d64 < D; d64 += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t d = (uint32_t)d64;
        const unsigned long long me = __atomic_load_n(pd + d, __ATOMIC_RELAXED);
        if (me == ((unsigned long long)d << 32)) continue;
        const uint32_t p = (uint32_t)(me >> 32);
        const unsigned long long up = __atomic_load_n(pd + p, __ATOMIC_RELAXED);
        if (up == ((unsigned long long)p << 32)) continue;   // p is a first k-mer: resolved
        uint32_t dist = (uint32_t)me + (uint32_t)up;
        if (dist < (uint32_t)me) dist = 0xFFFFFFFFu;
        __atomic_store_n(pd + d, (up & 0xFFFFFFFF00000000ull) | dist, __ATOMIC_RELAXED);
        ++mine;
    }
    __shared__ uint32_t block_sum;
    if (threadIdx.x == 0) block_sum = 0;
    __syncthreads();
    if (mine) atomicAdd(&block_sum, mine);
    __syncthreads();
    if (threadIdx.x == 0 && block_sum) atomicAdd(unresolved, block_sum);
}
// flags: first k-mer of a unitig / member of a pure cycle (its pointer never reaches a first k-mer)
__global__ __launch_bounds__(256) void pa_ib_classify_kernel(const unsigned long long* __restrict__ pd, uint32_t D, uint32_t* __restrict__ is_start,
                                                             uint32_t* __restrict__ is_cyclic) {
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    const unsigned long long me = pd[d];
    const uint32_t p = (uint32_t)(me >> 32);
    const bool first = me == ((unsigned long long)d << 32);
    is_start[d] = first ? 1u : 0u;
    is_cyclic[d] = (!first && pd[p] != ((unsigned long long)p << 32)) ? 1u : 0u;
}
// ---- 7. nodes ----
template <class KT>
__global__ __launch_bounds__(256) void pa_ib_partkey_kernel(const KT* __restrict__ dkmer, const uint32_t* __restrict__ starts, uint32_t ns, uint32_t logp,
                                                            uint32_t* __restrict__ key) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < ns) key[n] = (uint32_t)(DKmer<KT>::hash(dkmer[starts[n]]) >> (64 - logp));
}
__global__ __launch_bounds__(256) void pa_ib_nodeof_kernel(const uint32_t* __restrict__ order, uint32_t ns, uint32_t* __restrict__ node_of) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < ns) node_of[order[n]] = n;
}
// the last k-mer of a unitig (no joinable successor) knows the unitig's length
__global__ __launch_bounds__(256) void pa_ib_tails_kernel(const unsigned long long* __restrict__ pd, const uint32_t* __restrict__ succ, const uint32_t* __restrict__ is_cyclic,
                                                          const uint32_t* __restrict__ node_of, const uint32_t* __restrict__ dexts, const uint32_t* __restrict__ dcol,
                                                          uint32_t D, uint32_t k, uint32_t* __restrict__ node_len, unsigned long long* __restrict__ node_len64,
                                                          uint8_t* __restrict__ node_exts, uint32_t* __restrict__ node_colour) {
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D || is_cyclic[d] || succ[d] != NONE32) return;
    const unsigned long long me = pd[d];
    const uint32_t first = (uint32_t)(me >> 32), n = node_of[first];
    const uint32_t len = (uint32_t)me + k;   // distance + 1 k-mers
    node_len[n] = len;
    node_len64[n] = len;
    node_exts[n] = (uint8_t)((dexts[first] & 0xF0u) | (dexts[d] & 0x0Fu));
    node_colour[n] = dcol[first];
}
__device__ __forceinline__ void or_bits(unsigned long long* seq, uint64_t base_pos, uint64_t bits, uint32_t nbases) {   // nbases <= 32
    const uint64_t w = base_pos >> 5;
    const uint32_t o = (uint32_t)(base_pos & 31) * 2;
    atomicOr(seq + w, (unsigned long long)(bits << o));
    if (o && o + 2 * nbases > 64) atomicOr(seq + w + 1, (unsigned long long)(bits >> (64 - o)));
}
// the first k-mer of a unitig writes its k bases, every other k-mer its last base
template <class KT>
__global__ __launch_bounds__(256) void pa_ib_seq_kernel(const KT* __restrict__ dkmer, const unsigned long long* __restrict__ pd, const uint32_t* __restrict__ is_cyclic,
                                                        const uint32_t* __restrict__ node_of, const unsigned long long* __restrict__ node_start, uint32_t D,
                                                        uint32_t k, unsigned long long* __restrict__ seq) {
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D || is_cyclic[d]) return;
    const unsigned long long me = pd[d];
    const uint32_t dist = (uint32_t)me;
    const uint64_t at = node_start[node_of[(uint32_t)(me >> 32)]] + dist;
    const KT x = dkmer[d];
    if (dist == 0) {
        or_bits(seq, at, (uint64_t)x, k < 32 ? k : 32);
        if (k > 32) or_bits(seq, at + 32, (uint64_t)(x >> (sizeof(KT) > 8 ? 64 : 0)), k - 32);
    } else {
        or_bits(seq, at + k - 1, (uint64_t)(x >> (2 * (k - 1))) & 3u, 1);
    }
}
template <class KT>
__global__ __launch_bounds__(256) void pa_ib_gather_kernel(const uint32_t* __restrict__ which, uint32_t n, const KT* __restrict__ dkmer, const uint32_t* __restrict__ dexts,
                                                           const uint32_t* __restrict__ dcol, const uint32_t* __restrict__ succ, KT* __restrict__ okmer,
                                                           uint32_t* __restrict__ oexts, uint32_t* __restrict__ ocol, uint32_t* __restrict__ osucc) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t d = which[i];
    okmer[i] = dkmer[d]; oexts[i] = dexts[d]; ocol[i] = dcol[d]; osucc[i] = succ[d];
}

inline dim3 grid_of(uint64_t n) { return dim3((uint32_t)((n + 255) / 256)); }

struct Stage {   // PA_VERBOSE: stage times (device work is synchronised at each mark)
    bool on;
    std::chrono::steady_clock::time_point t;
    Stage() : on(std::getenv("PA_VERBOSE") != nullptr), t(std::chrono::steady_clock::now()) {}
    void mark(const char* what) {
        if (!on) return;
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[pa build gpu] %-28s %.3f s\n", what, std::chrono::duration<double>(now - t).count());
        t = now;
    }
};

template <class K, class V>
int sort_pairs(const K* kin, K* kout, const V* vin, V* vout, uint64_t n, uint32_t begin_bit, uint32_t end_bit) {
    size_t bytes = 0;
    IB_HIP(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, (size_t)n, begin_bit, end_bit, (hipStream_t) nullptr));
    DBuf tmp;
    IB_HIP(tmp.alloc(bytes));
    IB_HIP(rocprim::radix_sort_pairs(tmp.p, bytes, kin, kout, vin, vout, (size_t)n, begin_bit, end_bit, (hipStream_t) nullptr));
    IB_HIP(hipStreamSynchronize(nullptr));   // tmp is freed on return
    return PA_OK;
}
template <class T>
int scan_incl(const T* in, T* out, uint64_t n) {
    size_t bytes = 0;
    IB_HIP(rocprim::inclusive_scan(nullptr, bytes, in, out, (size_t)n, rocprim::plus<T>(), (hipStream_t) nullptr));
    DBuf tmp;
    IB_HIP(tmp.alloc(bytes));
    IB_HIP(rocprim::inclusive_scan(tmp.p, bytes, in, out, (size_t)n, rocprim::plus<T>(), (hipStream_t) nullptr));
    IB_HIP(hipStreamSynchronize(nullptr));
    return PA_OK;
}
template <class T>
int scan_excl(const T* in, T* out, uint64_t n) {
    size_t bytes = 0;
    IB_HIP(rocprim::exclusive_scan(nullptr, bytes, in, out, T(0), (size_t)n, rocprim::plus<T>(), (hipStream_t) nullptr));
    DBuf tmp;
    IB_HIP(tmp.alloc(bytes));
    IB_HIP(rocprim::exclusive_scan(tmp.p, bytes, in, out, T(0), (size_t)n, rocprim::plus<T>(), (hipStream_t) nullptr));
    IB_HIP(hipStreamSynchronize(nullptr));
    return PA_OK;
}
// indices i in [0, n) with flags[i] != 0, ascending
int select_flagged(const uint32_t* flags, uint64_t n, uint32_t* out, uint32_t* count_host) {
    DBuf cnt;
    IB_HIP(cnt.alloc(8));
    size_t bytes = 0;
    rocprim::counting_iterator<uint32_t> it(0);
    IB_HIP(rocprim::select(nullptr, bytes, it, flags, out, cnt.as<uint32_t>(), (size_t)n, (hipStream_t) nullptr));
    DBuf tmp;
    IB_HIP(tmp.alloc(bytes));
    IB_HIP(rocprim::select(tmp.p, bytes, it, flags, out, cnt.as<uint32_t>(), (size_t)n, (hipStream_t) nullptr));
    IB_HIP(hipMemcpy(count_host, cnt.p, 4, hipMemcpyDeviceToHost));
    return PA_OK;
}

template <class KT>
int build_graph_device_t(const uint64_t* packed_in, const uint64_t* tx_start, uint32_t num_tx, uint32_t k, HostIndex& out) {
    Stage stage;
    out = HostIndex();
    out.k = k;
    out.num_transcripts = num_tx;
    out.node_start.push_back(0);
    out.ec_offset.push_back(0);
    // ---- k-mers per transcript ----
    std::vector<uint64_t> kcum((size_t)num_tx + 1, 0);
    for (uint32_t t = 0; t < num_tx; ++t) {
        const uint64_t len = tx_start[t + 1] - tx_start[t];
        kcum[t + 1] = kcum[t] + (len >= k ? len - k + 1 : 0);
    }
    const uint64_t N = kcum[num_tx], total_bases = num_tx ? tx_start[num_tx] : 0;
    if (N == 0) { out.node_seq.assign(2, 0); return PA_OK; }
    if (num_tx >= (1u << 24)) return fail(PA_ERR_UNSUPPORTED, "the GPU builder packs transcript ids into 24 bits (%u transcripts)", num_tx);
    if (N >= 0xFFFFFFFFull) return fail(PA_ERR_UNSUPPORTED, "the GPU builder handles fewer than 2^32 k-mer occurrences (%llu)", (unsigned long long)N);
    uint32_t logp = 4;   // partitions of dbg_build.cpp: node order = (partition of the first k-mer, k-mer)
    while (logp < 12 && (N >> logp) > (1u << 20)) ++logp;
    const uint32_t P = 1u << logp;

    const uint64_t nwords = (total_bases + 31) / 32;
    DBuf d_packed, d_txs, d_kcum;
    IB_HIP(d_packed.alloc((nwords + 3) * 8));
    IB_HIP(hipMemset(d_packed.p, 0, (nwords + 3) * 8));
    IB_HIP(hipMemcpy(d_packed.p, packed_in, nwords * 8, hipMemcpyHostToDevice));
    IB_HIP(d_txs.alloc(((size_t)num_tx + 1) * 8));
    IB_HIP(hipMemcpy(d_txs.p, tx_start, ((size_t)num_tx + 1) * 8, hipMemcpyHostToDevice));
    IB_HIP(d_kcum.alloc(((size_t)num_tx + 1) * 8));
    IB_HIP(hipMemcpy(d_kcum.p, kcum.data(), ((size_t)num_tx + 1) * 8, hipMemcpyHostToDevice));

    // ---- 1 + 2. records, sorted by k-mer ----
    DBuf keys, vals;
    {
        DBuf keys0, vals0;
        IB_HIP(keys0.alloc(N * sizeof(KT)));
        IB_HIP(vals0.alloc(N * 4));
        IB_HIP(keys.alloc(N * sizeof(KT)));
        IB_HIP(vals.alloc(N * 4));
        hipLaunchKernelGGL(pa_ib_enum_kernel<KT>, grid_of(total_bases), dim3(256), 0, nullptr, d_packed.as<uint64_t>(), d_txs.as<uint64_t>(), d_kcum.as<uint64_t>(),
                           num_tx, total_bases, k, keys0.as<KT>(), vals0.as<uint32_t>());
        IB_HIP(hipGetLastError());
        stage.mark("enumerate k-mers");
        const int rc = sort_pairs(keys0.as<KT>(), keys.as<KT>(), vals0.as<uint32_t>(), vals.as<uint32_t>(), N, 0, 2 * k);
        if (rc != PA_OK) return rc;
        stage.mark("radix sort");
    }
    // ---- 3. segments ----
    DBuf seg;
    IB_HIP(seg.alloc(N * 4));
    {
        DBuf head;
        IB_HIP(head.alloc(N * 4));
        hipLaunchKernelGGL(pa_ib_heads_kernel<KT>, grid_of(N), dim3(256), 0, nullptr, keys.as<KT>(), N, head.as<uint32_t>());
        IB_HIP(hipGetLastError());
        const int rc = scan_incl(head.as<uint32_t>(), seg.as<uint32_t>(), N);
        if (rc != PA_OK) return rc;
    }
    uint32_t D = 0;
    IB_HIP(hipMemcpy(&D, seg.as<uint32_t>() + (N - 1), 4, hipMemcpyDeviceToHost));
    DBuf dkmer, dfirst, dexts, dcnt, dhash, coltmp, run_first;
    IB_HIP(dkmer.alloc((size_t)D * sizeof(KT)));
    IB_HIP(dfirst.alloc((size_t)D * 4));
    IB_HIP(dexts.alloc((size_t)D * 4));
    IB_HIP(dcnt.alloc((size_t)D * 4));
    IB_HIP(dhash.alloc((size_t)D * 8));
    IB_HIP(coltmp.alloc((size_t)D * 4));
    IB_HIP(hipMemset(dexts.p, 0, (size_t)D * 4));
    IB_HIP(hipMemset(dcnt.p, 0, (size_t)D * 4));
    // ---- 4. colours: runs of equal set hash, verified by content ----
    uint32_t C = 0;
    uint64_t seed = 0x243f6a8885a308d3ull;
    for (int attempt = 0;; ++attempt) {
        IB_HIP(hipMemset(dhash.p, 0, (size_t)D * 8));
        hipLaunchKernelGGL(pa_ib_segment_kernel<KT>, grid_of(N), dim3(256), 0, nullptr, keys.as<KT>(), vals.as<uint32_t>(), seg.as<uint32_t>(), N, seed,
                           dkmer.as<KT>(), dfirst.as<uint32_t>(), attempt == 0 ? dexts.as<uint32_t>() : nullptr, attempt == 0 ? dcnt.as<uint32_t>() : nullptr,
                           dhash.as<unsigned long long>());
        IB_HIP(hipGetLastError());
        DBuf hs, ds0, ds, rid, rh;
        IB_HIP(hs.alloc((size_t)D * 8));
        IB_HIP(ds0.alloc((size_t)D * 4));
        IB_HIP(ds.alloc((size_t)D * 4));
        IB_HIP(rid.alloc((size_t)D * 4));
        IB_HIP(rh.alloc((size_t)D * 4));
        hipLaunchKernelGGL(pa_ib_iota_kernel, grid_of(D), dim3(256), 0, nullptr, ds0.as<uint32_t>(), (uint64_t)D);
        { const int rc = sort_pairs(dhash.as<unsigned long long>(), hs.as<unsigned long long>(), ds0.as<uint32_t>(), ds.as<uint32_t>(), D, 0, 64); if (rc != PA_OK) return rc; }
        hipLaunchKernelGGL(pa_ib_runheads_kernel, grid_of(D), dim3(256), 0, nullptr, hs.as<unsigned long long>(), (uint64_t)D, rh.as<uint32_t>());
        IB_HIP(hipGetLastError());
        { const int rc = scan_incl(rh.as<uint32_t>(), rid.as<uint32_t>(), D); if (rc != PA_OK) return rc; }
        IB_HIP(hipMemcpy(&C, rid.as<uint32_t>() + (D - 1), 4, hipMemcpyDeviceToHost));
        IB_HIP(run_first.alloc((size_t)C * 4));
        hipLaunchKernelGGL(pa_ib_runs_kernel, grid_of(D), dim3(256), 0, nullptr, ds.as<uint32_t>(), rid.as<uint32_t>(), (uint64_t)D, run_first.as<uint32_t>(),
                           coltmp.as<uint32_t>());
        IB_HIP(hipGetLastError());
        DBuf mism;
        IB_HIP(mism.alloc(4));
        IB_HIP(hipMemset(mism.p, 0, 4));
        hipLaunchKernelGGL(pa_ib_verify_kernel, grid_of(D), dim3(256), 0, nullptr, vals.as<uint32_t>(), dfirst.as<uint32_t>(), dcnt.as<uint32_t>(), coltmp.as<uint32_t>(),
                           run_first.as<uint32_t>(), (uint64_t)D, N, mism.as<uint32_t>());
        IB_HIP(hipGetLastError());
        uint32_t bad = 0;
        IB_HIP(hipMemcpy(&bad, mism.p, 4, hipMemcpyDeviceToHost));
        if (bad == 0) break;
        if (attempt == 3) return fail(PA_ERR_INTERNAL, "colour interning: set hashes kept colliding (%u k-mers)", bad);
        seed = pa_mix64(seed + attempt + 1);   // two different id lists shared a hash: take another hash
    }
    stage.mark("segments + colour runs");
    // the distinct lists -> host, numbered in lexicographic order
    std::vector<unsigned long long> loff((size_t)C + 1, 0);
    std::vector<uint32_t> lids;
    {
        DBuf llen, d_off, d_ids;
        IB_HIP(llen.alloc(((size_t)C + 1) * 8));
        IB_HIP(d_off.alloc(((size_t)C + 1) * 8));
        IB_HIP(hipMemset(llen.p, 0, ((size_t)C + 1) * 8));
        hipLaunchKernelGGL(pa_ib_listlen_kernel, grid_of(C), dim3(256), 0, nullptr, run_first.as<uint32_t>(), dcnt.as<uint32_t>(), (uint64_t)C, llen.as<unsigned long long>());
        IB_HIP(hipGetLastError());
        { const int rc = scan_excl(llen.as<unsigned long long>(), d_off.as<unsigned long long>(), (uint64_t)C + 1); if (rc != PA_OK) return rc; }
        IB_HIP(hipMemcpy(loff.data(), d_off.p, ((size_t)C + 1) * 8, hipMemcpyDeviceToHost));
        const uint64_t nids = loff[C];
        IB_HIP(d_ids.alloc(nids * 4));
        hipLaunchKernelGGL(pa_ib_listwrite_kernel, grid_of(C), dim3(256), 0, nullptr, vals.as<uint32_t>(), dfirst.as<uint32_t>(), run_first.as<uint32_t>(),
                           d_off.as<unsigned long long>(), (uint64_t)C, (uint64_t)D, N, d_ids.as<uint32_t>());
        IB_HIP(hipGetLastError());
        lids.resize(nids);
        IB_HIP(hipMemcpy(lids.data(), d_ids.p, nids * 4, hipMemcpyDeviceToHost));
    }
    if (C >= NONE32) return fail(PA_ERR_UNSUPPORTED, "too many equivalence classes");
    std::vector<uint32_t> lorder(C), remap(C);
    std::iota(lorder.begin(), lorder.end(), 0u);
    std::sort(lorder.begin(), lorder.end(), [&](uint32_t a, uint32_t b) {
        return std::lexicographical_compare(lids.begin() + loff[a], lids.begin() + loff[a + 1], lids.begin() + loff[b], lids.begin() + loff[b + 1]);
    });
    out.ec_ids.reserve(lids.size());
    for (uint32_t c = 0; c < C; ++c) {
        const uint32_t r = lorder[c];
        remap[r] = c;
        out.ec_ids.insert(out.ec_ids.end(), lids.begin() + loff[r], lids.begin() + loff[r + 1]);
        out.ec_offset.push_back(out.ec_ids.size());
    }
    // the occurrence records are no longer needed
    keys.release(); vals.release(); seg.release(); dhash.release(); dfirst.release(); run_first.release(); dcnt.release();
    DBuf dcol;
    {
        DBuf d_remap;
        IB_HIP(d_remap.alloc((size_t)C * 4));
        IB_HIP(hipMemcpy(d_remap.p, remap.data(), (size_t)C * 4, hipMemcpyHostToDevice));
        IB_HIP(dcol.alloc((size_t)D * 4));
        hipLaunchKernelGGL(pa_ib_colour_kernel, grid_of(D), dim3(256), 0, nullptr, coltmp.as<uint32_t>(), d_remap.as<uint32_t>(), (uint64_t)D, dcol.as<uint32_t>());
        IB_HIP(hipGetLastError());
        IB_HIP(hipStreamSynchronize(nullptr));
    }
    coltmp.release();
    stage.mark("class numbering");

    // ---- 5 + 6. joins, pointer jumping ----
    DBuf succ, pd;
    IB_HIP(succ.alloc((size_t)D * 4));
    IB_HIP(pd.alloc((size_t)D * 8));
    hipLaunchKernelGGL(pa_ib_links_kernel<KT>, grid_of(D), dim3(256), 0, nullptr, dkmer.as<KT>(), dexts.as<uint32_t>(), dcol.as<uint32_t>(), D, k, succ.as<uint32_t>(),
                       pd.as<unsigned long long>());
    IB_HIP(hipGetLastError());
    stage.mark("joins");
    {
        DBuf cnt;
        IB_HIP(cnt.alloc(4));
        uint32_t prev = NONE32;
        for (int round = 0; round < 40; ++round) {
            IB_HIP(hipMemset(cnt.p, 0, 4));
            const uint32_t jump_blocks = (uint32_t)std::min<uint64_t>(((uint64_t)D + 255) / 256, 8192);
            hipLaunchKernelGGL(pa_ib_jump_kernel, dim3(jump_blocks), dim3(256), 0, nullptr, pd.as<unsigned long long>(), D, cnt.as<uint32_t>());
            IB_HIP(hipGetLastError());
            uint32_t now = 0;
            IB_HIP(hipMemcpy(&now, cnt.p, 4, hipMemcpyDeviceToHost));
            if (now == 0 || now == prev) break;   // only the members of pure cycles keep moving
            prev = now;
        }
    }
    stage.mark("pointer jumping");
    // ---- 7. nodes ----
    DBuf is_start, is_cyclic, starts, cyc;
    IB_HIP(is_start.alloc((size_t)D * 4));
    IB_HIP(is_cyclic.alloc((size_t)D * 4));
    hipLaunchKernelGGL(pa_ib_classify_kernel, grid_of(D), dim3(256), 0, nullptr, pd.as<unsigned long long>(), D, is_start.as<uint32_t>(), is_cyclic.as<uint32_t>());
    IB_HIP(hipGetLastError());
    IB_HIP(starts.alloc((size_t)D * 4));
    uint32_t ns = 0, ncyc = 0;
    { const int rc = select_flagged(is_start.as<uint32_t>(), D, starts.as<uint32_t>(), &ns); if (rc != PA_OK) return rc; }
    IB_HIP(cyc.alloc((size_t)D * 4));
    { const int rc = select_flagged(is_cyclic.as<uint32_t>(), D, cyc.as<uint32_t>(), &ncyc); if (rc != PA_OK) return rc; }
    is_start.release();
    if (ns >= NONE32) return fail(PA_ERR_UNSUPPORTED, "too many nodes");
    std::vector<uint32_t> h_len(ns), h_col(ns);
    std::vector<uint8_t> h_exts(ns);
    std::vector<unsigned long long> h_start((size_t)ns + 1, 0);
    std::vector<uint64_t> h_seq;
    uint64_t nbases = 0;
    if (ns) {
        DBuf pkey, pkey2, order, node_of, nlen, nlen64, nstart, nexts, ncol, seq;
        IB_HIP(pkey.alloc((size_t)ns * 4));
        IB_HIP(pkey2.alloc((size_t)ns * 4));
        IB_HIP(order.alloc((size_t)ns * 4));
        hipLaunchKernelGGL(pa_ib_partkey_kernel<KT>, grid_of(ns), dim3(256), 0, nullptr, dkmer.as<KT>(), starts.as<uint32_t>(), ns, logp, pkey.as<uint32_t>());
        IB_HIP(hipGetLastError());
        // the first k-mers are in k-mer order; a STABLE sort by partition gives (partition, k-mer) order
        { const int rc = sort_pairs(pkey.as<uint32_t>(), pkey2.as<uint32_t>(), starts.as<uint32_t>(), order.as<uint32_t>(), ns, 0, logp); if (rc != PA_OK) return rc; }
        IB_HIP(node_of.alloc((size_t)D * 4));
        hipLaunchKernelGGL(pa_ib_nodeof_kernel, grid_of(ns), dim3(256), 0, nullptr, order.as<uint32_t>(), ns, node_of.as<uint32_t>());
        IB_HIP(hipGetLastError());
        IB_HIP(nlen.alloc((size_t)ns * 4));
        IB_HIP(nlen64.alloc(((size_t)ns + 1) * 8));
        IB_HIP(nstart.alloc(((size_t)ns + 1) * 8));
        IB_HIP(nexts.alloc(ns));
        IB_HIP(ncol.alloc((size_t)ns * 4));
        IB_HIP(hipMemset(nlen64.p, 0, ((size_t)ns + 1) * 8));
        hipLaunchKernelGGL(pa_ib_tails_kernel, grid_of(D), dim3(256), 0, nullptr, pd.as<unsigned long long>(), succ.as<uint32_t>(), is_cyclic.as<uint32_t>(),
                           node_of.as<uint32_t>(), dexts.as<uint32_t>(), dcol.as<uint32_t>(), D, k, nlen.as<uint32_t>(), nlen64.as<unsigned long long>(),
                           nexts.as<uint8_t>(), ncol.as<uint32_t>());
        IB_HIP(hipGetLastError());
        { const int rc = scan_excl(nlen64.as<unsigned long long>(), nstart.as<unsigned long long>(), (uint64_t)ns + 1); if (rc != PA_OK) return rc; }
        IB_HIP(hipMemcpy(h_start.data(), nstart.p, ((size_t)ns + 1) * 8, hipMemcpyDeviceToHost));
        nbases = h_start[ns];
        const uint64_t sw = (nbases + 31) / 32 + 2;
        IB_HIP(seq.alloc(sw * 8));
        IB_HIP(hipMemset(seq.p, 0, sw * 8));
        hipLaunchKernelGGL(pa_ib_seq_kernel<KT>, grid_of(D), dim3(256), 0, nullptr, dkmer.as<KT>(), pd.as<unsigned long long>(), is_cyclic.as<uint32_t>(),
                           node_of.as<uint32_t>(), nstart.as<unsigned long long>(), D, k, seq.as<unsigned long long>());
        IB_HIP(hipGetLastError());
        h_seq.resize(sw);
        IB_HIP(hipMemcpy(h_seq.data(), seq.p, sw * 8, hipMemcpyDeviceToHost));
        IB_HIP(hipMemcpy(h_len.data(), nlen.p, (size_t)ns * 4, hipMemcpyDeviceToHost));
        IB_HIP(hipMemcpy(h_col.data(), ncol.p, (size_t)ns * 4, hipMemcpyDeviceToHost));
        IB_HIP(hipMemcpy(h_exts.data(), nexts.p, ns, hipMemcpyDeviceToHost));
    }
    stage.mark("nodes + sequences");
    // ---- pure cycles: walked on the host in (partition, k-mer) order, exactly as dbg_build.cpp does after its start walks ----
    struct CycNode { std::vector<uint32_t> bases; uint32_t colour; uint8_t exts; };
    std::vector<CycNode> cyc_nodes;
    if (ncyc) {
        DBuf ck, ce, cc, cs;
        IB_HIP(ck.alloc((size_t)ncyc * sizeof(KT)));
        IB_HIP(ce.alloc((size_t)ncyc * 4));
        IB_HIP(cc.alloc((size_t)ncyc * 4));
        IB_HIP(cs.alloc((size_t)ncyc * 4));
        hipLaunchKernelGGL(pa_ib_gather_kernel<KT>, grid_of(ncyc), dim3(256), 0, nullptr, cyc.as<uint32_t>(), ncyc, dkmer.as<KT>(), dexts.as<uint32_t>(),
                           dcol.as<uint32_t>(), succ.as<uint32_t>(), ck.as<KT>(), ce.as<uint32_t>(), cc.as<uint32_t>(), cs.as<uint32_t>());
        IB_HIP(hipGetLastError());
        std::vector<uint32_t> which(ncyc), ce_h(ncyc), cc_h(ncyc), cs_h(ncyc);
        std::vector<KT> ck_h(ncyc);
        IB_HIP(hipMemcpy(which.data(), cyc.p, (size_t)ncyc * 4, hipMemcpyDeviceToHost));
        IB_HIP(hipMemcpy(ck_h.data(), ck.p, (size_t)ncyc * sizeof(KT), hipMemcpyDeviceToHost));
        IB_HIP(hipMemcpy(ce_h.data(), ce.p, (size_t)ncyc * 4, hipMemcpyDeviceToHost));
        IB_HIP(hipMemcpy(cc_h.data(), cc.p, (size_t)ncyc * 4, hipMemcpyDeviceToHost));
        IB_HIP(hipMemcpy(cs_h.data(), cs.p, (size_t)ncyc * 4, hipMemcpyDeviceToHost));
        std::unordered_map<uint32_t, uint32_t> local;   // k-mer index -> position in the gathered arrays
        local.reserve(ncyc * 2);
        for (uint32_t i = 0; i < ncyc; ++i) local[which[i]] = i;
        std::vector<uint32_t> ord(ncyc);
        std::iota(ord.begin(), ord.end(), 0u);
        std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) {   // `which` ascends with the k-mer
            return (DKmer<KT>::hash(ck_h[a]) >> (64 - logp)) < (DKmer<KT>::hash(ck_h[b]) >> (64 - logp));
        });
        std::vector<uint8_t> visited(ncyc, 0);
        const uint32_t topshift = 2 * (k - 1);
        for (uint32_t oi = 0; oi < ncyc; ++oi) {
            const uint32_t st = ord[oi];
            if (visited[st]) continue;
            CycNode nd;
            for (uint32_t i = 0; i < k; ++i) nd.bases.push_back((uint32_t)(ck_h[st] >> (2 * i)) & 3u);
            uint32_t cur = st;
            visited[cur] = 1;
            for (;;) {
                const uint32_t nx_d = cs_h[cur];
                if (nx_d == NONE32) break;
                const auto it = local.find(nx_d);
                if (it == local.end() || visited[it->second]) break;
                cur = it->second;
                visited[cur] = 1;
                nd.bases.push_back((uint32_t)(ck_h[cur] >> topshift) & 3u);
            }
            nd.colour = cc_h[st];
            nd.exts = (uint8_t)((ce_h[st] & 0xF0u) | (ce_h[cur] & 0x0Fu));
            cyc_nodes.push_back(std::move(nd));
        }
    }
    (void)P;
    // ---- assemble ----
    uint64_t total_bases_out = nbases;
    for (const auto& c : cyc_nodes) total_bases_out += c.bases.size();
    const uint64_t nn = (uint64_t)ns + cyc_nodes.size();
    if (nn >= NONE32) return fail(PA_ERR_UNSUPPORTED, "too many nodes");
    out.node_seq.assign((total_bases_out + 31) / 32 + 2, 0);
    std::copy(h_seq.begin(), h_seq.begin() + std::min<size_t>(h_seq.size(), out.node_seq.size()), out.node_seq.begin());
    out.node_start.assign(h_start.begin(), h_start.end());
    if (out.node_start.empty()) out.node_start.push_back(0);
    out.node_len.assign(h_len.begin(), h_len.end());
    out.node_colour.assign(h_col.begin(), h_col.end());
    out.node_exts.assign(h_exts.begin(), h_exts.end());
    uint64_t cursor = nbases;
    for (const auto& c : cyc_nodes) {
        for (size_t j = 0; j < c.bases.size(); ++j) set_base(out.node_seq.data(), cursor + j, c.bases[j]);
        cursor += c.bases.size();
        out.node_start.push_back(cursor);
        out.node_len.push_back((uint32_t)c.bases.size());
        out.node_colour.push_back(c.colour);
        out.node_exts.push_back(c.exts);
    }
    stage.mark("cycles + assemble");
    return PA_OK;
}

}  // namespace

int build_graph_device(const uint64_t* packed, const uint64_t* tx_start, uint32_t num_tx, uint32_t k, int device, HostIndex& out) {
    if (k < PA_MIN_K || k > PA_MAX_K) return fail(PA_ERR_UNSUPPORTED, "k=%u outside [%u,%u]", k, PA_MIN_K, PA_MAX_K);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(PA_ERR_NO_DEVICE, "no HIP device: the GPU index builder needs one");
    if (device < 0 || device >= ndev) return fail(PA_ERR_INVALID_ARG, "device %d out of range (%d devices)", device, ndev);
    IB_HIP(hipSetDevice(device));
    try {
        return k <= 32 ? build_graph_device_t<uint64_t>(packed, tx_start, num_tx, k, out) : build_graph_device_t<u128>(packed, tx_start, num_tx, k, out);
    } catch (const std::bad_alloc&) {
        return fail(PA_ERR_OOM, "out of host memory while building the index");
    }
}

}  // namespace pa

using namespace pa;

// ---- C ABI (include/pseudoaligner_amd.h) ----
extern "C" {

int pa_host_index_build_packed_device(const uint64_t* packed, const uint64_t* tx_start, uint32_t num_tx, uint32_t k, int device,
                                      pa_host_index** out) {
    if (!packed || !tx_start || !out) return fail(PA_ERR_INVALID_ARG, "null argument");
    pa_host_index* h = new (std::nothrow) pa_host_index();
    if (!h) return fail(PA_ERR_OOM, "out of memory");
    try {
        int rc = build_graph_device(packed, tx_start, num_tx, k, device, h->h);
        if (rc != PA_OK) { delete h; return rc; }
        const uint64_t nb = tx_start[num_tx];
        h->h.tx_packed.assign(packed, packed + (nb + 31) / 32);
        h->h.tx_packed.push_back(0);
        h->h.tx_packed.push_back(0);
        h->h.tx_start.assign(tx_start, tx_start + num_tx + 1);
    } catch (const std::bad_alloc&) {
        delete h;
        return fail(PA_ERR_OOM, "out of memory while building the index");
    }
    *out = h;
    return PA_OK;
}

int pa_host_index_build_fasta_device(const char* fasta_path, uint32_t k, int device, pa_host_index** out) {
    if (!fasta_path || !out) return fail(PA_ERR_INVALID_ARG, "null argument");
    Txome t;
    int rc = read_fasta(fasta_path, t);
    if (rc != PA_OK) return rc;
    rc = pa_host_index_build_packed_device(t.packed.data(), t.tx_start.data(), t.num_tx(), k, device, out);
    if (rc != PA_OK) return rc;
    (*out)->h.tx_names = std::move(t.names);
    (*out)->h.tx_genes = std::move(t.genes);
    return PA_OK;
}

}  // extern "C"
