// pa_index_create's device side: the k-mer dictionary is filled and verified ON the GPU, from the chain blocks already
// resident in HBM — nothing of the 6.6 GB table (config 3) is built on the host or crosses PCIe.
//
// Replaces make_dbg_index (src/build_index.rs:182-221: boomphf MPHF + (node id, offset) scatter over every k-mer of every
// node):
//   pa_fill_insert_kernel   one thread per k-mer of the graph. k <= 32: two passes (dict_slots.hpp) — compare-and-swap on the handle
//                           word of the key's home slot; then the keys that did not get it take another slot of the bucket and
//                           flag the home slot. k > 32: compare-and-swap on the handle word of the first free entry of the line.
//                           The host flattener (device_flatten.cpp, kept for the CPU-only test tier) runs the same text
//   pa_fill_verify_kernel   every k-mer is looked up again: it must come back as (its block, its position) — a k-mer that
//                           occurs twice in the graph does not — within the 15 overflow buckets the mapping kernel follows
// (The edges — Node::r_edges / l_edges — are derived by the flattener, which needs them to merge nodes into chains.)
#include <hip/hip_runtime.h>
#include <cstdlib>

#include <algorithm>

#include "device_flatten.hpp"
#include "dict_slots.hpp"
#include "lane_steps.hpp"
#include "pa_common.hpp"

namespace pa {
namespace {

#define FILL_HIP(expr) do { const hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(PA_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)

__device__ __forceinline__ uint64_t win32(const uint64_t* w, uint32_t pos) {
    const uint32_t i = pos >> 5, s = (pos & 31) * 2;
    return s ? (w[i] >> s) | (w[i + 1] << (64 - s)) : w[i];
}

template <class KT> struct FillOps;
template <> struct FillOps<uint64_t> {
    static constexpr uint32_t SLOTS = SLOTS_PER_BUCKET;
    static constexpr double LOAD = DICT_LOAD;
    __host__ __device__ static uint64_t mask(uint32_t k) { return k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1); }
    __device__ static uint64_t get(const uint64_t* seq, uint32_t o, uint32_t k) { return win32(seq, o) & mask(k); }
    // (the k <= 32 dictionary is built and read by dict_slots.hpp: two insertion passes)
};
struct DeviceAtomics {
    __device__ static bool cas(uint32_t* p, uint32_t expect, uint32_t v) { return atomicCAS(p, expect, v) == expect; }
    __device__ static void and_(uint32_t* p, uint32_t m) { atomicAnd(p, m); }
};
template <> struct FillOps<u128> {
    static constexpr uint32_t SLOTS = 2;
    static constexpr double LOAD = 1.0 / 3.0;
    __host__ __device__ static u128 mask(uint32_t k) { return k >= 64 ? ~(u128)0 : (((u128)1 << (2 * k)) - 1); }
    __device__ static u128 get(const uint64_t* seq, uint32_t o, uint32_t k) {
        return ((u128)(win32(seq, o + 32) & FillOps<uint64_t>::mask(k - 32)) << 64) | win32(seq, o);
    }
    __device__ static uint32_t bucket(u128 km, uint32_t nbuckets) { return __umulhi((uint32_t)(pa_mix128((uint64_t)km, (uint64_t)(km >> 64)) >> 32), nbuckets); }
    __device__ static bool try_insert(uint32_t* line, u128 km, uint32_t handle, uint32_t off) {
        for (uint32_t i = 0; i < 2; ++i)
            if (atomicCAS(line + 8 * i + 4, NO_HANDLE, handle) == NO_HANDLE) {
                line[8 * i] = (uint32_t)km; line[8 * i + 1] = (uint32_t)(km >> 32);
                line[8 * i + 2] = (uint32_t)(km >> 64); line[8 * i + 3] = (uint32_t)(km >> 96);
                line[8 * i + 5] = off;
                return true;
            }
        return false;
    }
    __device__ static int look(const uint32_t* line, u128 km, uint32_t& handle, uint32_t& off) {
        bool full = true;
        for (uint32_t i = 0; i < 2; ++i) {
            if (line[8 * i + 4] == NO_HANDLE) { full = false; continue; }
            if (line[8 * i] == (uint32_t)km && line[8 * i + 1] == (uint32_t)(km >> 32) && line[8 * i + 2] == (uint32_t)(km >> 64) &&
                line[8 * i + 3] == (uint32_t)(km >> 96)) {
                handle = line[8 * i + 4];
                off = line[8 * i + 5];
                return 1;
            }
        }
        return full ? 2 : 0;
    }
};

template <class KT>
__device__ __forceinline__ bool dict_find(const uint32_t* table, uint32_t nbuckets, KT km, uint32_t& handle, uint32_t& off, uint32_t& probes);
template <>
__device__ __forceinline__ bool dict_find<uint64_t>(const uint32_t* table, uint32_t nbuckets, uint64_t km, uint32_t& handle, uint32_t& off, uint32_t& probes) {
    return dict_find64(table, nbuckets, km, handle, off, probes);
}
template <>
__device__ __forceinline__ bool dict_find<u128>(const uint32_t* table, uint32_t nbuckets, u128 km, uint32_t& handle, uint32_t& off, uint32_t& probes) {
    uint32_t b = FillOps<u128>::bucket(km, nbuckets);
    for (probes = 0; probes < nbuckets; ++probes) {
        const int r = FillOps<u128>::look(table + (uint64_t)b * BUCKET_WORDS, km, handle, off);
        if (r != 2) return r == 1;
        if (++b == nbuckets) b = 0;
    }
    return false;
}

// the g-th k-mer of the graph: its node i (kcum), offset o in it, chain position c = node_s[i] + o; the block it starts in
// holds all of it (k <= 64 bases from a window position < 64)
struct KmerAt {
    uint32_t node, block, off;   // block handle, second word of the dictionary entry (device_layout.hpp)
    const uint64_t* seq;         // the block's sequence words
    uint32_t rel;                // position of the k-mer's first base in them
};

// node of the g-th k-mer of the graph: the last i with kcum[i] <= g
__device__ __forceinline__ uint32_t node_of_kmer(const uint64_t* kcum, uint32_t num_nodes, uint64_t g) {
    uint32_t lo = 0, hi = num_nodes - 1;
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo + 1) / 2;
        if (kcum[mid] <= g) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__device__ __forceinline__ KmerAt kmer_at(const uint8_t* blobs, const uint32_t* handle, const uint32_t* node_s, const uint64_t* kcum, uint32_t num_nodes, uint64_t g, uint32_t k) {
    KmerAt a;
    a.node = node_of_kmer(kcum, num_nodes, g);
    const uint32_t o = (uint32_t)(g - kcum[a.node]), c = node_s[a.node] + o;
    a.block = handle[a.node] + (c >> CH_STRIDE_LOG2);
    a.off = dict_entry_off(c, o == 0, block_slot_of(blobs + (uint64_t)a.block * CH_BLOCK, (c & (CH_STRIDE - 1)) + k - 1));
    a.seq = reinterpret_cast<const uint64_t*>(blobs + (uint64_t)a.block * CH_BLOCK + CH_SEQ_BYTES);
    a.rel = c & (CH_STRIDE - 1);
    return a;
}

template <class KT>
__global__ __launch_bounds__(256) void pa_fill_insert_kernel(const uint8_t* __restrict__ blobs, const uint32_t* __restrict__ handle, const uint32_t* __restrict__ node_s,
                                                             const uint64_t* __restrict__ kcum, uint32_t num_nodes, uint64_t nk, uint32_t k, uint32_t* table,
                                                             uint32_t nbuckets, int pass) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nk) return;
    const KmerAt a = kmer_at(blobs, handle, node_s, kcum, num_nodes, g, k);
    const KT km = FillOps<KT>::get(a.seq, a.rel, k);
    if constexpr (sizeof(KT) == 8) {   // k <= 32: pass 0 = home slots, pass 1 = the keys that did not get theirs (dict_slots.hpp)
        if (pass == 0) dict_insert_home<DeviceAtomics>(table, nbuckets, km, a.block, a.off);
        else dict_insert_rest<DeviceAtomics>(table, nbuckets, km, a.block, a.off);
    } else {
        if (pass != 0) return;
        uint32_t b = FillOps<KT>::bucket(km, nbuckets);
        for (;;) {   // load <= 1/3: a free entry exists
            if (FillOps<KT>::try_insert(table + (uint64_t)b * BUCKET_WORDS, km, a.block, a.off)) return;
            if (++b == nbuckets) b = 0;
        }
    }
}

// flags[0] = a node one of whose k-mers does not come back as itself (a k-mer that occurs twice in the graph), else NO_HANDLE;
// flags[1] = 1 when some k-mer sits more than 15 buckets from home
template <class KT>
__global__ __launch_bounds__(256) void pa_fill_verify_kernel(const uint8_t* __restrict__ blobs, const uint32_t* __restrict__ handle, const uint32_t* __restrict__ node_s,
                                                             const uint64_t* __restrict__ kcum, uint32_t num_nodes, uint64_t nk, uint32_t k,
                                                             const uint32_t* __restrict__ table, uint32_t nbuckets, uint32_t* __restrict__ flags) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nk) return;
    const KmerAt a = kmer_at(blobs, handle, node_s, kcum, num_nodes, g, k);
    const KT km = FillOps<KT>::get(a.seq, a.rel, k);
    uint32_t fh = 0, fo = 0, probes = 0;
    if (!dict_find<KT>(table, nbuckets, km, fh, fo, probes) || fh != a.block || fo != a.off) atomicMin(flags, a.node);
    if (probes > DICT_MAX_PROBES) flags[1] = 1;
}

template <class KT>
int fill_t(const FlatDevice& fd, void* d_blobs, void** d_table, uint64_t* nbuckets_out) {
    const uint32_t N = fd.num_nodes, k = fd.k;
    const uint64_t nk = fd.num_kmers;
    *d_table = nullptr;
    void *d_handle = nullptr, *d_node_s = nullptr, *d_kcum = nullptr, *d_flags = nullptr;
    auto done = [&](int rc) {
        for (void* p : {d_handle, d_node_s, d_kcum, d_flags})
            if (p) (void)hipFree(p);
        if (rc != PA_OK && *d_table) { (void)hipFree(*d_table); *d_table = nullptr; }
        return rc;
    };
#define FILL_TRY(expr) do { const hipError_t e_ = (expr); if (e_ != hipSuccess) return done(fail(PA_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_))); } while (0)
    FILL_TRY(hipMalloc(&d_handle, (size_t)(N ? N : 1) * 4));
    FILL_TRY(hipMalloc(&d_node_s, (size_t)(N ? N : 1) * 4));
    FILL_TRY(hipMalloc(&d_kcum, ((size_t)N + 1) * 8));
    FILL_TRY(hipMalloc(&d_flags, 16));
    if (N) FILL_TRY(hipMemcpy(d_handle, fd.handle.data(), (size_t)N * 4, hipMemcpyHostToDevice));
    if (N) FILL_TRY(hipMemcpy(d_node_s, fd.node_s.data(), (size_t)N * 4, hipMemcpyHostToDevice));
    FILL_TRY(hipMemcpy(d_kcum, fd.node_kcum.data(), ((size_t)N + 1) * 8, hipMemcpyHostToDevice));
    uint64_t nbuckets = 0;
    double load0 = FillOps<KT>::LOAD;
    {   // a device short of memory gets the denser table (it costs 4 % of the mapping rate, device_layout.hpp, not the index)
        size_t free_b = 0, total_b = 0;
        const double dense = sizeof(KT) == 8 ? 0.5 : FillOps<KT>::LOAD;
        if (load0 < dense && hipMemGetInfo(&free_b, &total_b) == hipSuccess && (double)nk / (FillOps<KT>::SLOTS * load0) * BUCKET_WORDS * 4 > (double)free_b / 4) load0 = dense;
    }
    if (const char* v = knob_str("PA_DICT_LOAD")) { const double x = atof(v); if (x > 0.01 && x <= 0.95) load0 = x; }   // A/B runs only
    for (double load = load0;; load *= 0.75) {
        nbuckets = std::max<uint64_t>(1, (uint64_t)((double)nk / (FillOps<KT>::SLOTS * load)) + 1);
        if (nbuckets >= 0xFFFFFFFFull) return done(fail(PA_ERR_UNSUPPORTED, "dictionary exceeds 2^32 buckets"));
        if (*d_table) { (void)hipFree(*d_table); *d_table = nullptr; }
        const hipError_t em = hipMalloc(d_table, nbuckets * BUCKET_WORDS * 4);
        if (em != hipSuccess) { *d_table = nullptr; return done(fail(PA_ERR_OOM, "hipMalloc(%llu) for the dictionary: %s", (unsigned long long)(nbuckets * BUCKET_WORDS * 4), hipGetErrorString(em))); }
        FILL_TRY(hipMemsetAsync(*d_table, 0xFF, nbuckets * BUCKET_WORDS * 4, nullptr));   // empty slots, no flags (NO_HANDLE in every word)
        const uint32_t init[4] = {NO_HANDLE, 0u, NO_HANDLE, NO_HANDLE};
        FILL_TRY(hipMemcpy(d_flags, init, 16, hipMemcpyHostToDevice));
        if (nk) {
            const dim3 grid((uint32_t)((nk + 255) / 256));
            for (int pass = 0; pass < (sizeof(KT) == 8 ? 2 : 1); ++pass) {   // (stream order is the barrier between the passes)
                hipLaunchKernelGGL(pa_fill_insert_kernel<KT>, grid, dim3(256), 0, nullptr, static_cast<const uint8_t*>(d_blobs), static_cast<const uint32_t*>(d_handle),
                                   static_cast<const uint32_t*>(d_node_s), static_cast<const uint64_t*>(d_kcum), N, nk, k, static_cast<uint32_t*>(*d_table), (uint32_t)nbuckets, pass);
                FILL_TRY(hipGetLastError());
            }
            hipLaunchKernelGGL(pa_fill_verify_kernel<KT>, grid, dim3(256), 0, nullptr, static_cast<const uint8_t*>(d_blobs), static_cast<const uint32_t*>(d_handle),
                               static_cast<const uint32_t*>(d_node_s), static_cast<const uint64_t*>(d_kcum), N, nk, k, static_cast<const uint32_t*>(*d_table), (uint32_t)nbuckets,
                               static_cast<uint32_t*>(d_flags));
            FILL_TRY(hipGetLastError());
        }
        uint32_t flags[4];
        FILL_TRY(hipMemcpy(flags, d_flags, 16, hipMemcpyDeviceToHost));
        if (flags[0] != NO_HANDLE) return done(fail(PA_ERR_FORMAT, "a k-mer of node %u occurs twice in the graph", flags[0]));
        if (!flags[1]) break;   // else: some key sits further from home than the kernel follows; a larger table
    }
#undef FILL_TRY
    *nbuckets_out = nbuckets;
    return done(PA_OK);
}

}  // namespace

int device_fill_index(const FlatDevice& fd, void* d_blobs, void** d_table, uint64_t* nbuckets) {
    return fd.k <= 32 ? fill_t<uint64_t>(fd, d_blobs, d_table, nbuckets) : fill_t<u128>(fd, d_blobs, d_table, nbuckets);
}

}  // namespace pa
