// The multi-GPU side of the class-count table:
//   * pa_overflow — the per-GPU table of NOVEL classes: results that are no index class are counted in ONE slot of the dense
//     table (counts[num_classes]); which id sets they were is kept here, keyed by content, so that the reduction over GPUs
//     loses nothing: dense table -> all-reduce (sum), overflow tables -> all-gather + merge by content;
//   * pa_comm — an RCCL communicator (one rank per GPU, xGMI) owned by the library, so that a host without torch (the Rust
//     pipeline of north_star) can run the final reduce: pa_counts_allreduce, pa_overflow_allgather.
// RCCL is bound at run time (dlopen of librccl.so.1): the library loads, and maps reads, on a host without RCCL; the
// collective entry points then fail with PA_ERR_UNSUPPORTED and say why.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <vector>

#include "kernel_utils.hpp"
#include "kernels.hpp"
#include "pa_common.hpp"

using namespace pa;

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) return fail(PA_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));  \
    } while (0)

// ------------------------------------------------------------------------------------------------ overflow table
// Device layout: open addressing over `cap` slots (power of two):
//   keys[cap]   u64  0 = free, else the content hash of the id list with bit 63 forced
//   meta[cap]   {u32 pool_off, u32 len, u64 count, u32 ready, u32 pad} (24 bytes as 6 u32)
//   pool[pool_cap] u32 the id lists, ctl[0] = pool top, ctl[1] = status, ctl[2] = export cursor, ctl[3] = entries
// Two lanes that insert the same new class at the same moment may both create an entry (a reader can see the other's ids
// late); entries are merged by content at export time, so the exported table is exact either way.
struct pa_overflow {
    int device = 0;
    uint64_t cap = 0, pool_cap = 0;
    unsigned long long* d_keys = nullptr;
    uint32_t* d_meta = nullptr;
    uint32_t* d_pool = nullptr;
    unsigned long long* d_ctl = nullptr;   // [0] pool top, [1] status, [2] export cursor, [3] entries
    uint32_t* d_export = nullptr;          // serialised records (export_cap u32)
    uint64_t export_cap = 0;
    std::vector<uint32_t> h_export, h_merged;
    std::mutex mu;
};

namespace {

constexpr uint32_t OVF_META_WORDS = 6;
constexpr unsigned long long OVF_STATUS_TABLE_FULL = 1, OVF_STATUS_POOL_FULL = 2, OVF_STATUS_LIST_FULL = 4, OVF_STATUS_EXPORT_FULL = 8;

__global__ __launch_bounds__(256) void pa_overflow_insert_kernel(const uint32_t* __restrict__ novel, const unsigned long long* __restrict__ n_ptr,
                                                                 uint64_t novel_cap, const uint32_t* __restrict__ arena,
                                                                 unsigned long long* keys, uint32_t* meta, uint32_t* pool, unsigned long long* ctl,
                                                                 uint64_t cap, uint64_t pool_cap) {
    unsigned long long n = *n_ptr;
    if (n > novel_cap) n = novel_cap;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t off = novel[2 * i], len = novel[2 * i + 1];
        const uint32_t* ids = arena + off;
        const unsigned long long h = list_hash_dev(ids, len) | (1ull << 63);
        uint64_t slot = (h * 0x9e3779b97f4a7c15ull) >> 20 & (cap - 1);
        bool done = false;
        for (uint64_t probes = 0; !done && probes < cap;) {
            unsigned long long k = __hip_atomic_load(keys + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (k == 0) {
                k = atomicCAS(keys + slot, 0ull, h);
                if (k == 0) {   // this lane owns the new entry: ids into the pool, then publish
                    const unsigned long long po = atomicAdd(ctl + 0, (unsigned long long)len);
                    uint32_t* m = meta + slot * OVF_META_WORDS;
                    if (po + len > pool_cap) {
                        atomicOr(ctl + 1, OVF_STATUS_POOL_FULL);
                        m[0] = 0xFFFFFFFFu;
                    } else {
                        for (uint32_t t = 0; t < len; ++t) pool[po + t] = ids[t];
                        m[0] = (uint32_t)po;
                    }
                    m[1] = len;
                    m[2] = 1;
                    m[3] = 0;
                    atomicAdd(ctl + 3, 1ull);
                    __threadfence();
                    __hip_atomic_store(m + 4, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    done = true;
                    continue;
                }
            }
            if (k == h) {
                uint32_t* m = meta + slot * OVF_META_WORDS;
                if (__hip_atomic_load(m + 4, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0) continue;   // being written: look again
                const uint32_t po = __hip_atomic_load(m + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bool same = __hip_atomic_load(m + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == len && po != 0xFFFFFFFFu;
                for (uint32_t t = 0; same && t < len; ++t) same = __hip_atomic_load(pool + po + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ids[t];
                if (same) {
                    atomicAdd(reinterpret_cast<unsigned long long*>(m + 2), 1ull);
                    done = true;
                    continue;
                }
            }
            slot = (slot + 1) & (cap - 1);
            ++probes;
        }
        if (!done) atomicOr(ctl + 1, OVF_STATUS_TABLE_FULL);
    }
}

// serialised form (u32 words): [0] records, [1] words used incl. this header, then per record {len, count lo, count hi, ids[len]}
__global__ __launch_bounds__(256) void pa_overflow_export_kernel(const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ meta,
                                                                 const uint32_t* __restrict__ pool, unsigned long long* ctl, uint64_t cap,
                                                                 uint32_t* out, uint64_t out_cap) {
    const uint64_t slot = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= cap || keys[slot] == 0) return;
    const uint32_t* m = meta + slot * OVF_META_WORDS;
    const uint32_t po = m[0], len = m[1];
    if (po == 0xFFFFFFFFu) return;   // the pool was full: reported through the status word
    const unsigned long long pos = atomicAdd(ctl + 2, (unsigned long long)(3 + len));
    if (pos + 3 + len > out_cap) { atomicOr(ctl + 1, OVF_STATUS_EXPORT_FULL); return; }
    out[pos] = len;
    out[pos + 1] = m[2];
    out[pos + 2] = m[3];
    for (uint32_t t = 0; t < len; ++t) out[pos + 3 + t] = pool[po + t];
    atomicAdd(out, 1u);   // header word 0 = records (word 1 = words used, written by the host side of the export)
}

int merge_serialised(const uint32_t* const* bufs, const uint64_t* n_words, int nbufs, std::vector<uint32_t>& out) {
    std::map<std::vector<uint32_t>, unsigned long long> acc;   // ordered: the merged table is canonical (lexicographic by id list)
    for (int b = 0; b < nbufs; ++b) {
        const uint32_t* w = bufs[b];
        const uint64_t nw = n_words[b];
        if (nw == 0 || (nw >= 2 && w[0] == 0 && w[1] == 0)) continue;   // nothing, or nothing but padding
        if (nw < 2 || w[1] > nw || w[1] < 2) return fail(PA_ERR_FORMAT, "overflow buffer %d: bad header", b);
        uint64_t p = 2;
        for (uint32_t r = 0; r < w[0]; ++r) {
            if (p + 3 > w[1] || p + 3 + w[p] > w[1]) return fail(PA_ERR_FORMAT, "overflow buffer %d: record %u runs past the end", b, r);
            const uint32_t len = w[p];
            acc[std::vector<uint32_t>(w + p + 3, w + p + 3 + len)] += (unsigned long long)w[p + 1] | ((unsigned long long)w[p + 2] << 32);
            p += 3 + len;
        }
    }
    out.assign(2, 0);
    for (const auto& kv : acc) {
        out.push_back((uint32_t)kv.first.size());
        out.push_back((uint32_t)kv.second);
        out.push_back((uint32_t)(kv.second >> 32));
        out.insert(out.end(), kv.first.begin(), kv.first.end());
    }
    if (acc.size() > 0xFFFFFFFFull || out.size() > 0xFFFFFFFFull) return fail(PA_ERR_UNSUPPORTED, "merged overflow table too large");
    out[0] = (uint32_t)acc.size();
    out[1] = (uint32_t)out.size();
    return PA_OK;
}

// ------------------------------------------------------------------------------------------------ RCCL, bound at run time
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { ncclSum = 0, ncclMax = 2, ncclUint32 = 3, ncclUint64 = 5 };   // rccl.h: ncclRedOp_t / ncclDataType_t

struct Rccl {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.so) break;
        }
        if (!r.so) { r.why = std::string("librccl.so.1 not found: ") + (dlerror() ? dlerror() : "?"); return; }
        auto sym = [&](const char* n) { void* p = dlsym(r.so, n); if (!p && r.why.empty()) r.why = std::string("librccl lacks ") + n; return p; };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return r;
}

int rccl_ready() {
    Rccl& r = rccl();
    if (!r.why.empty()) return fail(PA_ERR_UNSUPPORTED, "RCCL is not usable: %s", r.why.c_str());
    return PA_OK;
}

#define NCCL_TRY(expr)                                                                                                  \
    do {                                                                                                                \
        ncclResult_t _r = (expr);                                                                                       \
        if (_r != 0) return fail(PA_ERR_HIP, "%s failed: %s", #expr, rccl().GetErrorString ? rccl().GetErrorString(_r) : "?"); \
    } while (0)

}  // namespace

struct pa_comm {
    int device = 0, nranks = 1, rank = 0;
    ncclComm_t comm = nullptr;
    unsigned long long* d_scalar = nullptr;   // two u64 for the size / status exchange of the overflow gather
};

// hooks for device_index.hip: what the map launch needs to know about an attached overflow table
namespace pa {

void overflow_launch_params(pa_overflow* o, MapParams& p) { p.novel_status = o->d_ctl + 1; }

// files the novel results a launch listed (per stream: device_index.hip) in the table; same stream, right behind the launch
int overflow_after_map(pa_overflow* o, const uint32_t* novel_list, const unsigned long long* novel_ctr, uint64_t novel_cap, const uint32_t* d_arena,
                       hipStream_t stream) {
    hipLaunchKernelGGL(pa_overflow_insert_kernel, dim3(1024), dim3(256), 0, stream, novel_list, novel_ctr, novel_cap, d_arena, o->d_keys, o->d_meta,
                       o->d_pool, o->d_ctl, o->cap, o->pool_cap);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(PA_ERR_HIP, "overflow insert launch: %s", hipGetErrorString(e));
    return PA_OK;
}

int overflow_device(const pa_overflow* o) { return o->device; }

}  // namespace pa

extern "C" {

int pa_overflow_create(int device, uint64_t max_classes, uint64_t max_ids, pa_overflow** out) {
    if (!out || max_classes == 0 || max_ids == 0) return fail(PA_ERR_INVALID_ARG, "bad argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(PA_ERR_NO_DEVICE, "no HIP device available; this library has no CPU fallback");
    if (device < 0 || device >= n) return fail(PA_ERR_INVALID_ARG, "device %d out of range (have %d)", device, n);
    if (max_ids > 0xFFFFFFF0ull) return fail(PA_ERR_UNSUPPORTED, "at most 2^32-16 ids in an overflow table");
    HIP_TRY(hipSetDevice(device));
    pa_overflow* o = new (std::nothrow) pa_overflow();
    if (!o) return fail(PA_ERR_OOM, "out of memory");
    o->device = device;
    uint64_t cap = 1024;
    while (cap < 2 * max_classes) cap <<= 1;   // load <= 0.5
    o->cap = cap;
    o->pool_cap = max_ids;
    o->export_cap = 2 + 3 * cap + max_ids;   // a record per table slot: whatever the table can hold can be exported (TABLE_FULL / POOL_FULL are the only limits)
    hipError_t e = hipMalloc(&o->d_keys, cap * 8);
    if (e == hipSuccess) e = hipMalloc(&o->d_meta, cap * OVF_META_WORDS * 4);
    if (e == hipSuccess) e = hipMalloc(&o->d_pool, max_ids * 4);
    if (e == hipSuccess) e = hipMalloc(&o->d_ctl, 64);
    if (e == hipSuccess) e = hipMalloc(&o->d_export, o->export_cap * 4);
    if (e == hipSuccess) e = hipMemset(o->d_keys, 0, cap * 8);
    if (e == hipSuccess) e = hipMemset(o->d_meta, 0, cap * OVF_META_WORDS * 4);
    if (e == hipSuccess) e = hipMemset(o->d_ctl, 0, 64);
    if (e != hipSuccess) {
        pa_overflow_destroy(o);
        return fail(PA_ERR_OOM, "overflow table (%llu slots, %llu ids): %s", (unsigned long long)cap, (unsigned long long)max_ids, hipGetErrorString(e));
    }
    *out = o;
    return PA_OK;
}

void pa_overflow_destroy(pa_overflow* o) {
    if (!o) return;
    (void)hipSetDevice(o->device);
    for (void* p : {(void*)o->d_keys, (void*)o->d_meta, (void*)o->d_pool, (void*)o->d_ctl, (void*)o->d_export})
        if (p) (void)hipFree(p);
    delete o;
}

int pa_overflow_reset(pa_overflow* o, void* stream) {
    if (!o) return fail(PA_ERR_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> g(o->mu);
    HIP_TRY(hipSetDevice(o->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemsetAsync(o->d_keys, 0, o->cap * 8, st));
    HIP_TRY(hipMemsetAsync(o->d_meta, 0, o->cap * OVF_META_WORDS * 4, st));
    HIP_TRY(hipMemsetAsync(o->d_ctl, 0, 64, st));
    return PA_OK;
}

// serialise the table on the device; *n_words = words used (header included). Leaves the records in o->d_export.
static int export_locked(pa_overflow* o, hipStream_t st, uint64_t* n_words) {
    HIP_TRY(hipSetDevice(o->device));
    HIP_TRY(hipMemsetAsync(o->d_export, 0, 8, st));
    const unsigned long long two = 2;
    HIP_TRY(hipMemcpyAsync(o->d_ctl + 2, &two, 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(pa_overflow_export_kernel, dim3((uint32_t)((o->cap + 255) / 256)), dim3(256), 0, st, o->d_keys, o->d_meta, o->d_pool, o->d_ctl,
                       o->cap, o->d_export, o->export_cap);
    HIP_TRY(hipGetLastError());
    unsigned long long ctl[4];
    HIP_TRY(hipMemcpyAsync(ctl, o->d_ctl, sizeof ctl, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (ctl[1] & OVF_STATUS_LIST_FULL) return fail(PA_ERR_ARENA_FULL, "overflow: more novel results in one launch than its list holds");
    if (ctl[1] & OVF_STATUS_TABLE_FULL) return fail(PA_ERR_ARENA_FULL, "overflow table full: more than %llu distinct novel classes", (unsigned long long)(o->cap / 2));
    if (ctl[1] & OVF_STATUS_POOL_FULL) return fail(PA_ERR_ARENA_FULL, "overflow id pool full: %llu ids needed, %llu available", ctl[0], (unsigned long long)o->pool_cap);
    if (ctl[1] & OVF_STATUS_EXPORT_FULL) return fail(PA_ERR_INTERNAL, "overflow export buffer too small");
    const uint32_t words = (uint32_t)ctl[2];
    HIP_TRY(hipMemcpyAsync(o->d_export + 1, &words, 4, hipMemcpyHostToDevice, st));   // header word 1 = words used
    HIP_TRY(hipStreamSynchronize(st));
    *n_words = ctl[2];
    return PA_OK;
}

int pa_overflow_fetch(pa_overflow* o, void* stream, const uint32_t** words, uint64_t* n_words) {
    if (!o || !words || !n_words) return fail(PA_ERR_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> g(o->mu);
    hipStream_t st = static_cast<hipStream_t>(stream);
    uint64_t nw = 0;
    const int rc = export_locked(o, st, &nw);
    if (rc != PA_OK) return rc;
    o->h_export.resize(nw);
    HIP_TRY(hipMemcpy(o->h_export.data(), o->d_export, nw * 4, hipMemcpyDeviceToHost));
    const uint32_t* one[1] = {o->h_export.data()};
    const int rc2 = merge_serialised(one, &nw, 1, o->h_merged);   // canonical order, duplicate entries folded
    if (rc2 != PA_OK) return rc2;
    *words = o->h_merged.data();
    *n_words = o->h_merged.size();
    return PA_OK;
}

int pa_overflow_merge(const uint32_t* const* bufs, const uint64_t* n_words, int nbufs, uint32_t* out, uint64_t out_cap, uint64_t* out_words) {
    if (nbufs < 0 || (nbufs && (!bufs || !n_words)) || !out_words) return fail(PA_ERR_INVALID_ARG, "null argument");
    std::vector<uint32_t> m;
    const int rc = merge_serialised(bufs, n_words, nbufs, m);
    if (rc != PA_OK) return rc;
    *out_words = m.size();
    if (m.size() > out_cap) return fail(PA_ERR_ARENA_FULL, "merged overflow table needs %zu words", m.size());
    if (out) memcpy(out, m.data(), m.size() * 4);
    return PA_OK;
}

// ---- communicator ----
int pa_comm_unique_id(uint8_t id[128]) {
    if (!id) return fail(PA_ERR_INVALID_ARG, "null argument");
    const int rc = rccl_ready();
    if (rc != PA_OK) return rc;
    ncclUniqueId u;
    NCCL_TRY(rccl().GetUniqueId(&u));
    memcpy(id, u.internal, 128);
    return PA_OK;
}

int pa_comm_create(int device, int nranks, int rank, const uint8_t id[128], pa_comm** out) {
    if (!id || !out || nranks < 1 || rank < 0 || rank >= nranks) return fail(PA_ERR_INVALID_ARG, "bad argument");
    const int rc = rccl_ready();
    if (rc != PA_OK) return rc;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(PA_ERR_NO_DEVICE, "no HIP device available; this library has no CPU fallback");
    if (device < 0 || device >= n) return fail(PA_ERR_INVALID_ARG, "device %d out of range (have %d)", device, n);
    HIP_TRY(hipSetDevice(device));
    pa_comm* c = new (std::nothrow) pa_comm();
    if (!c) return fail(PA_ERR_OOM, "out of memory");
    c->device = device;
    c->nranks = nranks;
    c->rank = rank;
    ncclUniqueId u;
    memcpy(u.internal, id, 128);
    ncclResult_t r = rccl().CommInitRank(&c->comm, nranks, u, rank);
    if (r != 0) { delete c; return fail(PA_ERR_HIP, "ncclCommInitRank(%d of %d): %s", rank, nranks, rccl().GetErrorString(r)); }
    if (hipMalloc(&c->d_scalar, 16) != hipSuccess) { rccl().CommDestroy(c->comm); delete c; return fail(PA_ERR_OOM, "hipMalloc"); }
    *out = c;
    return PA_OK;
}

void pa_comm_destroy(pa_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->comm) rccl().CommDestroy(c->comm);
    if (c->d_scalar) (void)hipFree(c->d_scalar);
    delete c;
}

int pa_comm_rank(const pa_comm* c) { return c ? c->rank : 0; }
int pa_comm_size(const pa_comm* c) { return c ? c->nranks : 1; }

int pa_counts_allreduce(pa_index* idx, uint64_t* d_counts, pa_comm* comm, void* stream) {
    if (!idx || !d_counts) return fail(PA_ERR_INVALID_ARG, "null argument");
    if (!comm) return PA_OK;   // one GPU: the local table is the global one
    HIP_TRY(hipSetDevice(comm->device));
    NCCL_TRY(rccl().AllReduce(d_counts, d_counts, (size_t)pa_counts_len(idx), ncclUint64, ncclSum, comm->comm, static_cast<hipStream_t>(stream)));
    return PA_OK;
}

int pa_overflow_allgather(pa_overflow* o, pa_comm* comm, void* stream, const uint32_t** words, uint64_t* n_words) {
    if (!o || !words || !n_words) return fail(PA_ERR_INVALID_ARG, "null argument");
    // no communicator = one GPU: the local table is the global one. A communicator of ONE rank deliberately takes the collective
    // path below (it costs a millisecond): it is the only way a one-GPU box — every test box — executes that code at all
    if (!comm) return pa_overflow_fetch(o, stream, words, n_words);
    std::lock_guard<std::mutex> g(o->mu);
    hipStream_t st = static_cast<hipStream_t>(stream);
    uint64_t nw = 0;
    // A rank-local failure (table / pool / novel list full: the documented PA_ERR_ARENA_FULL) must not keep this rank out of
    // the collectives the other ranks are about to enter — they would wait for it forever. Every rank always takes part in
    // the exchange of {words, status}; after it all of them know the largest table and whether ANY rank failed, and all
    // return the same error.
    int rc_local = export_locked(o, st, &nw);
    std::string why_local = rc_local != PA_OK ? last_error_ref() : std::string();
    // {words, status} -> the maximum over the ranks. A HIP call of THIS rank that fails on the way is folded into the status
    // this rank reports (and into rc_local), never returned before the all-reduce: the other ranks are about to enter it
    auto exchange = [&](unsigned long long (&mine)[2], unsigned long long (&most)[2]) {
        auto note = [&](hipError_t e, const char* what) {
            if (e == hipSuccess || rc_local != PA_OK) return;
            rc_local = PA_ERR_HIP;
            why_local = std::string(what) + ": " + hipGetErrorString(e);
            mine[0] = 0;
            mine[1] = (unsigned long long)(-PA_ERR_HIP);
        };
        note(hipSetDevice(comm->device), "hipSetDevice");
        note(hipMemcpyAsync(comm->d_scalar, mine, 16, hipMemcpyHostToDevice, st), "hipMemcpyAsync(status)");
        if (rc_local == PA_ERR_HIP) (void)hipMemcpyAsync(comm->d_scalar, mine, 16, hipMemcpyHostToDevice, st);   // (the status, if the device still takes it)
        const ncclResult_t r = rccl().AllReduce(comm->d_scalar, comm->d_scalar, 2, ncclUint64, ncclMax, comm->comm, st);
        if (r != 0 && rc_local == PA_OK) { rc_local = PA_ERR_HIP; why_local = std::string("ncclAllReduce: ") + rccl().GetErrorString(r); }
        note(hipMemcpyAsync(most, comm->d_scalar, 16, hipMemcpyDeviceToHost, st), "hipMemcpyAsync(result)");
        note(hipStreamSynchronize(st), "hipStreamSynchronize");
    };
    unsigned long long mine[2] = {rc_local == PA_OK ? nw : 0ull, (unsigned long long)(rc_local == PA_OK ? 0 : -rc_local)}, most[2] = {0, 0};
    exchange(mine, most);
    if (rc_local != PA_OK) return fail(rc_local, "%s", why_local.c_str());   // this rank's own reason
    if (most[1] != 0)      // some other rank failed: everybody reports it
        return fail(-(int)most[1], "overflow gather: another rank could not export its table (status %d there); no rank merged anything", -(int)most[1]);
    const uint64_t len = most[0];
    // send buffer of the common length (the largest table), zero padded: independent of this rank's own export capacity. A rank that
    // cannot allocate it says so in a second exchange, so that no rank enters the gather alone
    uint32_t *d_send = nullptr, *d_all = nullptr;
    hipError_t e = hipMalloc(&d_send, (size_t)len * 4);
    if (e == hipSuccess) e = hipMalloc(&d_all, (size_t)len * 4 * comm->nranks);
    if (e == hipSuccess) e = hipMemcpyAsync(d_send, o->d_export, (size_t)nw * 4, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess && len > nw) e = hipMemsetAsync(d_send + nw, 0, (size_t)(len - nw) * 4, st);
    if (e != hipSuccess) { rc_local = PA_ERR_OOM; why_local = std::string("overflow gather buffers: ") + hipGetErrorString(e); }
    unsigned long long mine2[2] = {0ull, (unsigned long long)(rc_local == PA_OK ? 0 : -rc_local)}, most2[2] = {0, 0};
    exchange(mine2, most2);
    if (rc_local != PA_OK || most2[1] != 0) {
        (void)hipFree(d_send);
        (void)hipFree(d_all);
        if (rc_local != PA_OK) return fail(rc_local, "%s (%llu words x %d ranks)", why_local.c_str(), (unsigned long long)len, comm->nranks);
        return fail(-(int)most2[1], "overflow gather: another rank could not allocate its gather buffers (status %d there); no rank merged anything", -(int)most2[1]);
    }
    ncclResult_t r = rccl().AllGather(d_send, d_all, (size_t)len, ncclUint32, comm->comm, st);
    if (r != 0) { (void)hipFree(d_send); (void)hipFree(d_all); return fail(PA_ERR_HIP, "ncclAllGather: %s", rccl().GetErrorString(r)); }
    std::vector<uint32_t> all((size_t)len * comm->nranks);
    e = hipMemcpyAsync(all.data(), d_all, all.size() * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d_send);
    (void)hipFree(d_all);
    if (e != hipSuccess) return fail(PA_ERR_HIP, "overflow gather copy: %s", hipGetErrorString(e));
    std::vector<const uint32_t*> bufs(comm->nranks);
    std::vector<uint64_t> sizes(comm->nranks, len);
    for (int i = 0; i < comm->nranks; ++i) bufs[i] = all.data() + (size_t)i * len;
    const int rc = merge_serialised(bufs.data(), sizes.data(), comm->nranks, o->h_merged);
    if (rc != PA_OK) return rc;
    *words = o->h_merged.data();
    *n_words = o->h_merged.size();
    return PA_OK;
}

}  // extern "C"
