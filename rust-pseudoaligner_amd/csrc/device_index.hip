// Device-side index object and the C ABI entry points that launch the HIP kernels.
// There is NO CPU fallback anywhere in this file: without a usable GPU every entry point fails with
// PA_ERR_NO_DEVICE / PA_ERR_HIP and a message in pa_last_error().
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

#include "device_flatten.hpp"
#include "kernels.hpp"
#include "lane_steps.hpp"
#include "pa_common.hpp"
#include "synth_common.hpp"

using namespace pa;

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) return fail(PA_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));  \
    } while (0)

namespace {

struct DevBuf {   // grow-only device scratch
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return PA_OK;
        if (p) { hipError_t e = hipFree(p); p = nullptr; bytes = 0; if (e != hipSuccess) return fail(PA_ERR_HIP, "hipFree: %s", hipGetErrorString(e)); }
        const size_t want = need + need / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return fail(PA_ERR_OOM, "hipMalloc(%zu): %s", want, hipGetErrorString(e)); }
        bytes = want;
        return PA_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

uint64_t list_hash_host(const uint32_t* v, uint32_t n) {   // must equal list_hash_dev (kernel_utils.hpp)
    uint64_t h = 0x243f6a8885a308d3ull ^ n;
    for (uint32_t i = 0; i < n; ++i) h = mix64(h ^ v[i]) + 0x9e3779b97f4a7c15ull;
    return h;
}

}  // namespace

struct LaunchCtx {
    std::mutex mu;
    DevBuf ctl;      // [0..7] arena_top (u64), [8..11] status, [12..15] tile counter, [16..] statistics, [448..455] novel results listed, [456..463] count keys handed out
    DevBuf spill, trace, novel;
    DevBuf keys, keys_sorted, keys_ctl;   // class-count launches: the waves' key streams (+ one key per deferred read), the keys partitioned by range, histograms / cursors (count_sort.hip)
    DevBuf defer;                          // reads whose class is looked up by content after the launch (resolve.hip): 32 bytes each, sized for every read
    uint32_t last_grid = 0;
    uint64_t last_arena_cap = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;   // before the map kernel / after it / after the resolve kernel / after the count kernels of the last launch (pa_index_set_timing)
    bool timed = false;
    void release() {
        for (DevBuf* b : {&ctl, &spill, &trace, &novel, &keys, &keys_sorted, &keys_ctl, &defer}) b->release();
        for (hipEvent_t e : {ev0, ev1, ev2, ev3})
            if (e) (void)hipEventDestroy(e);
        ev0 = ev1 = ev2 = ev3 = nullptr;
        timed = false;
    }
};

struct pa_index {
    int device = 0;
    int num_cus = 0;
    DevIndexView dv{};
    void *d_table = nullptr, *d_blobs = nullptr, *d_ledge = nullptr, *d_seg_g = nullptr, *d_seg_nid = nullptr, *d_ec = nullptr, *d_class_ref = nullptr, *d_class_len = nullptr,
         *d_class_table = nullptr, *d_wtable = nullptr;
    uint64_t class_table_size = 0;
    pa_index_stats stats{};
    // per-launch scratch: one context per stream the caller launches on, so that launches on different streams (from one or
    // several host threads) run concurrently; launches on ONE stream share a context and are ordered by the stream
    std::mutex mu;                // guards `ctxs` and `ovf`
    std::map<hipStream_t, std::shared_ptr<LaunchCtx>> ctxs;   // shared: a launch that looked its context up keeps it alive across pa_index_release_stream
    pa_overflow* ovf = nullptr;   // attached overflow table of novel classes (collective.hip), not owned
    bool timing = false;          // pa_index_set_timing: HIP events around the map kernel of every launch
    std::mutex hmu;               // the host-buffer convenience path (b_* below) is one batch at a time
    // host-buffer convenience path
    DevBuf b_ascii, b_offsets, b_tiles, b_lens, b_results, b_arena, b_colour, b_nodes, b_nodes_len;
    std::vector<uint32_t> h_class_ids;
    std::vector<uint32_t> h_ec, h_class_ref, h_class_len;
    // every index class rendered once as the reference prints it between the brackets ("1, 5, 9"): text of class c =
    // h_class_text[h_class_text_off[c] .. h_class_text_off[c + 1]). Built on first use by the ingest pipelines (ingest.hpp): a read
    // whose class comes back by reference then costs one copy instead of a table walk and a decimal conversion per id.
    std::once_flag class_text_once;
    std::vector<uint64_t> h_class_text_off;
    std::vector<char> h_class_text;
    std::vector<uint32_t> h_arena;
    void *d_class_text_off = nullptr, *d_class_text = nullptr;   // device copy of the rendered classes (uploaded on first use, under `mu`)
    // parked by fastq.cpp / record_stream.cpp between calls (guarded by `mu`): the buffer sets of up to four lanes (pa_process_reads_multi with the
    // handle listed several times, concurrent callers)
    std::vector<std::pair<void*, void (*)(void*)>> ingest_caches;
    std::vector<std::pair<void*, void (*)(void*)>> host_pipes;   // ... and of pa_map_tiles_host (host_batch.cpp): streams + staging buffers of the chunks in flight
};

extern "C" {

int pa_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int use_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(PA_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(PA_ERR_INVALID_ARG, "device %d out of range (have %d)", device, n);
    HIP_TRY(hipSetDevice(device));
    return PA_OK;
}

static int upload(const void* src, size_t bytes, void** dst) {
    HIP_TRY(hipMalloc(dst, bytes ? bytes : 16));
    if (bytes) HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return PA_OK;
}

extern "C++" {
namespace pa {
void index_host_classes(const pa_index* idx, const uint32_t** ec, const uint32_t** class_ref, int* device) {
    *ec = idx->h_ec.data();
    *class_ref = idx->h_class_ref.data();
    *device = idx->device;
}
void index_host_class_text(pa_index* idx, const uint64_t** off, const char** text) {
    std::call_once(idx->class_text_once, [idx] {
        const uint32_t nc = idx->stats.num_classes;
        const uint32_t* ec = idx->h_ec.data();
        const uint32_t* cref = idx->h_class_ref.data();
        auto digits = [](uint32_t v) { uint32_t d = 1; while (v >= 10) { v /= 10; ++d; } return d; };
        std::vector<uint64_t>& off_v = idx->h_class_text_off;
        off_v.assign((size_t)nc + 1, 0);
        const int T = std::max(1, std::min(16, usable_threads()));
        const uint32_t* clen = idx->h_class_len.data();
        auto ids_of = [&](uint32_t c, const uint32_t*& ids, uint32_t& n) {   // (the length the flattener noted: nothing is derived from where the next record lies)
            ids = ec + 4ull * cref[c] + 1;
            n = clen[c];
        };
        {
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t] {
                    for (uint64_t c = (uint64_t)nc * t / T; c < (uint64_t)nc * (t + 1) / T; ++c) {
                        const uint32_t* ids; uint32_t n;
                        ids_of((uint32_t)c, ids, n);
                        uint64_t len = n ? 2ull * (n - 1) : 0;
                        for (uint32_t j = 0; j < n; ++j) len += digits(ids[j]);
                        off_v[c + 1] = len;
                    }
                });
            for (auto& x : th) x.join();
        }
        for (uint32_t c = 0; c < nc; ++c) off_v[c + 1] += off_v[c];
        idx->h_class_text.resize(off_v[nc] + 16);
        {
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t] {
                    for (uint64_t c = (uint64_t)nc * t / T; c < (uint64_t)nc * (t + 1) / T; ++c) {
                        const uint32_t* ids; uint32_t n;
                        ids_of((uint32_t)c, ids, n);
                        char* o = idx->h_class_text.data() + off_v[c];
                        for (uint32_t j = 0; j < n; ++j) {
                            if (j) { *o++ = ','; *o++ = ' '; }
                            char b[10]; int k = 10; uint32_t v = ids[j];
                            do { b[--k] = (char)('0' + v % 10); v /= 10; } while (v);
                            memcpy(o, b + k, (size_t)(10 - k)); o += 10 - k;
                        }
                    }
                });
            for (auto& x : th) x.join();
        }
    });
    *off = idx->h_class_text_off.data();
    *text = idx->h_class_text.data();
}
int index_device_class_text(pa_index* idx, const uint64_t** d_off, const uint8_t** d_text) {
    const uint64_t* off = nullptr;
    const char* txt = nullptr;
    index_host_class_text(idx, &off, &txt);
    std::lock_guard<std::mutex> g(idx->mu);
    if (!idx->d_class_text) {
        if (hipSetDevice(idx->device) != hipSuccess) return fail(PA_ERR_HIP, "hipSetDevice failed");
        void *a = nullptr, *b = nullptr;
        const size_t nb = idx->h_class_text.size(), no = idx->h_class_text_off.size() * 8;
        if (hipMalloc(&a, no ? no : 16) != hipSuccess || hipMalloc(&b, nb ? nb : 16) != hipSuccess) { if (a) (void)hipFree(a); return fail(PA_ERR_OOM, "hipMalloc for the rendered class table"); }
        if (hipMemcpy(a, off, no, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(b, txt, nb, hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(a); (void)hipFree(b);
            return fail(PA_ERR_HIP, "upload of the rendered class table failed");
        }
        idx->d_class_text_off = a;
        idx->d_class_text = b;
    }
    *d_off = static_cast<const uint64_t*>(idx->d_class_text_off);
    *d_text = static_cast<const uint8_t*>(idx->d_class_text);
    return PA_OK;
}
void* index_take_ingest_cache(pa_index* idx) {
    std::lock_guard<std::mutex> g(idx->mu);
    if (idx->ingest_caches.empty()) return nullptr;
    void* c = idx->ingest_caches.back().first;
    idx->ingest_caches.pop_back();
    return c;
}
void* index_take_host_pipe(pa_index* idx) {
    std::lock_guard<std::mutex> g(idx->mu);
    if (idx->host_pipes.empty()) return nullptr;
    void* c = idx->host_pipes.back().first;
    idx->host_pipes.pop_back();
    return c;
}
void index_put_host_pipe(pa_index* idx, void* pipe, void (*free_fn)(void*)) {
    {
        std::lock_guard<std::mutex> g(idx->mu);
        if (idx->host_pipes.size() < 2) {
            idx->host_pipes.emplace_back(pipe, free_fn);
            return;
        }
    }
    free_fn(pipe);
}
void index_put_ingest_cache(pa_index* idx, void* cache, void (*free_fn)(void*)) {
    {
        std::lock_guard<std::mutex> g(idx->mu);
        if (idx->ingest_caches.size() < 4) {
            idx->ingest_caches.emplace_back(cache, free_fn);
            return;
        }
    }
    free_fn(cache);
}
}  // namespace pa
}

void pa_index_destroy(pa_index* idx) {
    if (!idx) return;
    (void)hipSetDevice(idx->device);
    for (void* p : {idx->d_table, idx->d_blobs, idx->d_ledge, idx->d_seg_g, idx->d_seg_nid, idx->d_ec, idx->d_class_ref, idx->d_class_len, idx->d_class_table, idx->d_wtable,
                    idx->d_class_text_off, idx->d_class_text})
        if (p) (void)hipFree(p);
    {   // (releases their streams' contexts)
        std::vector<std::pair<void*, void (*)(void*)>> parked;
        { std::lock_guard<std::mutex> g(idx->mu); parked.swap(idx->ingest_caches); parked.insert(parked.end(), idx->host_pipes.begin(), idx->host_pipes.end()); idx->host_pipes.clear(); }
        for (auto& c : parked) c.second(c.first);
    }
    for (auto& kv : idx->ctxs) kv.second->release();
    for (DevBuf* b : {&idx->b_ascii, &idx->b_offsets, &idx->b_tiles, &idx->b_lens, &idx->b_results,
                      &idx->b_arena, &idx->b_colour, &idx->b_nodes, &idx->b_nodes_len})
        b->release();
    delete idx;
}

int pa_index_create(const pa_flat_index* flat, int device, pa_index** out) {
    if (!flat || !out) return fail(PA_ERR_INVALID_ARG, "null argument");
    int rc = use_device(device);
    if (rc != PA_OK) return rc;
    FlatDevice fd;
    int threads = usable_threads();
    if (threads < 1) threads = 1;
    // classes, window table, edges, chains and their blocks on the host; the dictionary (6.6 GB at config 3) is built on the GPU
    // from the uploaded blocks (index_fill.hip) — nothing of the table exists on the host or crosses PCIe
    rc = flatten_for_device(*flat, threads, fd, /*device_dict=*/true);
    if (rc != PA_OK) return rc;

    // class-list hash table for the count kernel: open addressing of class ids keyed by the hash of the id list
    std::vector<uint32_t> ctab((size_t)fd.num_classes * 2 + 16, 0xFFFFFFFFu);
    for (uint32_t c = 0; c < fd.num_classes; ++c) {
        uint64_t j = list_hash_host(fd.ec.data() + 4ull * fd.class_ref[c] + 1, fd.class_len[c]) % ctab.size();
        while (ctab[j] != 0xFFFFFFFFu)
            if (++j == ctab.size()) j = 0;
        ctab[j] = c;
    }

    pa_index* idx = new (std::nothrow) pa_index();
    if (!idx) return fail(PA_ERR_OOM, "out of memory");
    idx->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) idx->num_cus = prop.multiProcessorCount;
    if (idx->num_cus <= 0) idx->num_cus = 256;
    rc = upload(fd.blobs.data(), fd.blobs.size(), &idx->d_blobs);
    if (rc == PA_OK) rc = upload(fd.ledge.data(), fd.ledge.size() * 4, &idx->d_ledge);
    if (rc == PA_OK) rc = device_fill_index(fd, idx->d_blobs, &idx->d_table, &fd.nbuckets);
    if (rc == PA_OK) rc = upload(fd.seg_g.data(), fd.seg_g.size() * 8, &idx->d_seg_g);
    if (rc == PA_OK) rc = upload(fd.seg_nid.data(), fd.seg_nid.size() * 4, &idx->d_seg_nid);
    if (rc == PA_OK) rc = upload(fd.ec.data(), fd.ec.size() * 4, &idx->d_ec);
    if (rc == PA_OK) rc = upload(fd.class_ref.data(), fd.class_ref.size() * 4, &idx->d_class_ref);
    if (rc == PA_OK) rc = upload(fd.class_len.data(), fd.class_len.size() * 4, &idx->d_class_len);
    if (rc == PA_OK) rc = upload(ctab.data(), ctab.size() * 4, &idx->d_class_table);
    if (rc == PA_OK) rc = upload(fd.wtable.data(), fd.wtable.size() * 4, &idx->d_wtable);
    if (rc != PA_OK) { pa_index_destroy(idx); return rc; }
    idx->class_table_size = ctab.size();
    idx->dv = fd.host_view();
    idx->dv.table = static_cast<const uint32_t*>(idx->d_table);
    idx->dv.blobs = static_cast<const uint8_t*>(idx->d_blobs);
    idx->dv.ledge = static_cast<const uint32_t*>(idx->d_ledge);
    idx->dv.seg_g = static_cast<const uint64_t*>(idx->d_seg_g);
    idx->dv.seg_nid = static_cast<const uint32_t*>(idx->d_seg_nid);
    idx->h_ec = fd.ec;   // host copy of the class table: pa_map_batch resolves by-reference classes from it
    idx->h_class_ref = fd.class_ref;
    idx->h_class_len = fd.class_len;
    idx->dv.ec = static_cast<const uint32_t*>(idx->d_ec);
    idx->dv.class_ref = static_cast<const uint32_t*>(idx->d_class_ref);
    idx->dv.class_len = static_cast<const uint32_t*>(idx->d_class_len);
    idx->dv.wtable = static_cast<const uint32_t*>(idx->d_wtable);
    // a dictionary beyond the Infinity Cache (256 MB) is streamed past the caches (ld_stream): every line of it is used once per probe,
    // and left to itself it evicts the chain blocks, which every read comes back to
    idx->dv.stream_nt = (fd.k <= 32 && fd.nbuckets * BUCKET_WORDS * 4 > (512ull << 20)) ? 1u : 0u;   // (the two-word dictionary's probe is four loads of one line: slower with the hint)
    pa_index_stats& s = idx->stats;
    s.num_kmers = fd.num_kmers;
    s.table_slots = fd.nbuckets * SLOTS_PER_BUCKET;
    s.bytes_table = fd.nbuckets * BUCKET_WORDS * 4;
    s.bytes_graph = fd.blobs.size() + fd.ledge.size() * 4 + fd.seg_g.size() * 12;
    s.bytes_classes = (fd.ec.size() + fd.class_ref.size() + fd.class_len.size() + ctab.size() + fd.wtable.size()) * 4;
    s.bytes_total = s.bytes_table + s.bytes_graph + s.bytes_classes;
    s.num_nodes = fd.num_nodes;
    s.num_classes = fd.num_classes;
    s.k = fd.k;
    s.max_class_len = fd.max_class_len;
    *out = idx;
    return PA_OK;
}

int pa_index_create_multi(const pa_flat_index* flat, const int* devices, int ndev, pa_index** out) {
    if (!flat || !devices || !out || ndev < 1) return fail(PA_ERR_INVALID_ARG, "null argument or ndev < 1");
    for (int i = 0; i < ndev; ++i) out[i] = nullptr;
    for (int i = 0; i < ndev; ++i) {
        for (int j = 0; j < i; ++j)
            if (devices[j] == devices[i]) {
                for (int t = 0; t < i; ++t) { pa_index_destroy(out[t]); out[t] = nullptr; }
                return fail(PA_ERR_INVALID_ARG, "device %d listed twice", devices[i]);
            }
        const int rc = pa_index_create(flat, devices[i], &out[i]);
        if (rc != PA_OK) {
            const std::string why = last_error_ref();
            for (int t = 0; t <= i; ++t) { pa_index_destroy(out[t]); out[t] = nullptr; }
            return fail(rc, "device %d: %s", devices[i], why.c_str());
        }
    }
    return PA_OK;
}

int pa_index_get_stats(const pa_index* idx, pa_index_stats* stats) {
    if (!idx || !stats) return fail(PA_ERR_INVALID_ARG, "null argument");
    *stats = idx->stats;
    return PA_OK;
}

uint32_t pa_words_per_read(uint32_t max_read_len) { return (max_read_len + 31) / 32 ? (max_read_len + 31) / 32 : 1; }
size_t pa_tiles_words(uint64_t n_reads, uint32_t words_per_read) { return (size_t)((n_reads + 63) / 64) * words_per_read * 64; }

int pa_encode_reads_device(const pa_index* idx, const uint8_t* d_ascii, const uint64_t* d_offsets, uint64_t n_reads,
                           uint32_t words_per_read, uint64_t* d_tiles, uint32_t* d_lens, void* stream) {
    if (!idx || !d_ascii || !d_offsets || !d_tiles || !d_lens || words_per_read == 0) return fail(PA_ERR_INVALID_ARG, "null argument");
    HIP_TRY(hipSetDevice(idx->device));
    const int e = launch_encode(d_ascii, d_offsets, n_reads, words_per_read, d_tiles, d_lens, static_cast<hipStream_t>(stream));
    if (e) return fail(PA_ERR_HIP, "encode launch: %s", hipGetErrorString((hipError_t)e));
    return PA_OK;
}

// ---- launch geometry ----
static int env_int(const char* name, int dflt) { return knob_int(name, dflt); }   // A/B knobs: -DPA_DEBUG_KNOBS builds only (pa_common.hpp)

// u32 words of a slot's row in the spill / trace scratch: list-mode header + (ref, len, class id, -) quads for >= 2 * max
// read length + 2 node visits; also the stride of the node lists of pa_map_batch_nodes
static uint32_t spill_cap_of(uint32_t wpr) { return 256 * wpr + 24; }

// pooled kernel: slots per wave such that `per_cu` workgroups share the 160 KiB of LDS of a CU
static int pool_geometry(pa_index* idx, uint64_t n_reads, uint32_t wpr, uint32_t* grid, size_t* lds, uint32_t* slots) {
    const size_t cu_lds = 160 * 1024 - 4096;   // LDS is allocated in granules: a pool sized to the last byte loses a whole workgroup per CU
    int per_cu = env_int("PA_MAP_BLOCKS_PER_CU", 3);
    uint32_t S = 0;
    for (; per_cu >= 1; --per_cu) {
        const size_t per_wave = cu_lds / (size_t)per_cu / (PA_MAP_BLOCK / 64);
        const size_t per_slot = pool_slot_bytes(wpr);
        if (per_wave < pool_fixed_bytes() + 64 * per_slot + 16) continue;
        S = (uint32_t)((per_wave - pool_fixed_bytes() - 16) / per_slot);
        break;
    }
    if (S < 64) return fail(PA_ERR_UNSUPPORTED, "reads of %u words do not fit the LDS of a compute unit", wpr);
    if (S > pool_max_slots()) S = pool_max_slots();
    const int want = env_int("PA_POOL_SLOTS", 0);
    if (want >= 64 && (uint32_t)want <= S) S = (uint32_t)want;
    *slots = S;
    *lds = pool_lds_bytes(wpr, S);
    int occ = 0;
    if (pool_kernel_occupancy(*lds, &occ) == 0 && occ >= 1 && occ < per_cu) per_cu = occ;
    const uint64_t ntiles = (n_reads + 63) / 64;
    const uint64_t waves_wanted = (ntiles + 7) / 8;                      // >= 8 tiles per wave when the batch allows
    uint64_t blocks = (waves_wanted + PA_MAP_BLOCK / 64 - 1) / (PA_MAP_BLOCK / 64);
    const uint64_t cap = (uint64_t)idx->num_cus * per_cu;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    *grid = (uint32_t)blocks;
    return PA_OK;
}

// the launch context of a stream (created on first use)
static int ctx_of(pa_index* idx, hipStream_t stream, std::shared_ptr<LaunchCtx>* out) {
    std::lock_guard<std::mutex> g(idx->mu);
    auto it = idx->ctxs.find(stream);
    if (it == idx->ctxs.end()) {
        std::shared_ptr<LaunchCtx> c(new (std::nothrow) LaunchCtx());
        if (!c) return fail(PA_ERR_OOM, "out of memory");
        const int rc = c->ctl.ensure(1024);
        if (rc != PA_OK) return rc;
        it = idx->ctxs.emplace(stream, std::move(c)).first;
    }
    *out = it->second;
    return PA_OK;
}

static int map_launch_locked(pa_index* idx, LaunchCtx* cx, const uint64_t* d_tiles, const uint32_t* d_lens, uint64_t n_reads, uint32_t wpr,
                             uint32_t allowed, pa_read_result* d_results, uint32_t* d_arena, uint64_t arena_cap, uint32_t* d_colour,
                             uint64_t* d_counts, uint32_t* d_nodes, uint32_t* d_nodes_len, hipStream_t stream, uint32_t uniform_len = 0) {
    if (n_reads >= 0xFFFFFFFFull) return fail(PA_ERR_UNSUPPORTED, "at most 2^32-2 reads per batch");
    if (wpr == 0 || wpr > (PA_MAX_READ_LEN + 31) / 32) return fail(PA_ERR_UNSUPPORTED, "words_per_read %u outside [1,%u]", wpr, (PA_MAX_READ_LEN + 31) / 32);
    uint32_t grid = 0;
    size_t lds = 0;
    uint32_t slots = 64;
    int rc = pool_geometry(idx, n_reads, wpr, &grid, &lds, &slots);
    if (rc != PA_OK) return rc;
    const uint32_t spill_cap = spill_cap_of(wpr);
    {   // long reads: rows of many KB per slot; fewer workgroups keep the scratch at a few GB (throughput of such batches is not the point)
        const size_t per_block = (size_t)(PA_MAP_BLOCK / 64) * slots * spill_cap * 4 * (d_nodes ? 2 : 1);
        const size_t budget = (size_t)6 << 30;
        if ((size_t)grid * per_block > budget) grid = (uint32_t)std::max<size_t>(1, budget / per_block);
    }
    const size_t lanes = (size_t)grid * (PA_MAP_BLOCK / 64) * slots;
    rc = cx->spill.ensure(lanes * spill_cap * 4);
    if (rc != PA_OK) return rc;
    if (d_nodes) { rc = cx->trace.ensure(lanes * spill_cap * 4); if (rc != PA_OK) return rc; }
    HIP_TRY(hipMemsetAsync(cx->ctl.p, 0, 512, stream));
    MapParams p{};
    p.ix = idx->dv;
    p.tiles = d_tiles;
    p.lens = d_lens;
    p.uniform_len = uniform_len;   // (d_lens == nullptr: every read has this many bases)
    p.n_reads = n_reads;
    p.wpr = wpr;
    p.allowed = allowed;
    p.results = d_results;
    p.arena = d_arena;
    // bit 31 of class_off means "class by reference" (PA_CLASS_REF): offsets handed out by the arena must stay below 2^31
    p.arena_cap = arena_cap > PA_MAX_ARENA_ENTRIES ? PA_MAX_ARENA_ENTRIES : arena_cap;
    p.colour_out = d_colour;
    p.arena_top = cx->ctl.as<unsigned long long>();
    p.status = cx->ctl.as<uint32_t>() + 2;
    p.tile_ctr = cx->ctl.as<uint32_t>() + 3;
    p.spill = cx->spill.as<uint32_t>();
    p.spill_cap = spill_cap;
    const uint64_t counts_len = (uint64_t)idx->stats.num_classes + 3;
    // reads whose class has to be looked up by content are resolved after the launch (resolve.hip): 32 bytes per read in the worst case
    const uint64_t defer_cap = defer_capacity(n_reads, grid * (PA_MAP_BLOCK / 64));
    if ((rc = cx->defer.ensure(defer_cap * 32))) return rc;
    p.defer = cx->defer.as<uint32_t>();
    p.defer_top = cx->ctl.as<unsigned long long>() + 58;
    p.defer_cap = defer_cap;
    p.keys_cap = 0;
    uint64_t keys_cap = 0;
    if (d_counts) {   // the waves' key streams (4.3 bytes per read), one key per deferred read behind them, and the keys partitioned by range (4 bytes per read)
        keys_cap = key_stream_capacity(n_reads, grid * (PA_MAP_BLOCK / 64));
        if ((rc = cx->keys.ensure((keys_cap + defer_cap) * 4)) || (rc = cx->keys_sorted.ensure((n_reads + 64) * 4)) || (rc = cx->keys_ctl.ensure(count_keys_ctl_bytes(counts_len)))) return rc;
        p.keys = cx->keys.as<uint32_t>();
        p.keys_top = cx->ctl.as<unsigned long long>() + 57;
        p.keys_cap = keys_cap;
        p.counts = reinterpret_cast<unsigned long long*>(d_counts);
    }
    p.class_table = static_cast<const uint32_t*>(idx->d_class_table);
    p.class_table_size = idx->class_table_size;
    p.pool_slots = slots;
    p.dbg = env_int("PA_MAP_STATS", 0) ? cx->ctl.as<unsigned long long>() + 2 : nullptr;
    p.ablate = (uint32_t)env_int("PA_MAP_ABLATE", 0);
    pa_overflow* ovf = nullptr;
    { std::lock_guard<std::mutex> g(idx->mu); ovf = idx->ovf; }
    if (d_counts && ovf) {   // novel results of this launch are listed (per stream) for the overflow table
        const uint64_t want = n_reads + 64;   // every read can end in a novel class: the list never overflows
        rc = cx->novel.ensure(want * 8);
        if (rc != PA_OK) return rc;
        p.novel_list = cx->novel.as<uint32_t>();
        p.novel_ctr = cx->ctl.as<unsigned long long>() + 56;
        p.novel_cap = want;
        overflow_launch_params(ovf, p);
    }
    p.trace = d_nodes ? cx->trace.as<uint32_t>() : nullptr;
    p.nodes_out = d_nodes;
    p.nodes_len = d_nodes_len;
    cx->last_grid = grid;
    cx->last_arena_cap = p.arena_cap;
    cx->timed = false;
    if (n_reads == 0) return PA_OK;
    bool timing = false;
    { std::lock_guard<std::mutex> g(idx->mu); timing = idx->timing; }
    if (timing) {
        if (!cx->ev0) { HIP_TRY(hipEventCreate(&cx->ev0)); HIP_TRY(hipEventCreate(&cx->ev1)); HIP_TRY(hipEventCreate(&cx->ev2)); HIP_TRY(hipEventCreate(&cx->ev3)); }
        HIP_TRY(hipEventRecord(cx->ev0, stream));
    }
    const int e = launch_map_pool(p, grid, lds, stream);
    if (e) return fail(PA_ERR_HIP, "map launch (grid %u, lds %zu): %s", grid, lds, hipGetErrorString((hipError_t)e));
    if (timing) { HIP_TRY(hipEventRecord(cx->ev1, stream)); cx->timed = true; }
    {
        const int e1 = launch_resolve(p, defer_cap, keys_cap, idx->num_cus, stream);
        if (e1) return fail(PA_ERR_HIP, "resolve launch: %s", hipGetErrorString((hipError_t)e1));
    }
    if (timing) HIP_TRY(hipEventRecord(cx->ev2, stream));
    if (d_counts) {
        const int e2 = launch_count_keys(p.keys, p.keys_top, keys_cap, p.defer_top, defer_cap, cx->keys_sorted.as<uint32_t>(), cx->keys_ctl.as<uint32_t>(),
                                         reinterpret_cast<unsigned long long*>(d_counts), counts_len, idx->num_cus, stream, n_reads);
        if (e2) return fail(PA_ERR_HIP, "count launch: %s", hipGetErrorString((hipError_t)e2));
        if (ovf) {
            rc = overflow_after_map(ovf, p.novel_list, p.novel_ctr, p.novel_cap, d_arena, stream);
            if (rc != PA_OK) return rc;
        }
    }
    if (timing) HIP_TRY(hipEventRecord(cx->ev3, stream));
    return PA_OK;
}

static int map_finish_locked(pa_index* idx, LaunchCtx* cx, hipStream_t stream, uint64_t* arena_used, uint64_t* arena_needed) {
    // (the control block comes back ON the launch's stream: a plain hipMemcpy runs on the null stream and waits for every other stream
    // of the process — a caller that copies its next batch in on another stream meanwhile would find this call waiting for that copy)
    struct { unsigned long long top; uint32_t status; uint32_t pad; } ctl;
    HIP_TRY(hipMemcpyAsync(&ctl, cx->ctl.p, 16, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (env_int("PA_MAP_STATS", 0)) {
        constexpr uint32_t NS = ST_COUNT + 4;   // ST_NSTAT of map_pool.hip: one entry per state, the dual iterations, the forward step in three parts
        unsigned long long d[3 * NS];
        HIP_TRY(hipMemcpy(d, cx->ctl.as<unsigned long long>() + 2, sizeof d, hipMemcpyDeviceToHost));
        static const char* names[NS] = {"refill", "seek", "fwd", "left", "pick+pop", "store+push", "fin_light", "fin_scan", "fin_coop", "fin_bits", "fin_mask", "fwd+seek", "fwd:issue", "fwd:wait", "fwd:wait+compute"};
        fprintf(stderr, "[pa map stats] grid=%u", cx->last_grid);
        for (uint32_t i = 0; i < NS; ++i)
            if (d[i])
                fprintf(stderr, " %s: %llu iters x %.1f lanes, %.0f ticks/iter;", names[i], d[i], (double)d[NS + i] / (double)d[i],
                        (double)d[2 * NS + i] / (double)d[i]);
        unsigned long long tops[2];
        HIP_TRY(hipMemcpy(tops, cx->ctl.as<unsigned long long>() + 57, sizeof tops, hipMemcpyDeviceToHost));
        fprintf(stderr, " key stream %llu entries, deferred stream %llu entries\n", tops[0], tops[1]);
    }
    // the counter includes every wave's partly used chunk and may run past the caller's arena without any allocation having
    // crossed its end: what may be copied back is min(top, capacity); `needed` is the capacity that would have sufficed
    if (arena_used) *arena_used = ctl.top < cx->last_arena_cap ? ctl.top : cx->last_arena_cap;
    if (arena_needed) *arena_needed = ctl.top;
    if (ctl.status & PA_STATUS_SPILL_OVERFLOW) return fail(PA_ERR_INTERNAL, "a per-launch stream (class rows, count keys or deferred reads) overflowed its buffer (should be impossible)");
    if (ctl.status & PA_STATUS_ARENA_FULL) return fail(PA_ERR_ARENA_FULL, "class arena too small: %llu entries needed", ctl.top);
    return PA_OK;
}

// The launch context of `stream` (control block, list-mode rows, key / deferred-read streams, novel list: 2 GB + 45 bytes per read at 150 bp on a 256-CU
// part) is freed; the next launch on that stream creates a fresh one. Callers that create and destroy streams call this
// before hipStreamDestroy — a context left behind would stay until pa_index_destroy and could be matched to a later stream
// whose handle value the runtime reuses.
int pa_index_release_stream(pa_index* idx, void* stream) {
    if (!idx) return fail(PA_ERR_INVALID_ARG, "null argument");
    std::shared_ptr<LaunchCtx> cx;
    {
        std::lock_guard<std::mutex> g(idx->mu);
        auto it = idx->ctxs.find(static_cast<hipStream_t>(stream));
        if (it == idx->ctxs.end()) return PA_OK;
        cx = std::move(it->second);
        idx->ctxs.erase(it);
    }
    std::lock_guard<std::mutex> g(cx->mu);   // (a launch in progress on another thread finishes first; one that only holds the pointer yet finds empty buffers and re-creates them)
    hipError_t e = hipSetDevice(idx->device);
    if (e == hipSuccess) e = hipStreamSynchronize(static_cast<hipStream_t>(stream));
    cx->release();                           // also on the error path: the context is no longer reachable from the index
    if (e != hipSuccess) return fail(PA_ERR_HIP, "pa_index_release_stream: %s", hipGetErrorString(e));
    return PA_OK;
}

int pa_map_batch_device(pa_index* idx, const uint64_t* d_tiles, const uint32_t* d_lens, uint64_t n_reads, uint32_t words_per_read,
                        uint32_t allowed_mismatches, pa_read_result* d_results, uint32_t* d_arena, uint64_t arena_cap,
                        uint32_t* d_colour, void* stream) {
    if (!idx || (n_reads && (!d_tiles || !d_lens || !d_results || !d_arena))) return fail(PA_ERR_INVALID_ARG, "null argument");
    std::shared_ptr<LaunchCtx> cx;
    const int rc0 = ctx_of(idx, static_cast<hipStream_t>(stream), &cx);
    if (rc0 != PA_OK) return rc0;
    std::lock_guard<std::mutex> g(cx->mu);
    HIP_TRY(hipSetDevice(idx->device));
    return map_launch_locked(idx, cx.get(), d_tiles, d_lens, n_reads, words_per_read, allowed_mismatches, d_results, d_arena, arena_cap, d_colour,
                             nullptr, nullptr, nullptr, static_cast<hipStream_t>(stream));
}

int pa_map_count_batch_device(pa_index* idx, const uint64_t* d_tiles, const uint32_t* d_lens, uint64_t n_reads, uint32_t words_per_read,
                              uint32_t allowed_mismatches, pa_read_result* d_results, uint32_t* d_arena, uint64_t arena_cap,
                              uint64_t* d_counts, void* stream) {
    if (!idx || !d_counts || (n_reads && (!d_tiles || !d_lens || !d_results || !d_arena))) return fail(PA_ERR_INVALID_ARG, "null argument");
    std::shared_ptr<LaunchCtx> cx;
    const int rc0 = ctx_of(idx, static_cast<hipStream_t>(stream), &cx);
    if (rc0 != PA_OK) return rc0;
    std::lock_guard<std::mutex> g(cx->mu);
    HIP_TRY(hipSetDevice(idx->device));
    return map_launch_locked(idx, cx.get(), d_tiles, d_lens, n_reads, words_per_read, allowed_mismatches, d_results, d_arena, arena_cap, nullptr,
                             d_counts, nullptr, nullptr, static_cast<hipStream_t>(stream));
}

int pa_map_count_batch_uniform_device(pa_index* idx, const uint64_t* d_tiles, uint32_t read_len, uint64_t n_reads, uint32_t words_per_read,
                                      uint32_t allowed_mismatches, pa_read_result* d_results, uint32_t* d_arena, uint64_t arena_cap, uint64_t* d_counts,
                                      void* stream) {
    if (!idx || !d_counts || (n_reads && (!d_tiles || !d_results || !d_arena))) return fail(PA_ERR_INVALID_ARG, "null argument");
    if (read_len == 0 || read_len > PA_MAX_READ_LEN || read_len > 32ull * words_per_read)
        return fail(PA_ERR_INVALID_ARG, "read_len %u does not fit %u words per read (or exceeds %u bases)", read_len, words_per_read, PA_MAX_READ_LEN);
    HIP_TRY(hipSetDevice(idx->device));
    std::shared_ptr<LaunchCtx> cx;
    int rc = ctx_of(idx, static_cast<hipStream_t>(stream), &cx);
    if (rc != PA_OK) return rc;
    std::lock_guard<std::mutex> g(cx->mu);
    return map_launch_locked(idx, cx.get(), d_tiles, nullptr, n_reads, words_per_read, allowed_mismatches, d_results, d_arena, arena_cap, nullptr,
                             d_counts, nullptr, nullptr, static_cast<hipStream_t>(stream), read_len);
}

int pa_map_finish(pa_index* idx, void* stream, uint64_t* arena_used, uint64_t* arena_needed) {
    if (!idx) return fail(PA_ERR_INVALID_ARG, "null argument");
    std::shared_ptr<LaunchCtx> cx;
    const int rc0 = ctx_of(idx, static_cast<hipStream_t>(stream), &cx);
    if (rc0 != PA_OK) return rc0;
    std::lock_guard<std::mutex> g(cx->mu);
    HIP_TRY(hipSetDevice(idx->device));
    return map_finish_locked(idx, cx.get(), static_cast<hipStream_t>(stream), arena_used, arena_needed);
}

int pa_index_set_timing(pa_index* idx, int on) {
    if (!idx) return fail(PA_ERR_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> g(idx->mu);
    idx->timing = on != 0;
    return PA_OK;
}

int pa_map_kernel_ms(pa_index* idx, void* stream, float* ms) {
    if (!idx || !ms) return fail(PA_ERR_INVALID_ARG, "null argument");
    std::shared_ptr<LaunchCtx> cx;
    const int rc0 = ctx_of(idx, static_cast<hipStream_t>(stream), &cx);
    if (rc0 != PA_OK) return rc0;
    std::lock_guard<std::mutex> g(cx->mu);
    if (!cx->timed) return fail(PA_ERR_INVALID_ARG, "no timed launch on this stream (pa_index_set_timing before the launch)");
    HIP_TRY(hipEventSynchronize(cx->ev1));
    HIP_TRY(hipEventElapsedTime(ms, cx->ev0, cx->ev1));
    return PA_OK;
}

int pa_map_stage_ms(pa_index* idx, void* stream, float ms[3]) {
    if (!idx || !ms) return fail(PA_ERR_INVALID_ARG, "null argument");
    std::shared_ptr<LaunchCtx> cx;
    const int rc0 = ctx_of(idx, static_cast<hipStream_t>(stream), &cx);
    if (rc0 != PA_OK) return rc0;
    std::lock_guard<std::mutex> g(cx->mu);
    if (!cx->timed) return fail(PA_ERR_INVALID_ARG, "no timed launch on this stream (pa_index_set_timing before the launch)");
    HIP_TRY(hipEventSynchronize(cx->ev3));
    HIP_TRY(hipEventElapsedTime(&ms[0], cx->ev0, cx->ev1));
    HIP_TRY(hipEventElapsedTime(&ms[1], cx->ev1, cx->ev2));
    HIP_TRY(hipEventElapsedTime(&ms[2], cx->ev2, cx->ev3));
    return PA_OK;
}

uint64_t pa_map_arena_hint(const pa_index* idx, uint64_t n_reads) {
    // enough for ~8 ids per read plus one partially used chunk per wave; pa_map_finish reports the exact need
    const uint64_t waves = idx ? (uint64_t)idx->num_cus * 8 * (PA_MAP_BLOCK / 64) : 8192;
    return n_reads * 8 + waves * PA_ARENA_CHUNK + 4096;
}

// ---- host-buffer convenience: H2D, encode, map (retry on arena overflow), D2H, CSR in read order ----
// The reads of a host batch: ASCII (concatenated, offsets[n+1]) or already 2-bit packed (what a DnaString holds, :450: every
// read starts on a word boundary of `words`, word_offsets[n+1] in words, lens[n] in bases; layout 0 = this library's
// LSB-first words, 1 = MSB-first words: base j in bits 62 - 2 (j % 32)).
struct HostReads {
    const uint8_t* ascii = nullptr;
    const uint64_t* offsets = nullptr;
    const uint64_t* words = nullptr;
    const uint64_t* word_offsets = nullptr;
    const uint32_t* lens = nullptr;
    int layout = 0;
};

static inline uint64_t msb_to_lsb_first(uint64_t w) {   // reverse the order of the 32 two-bit fields
    w = ((w >> 2) & 0x3333333333333333ull) | ((w & 0x3333333333333333ull) << 2);
    w = ((w >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((w & 0x0F0F0F0F0F0F0F0Full) << 4);
    return __builtin_bswap64(w);
}

static int map_batch_host(pa_index* idx, const HostReads& in, uint64_t n, uint32_t allowed,
                          pa_read_result* results, uint64_t* class_offsets, const uint32_t** class_ids, uint32_t* nodes_flat,
                          uint32_t nodes_stride_cap, uint32_t* nodes_len) {
    const bool packed = in.words != nullptr || in.word_offsets != nullptr;
    if (!idx || !results) return fail(PA_ERR_INVALID_ARG, "null argument");
    if (packed ? (!in.word_offsets || !in.lens || (n && !in.words && in.word_offsets[n] != in.word_offsets[0])) : (!in.offsets || (n && !in.ascii)))
        return fail(PA_ERR_INVALID_ARG, "null argument");
    if (packed && in.layout != 0 && in.layout != 1) return fail(PA_ERR_INVALID_ARG, "packed layout %d (0 = LSB-first, 1 = MSB-first words)", in.layout);
    std::lock_guard<std::mutex> hg(idx->hmu);
    std::shared_ptr<LaunchCtx> cx;
    const int rc0 = ctx_of(idx, nullptr, &cx);
    if (rc0 != PA_OK) return rc0;
    std::lock_guard<std::mutex> g(cx->mu);
    HIP_TRY(hipSetDevice(idx->device));
    uint64_t maxlen = 1;
    for (uint64_t i = 0; i < n; ++i) {
        if (packed) {
            if (in.word_offsets[i + 1] < in.word_offsets[i] || (uint64_t)(in.lens[i] + 31) / 32 > in.word_offsets[i + 1] - in.word_offsets[i])
                return fail(PA_ERR_INVALID_ARG, "read %llu: %u bases do not fit its words", (unsigned long long)i, in.lens[i]);
            maxlen = std::max<uint64_t>(maxlen, in.lens[i]);
        } else {
            if (in.offsets[i + 1] < in.offsets[i]) return fail(PA_ERR_INVALID_ARG, "offsets not monotone at read %llu", (unsigned long long)i);
            maxlen = std::max<uint64_t>(maxlen, in.offsets[i + 1] - in.offsets[i]);
        }
    }
    if (maxlen > PA_MAX_READ_LEN) return fail(PA_ERR_UNSUPPORTED, "read longer than %u bases", PA_MAX_READ_LEN);
    const uint32_t wpr = pa_words_per_read((uint32_t)maxlen);
    hipStream_t st = nullptr;
    int rc;
    if ((rc = idx->b_tiles.ensure(pa_tiles_words(n, wpr) * 8 + 8)) || (rc = idx->b_lens.ensure((n + 64) * 4)) ||
        (rc = idx->b_results.ensure((n + 1) * sizeof(pa_read_result))))
        return rc;
    if (n == 0) { if (class_offsets) class_offsets[0] = 0; if (class_ids) *class_ids = nullptr; return PA_OK; }
    if (packed) {   // the words go into the tile layout on the host (bases beyond a read's length cleared, as the encoder leaves them)
        std::vector<uint64_t> tiles(pa_tiles_words(n, wpr), 0);
        for (uint64_t i = 0; i < n; ++i) {
            const uint64_t* w = in.words + in.word_offsets[i];
            const uint32_t len = in.lens[i], nw = (len + 31) / 32;
            uint64_t* dst = tiles.data() + ((i >> 6) * wpr) * 64 + (i & 63);
            for (uint32_t j = 0; j < nw; ++j) {
                uint64_t v = in.layout == 1 ? msb_to_lsb_first(w[j]) : w[j];
                const uint32_t rem = len - 32 * j;
                if (rem < 32) v &= (1ull << (2 * rem)) - 1;
                dst[(uint64_t)j * 64] = v;
            }
        }
        HIP_TRY(hipMemcpyAsync(idx->b_tiles.p, tiles.data(), tiles.size() * 8, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(idx->b_lens.p, in.lens, n * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));   // (`tiles` is pageable and dies with this block)
    } else {
        const uint64_t total_ascii = in.offsets[n] - in.offsets[0];
        if ((rc = idx->b_ascii.ensure(total_ascii + 64)) || (rc = idx->b_offsets.ensure((n + 1) * 8))) return rc;
        std::vector<uint64_t> rel(n + 1);
        for (uint64_t i = 0; i <= n; ++i) rel[i] = in.offsets[i] - in.offsets[0];
        HIP_TRY(hipMemcpyAsync(idx->b_ascii.p, in.ascii + in.offsets[0], total_ascii, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(idx->b_offsets.p, rel.data(), (n + 1) * 8, hipMemcpyHostToDevice, st));
        int e = launch_encode(idx->b_ascii.as<uint8_t>(), idx->b_offsets.as<uint64_t>(), n, wpr, idx->b_tiles.as<uint64_t>(),
                              idx->b_lens.as<uint32_t>(), st);
        if (e) return fail(PA_ERR_HIP, "encode launch: %s", hipGetErrorString((hipError_t)e));
        HIP_TRY(hipStreamSynchronize(st));   // (`rel` is pageable and dies with this block)
    }
    const uint32_t spill_cap = spill_cap_of(wpr);
    uint32_t *d_nodes = nullptr, *d_nodes_len = nullptr;
    if (nodes_flat) {
        if ((rc = idx->b_nodes.ensure(n * spill_cap * 4)) || (rc = idx->b_nodes_len.ensure(n * 4))) return rc;
        d_nodes = idx->b_nodes.as<uint32_t>();
        d_nodes_len = idx->b_nodes_len.as<uint32_t>();
    }
    uint64_t cap = pa_map_arena_hint(idx, n), used = 0, need = 0;
    for (int attempt = 0;; ++attempt) {
        if ((rc = idx->b_arena.ensure(cap * 4))) return rc;
        rc = map_launch_locked(idx, cx.get(), idx->b_tiles.as<uint64_t>(), idx->b_lens.as<uint32_t>(), n, wpr, allowed,
                               idx->b_results.as<pa_read_result>(), idx->b_arena.as<uint32_t>(), cap, nullptr, nullptr, d_nodes, d_nodes_len, st);
        if (rc != PA_OK) return rc;
        rc = map_finish_locked(idx, cx.get(), st, &used, &need);
        if (rc == PA_ERR_ARENA_FULL && attempt < 3) { cap = need + need / 8 + 4096; continue; }
        if (rc != PA_OK) return rc;
        break;
    }
    HIP_TRY(hipMemcpy(results, idx->b_results.p, n * sizeof(pa_read_result), hipMemcpyDeviceToHost));
    idx->h_arena.resize(used + 1);
    if (used) HIP_TRY(hipMemcpy(idx->h_arena.data(), idx->b_arena.p, used * 4, hipMemcpyDeviceToHost));
    if (class_offsets || class_ids) {
        uint64_t total = 0;
        for (uint64_t i = 0; i < n; ++i) total += results[i].class_len;
        idx->h_class_ids.resize(total + 1);
        uint64_t o = 0;
        for (uint64_t i = 0; i < n; ++i) {
            if (class_offsets) class_offsets[i] = o;
            if (results[i].class_len) {
                const uint32_t* src = (results[i].class_off & PA_CLASS_REF)
                                          ? idx->h_ec.data() + 4ull * idx->h_class_ref[results[i].class_off & ~PA_CLASS_REF] + 1
                                          : idx->h_arena.data() + results[i].class_off;
                memcpy(idx->h_class_ids.data() + o, src, results[i].class_len * 4ull);
            }
            results[i].class_off = (uint32_t)o;
            o += results[i].class_len;
        }
        if (class_offsets) class_offsets[n] = o;
        if (class_ids) *class_ids = idx->h_class_ids.data();
    }
    if (nodes_flat) {
        std::vector<uint32_t> hn(n * (size_t)spill_cap), hl(n);
        HIP_TRY(hipMemcpy(hn.data(), d_nodes, hn.size() * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(hl.data(), d_nodes_len, n * 4, hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < n; ++i) {
            nodes_len[i] = hl[i];
            const uint32_t m = std::min(std::min(hl[i], spill_cap), nodes_stride_cap);
            memcpy(nodes_flat + i * nodes_stride_cap, hn.data() + i * spill_cap, m * 4ull);
        }
    }
    return PA_OK;
}

int pa_map_batch(pa_index* idx, const uint8_t* ascii, const uint64_t* offsets, uint64_t n_reads, uint32_t allowed_mismatches,
                 pa_read_result* results, uint64_t* class_offsets, const uint32_t** class_ids) {
    HostReads in;
    in.ascii = ascii;
    in.offsets = offsets;
    if (!offsets) return fail(PA_ERR_INVALID_ARG, "null argument");
    return map_batch_host(idx, in, n_reads, allowed_mismatches, results, class_offsets, class_ids, nullptr, 0, nullptr);
}

int pa_map_batch_packed(pa_index* idx, const uint64_t* words, const uint64_t* word_offsets, const uint32_t* lens, uint64_t n_reads, int layout,
                        uint32_t allowed_mismatches, pa_read_result* results, uint64_t* class_offsets, const uint32_t** class_ids) {
    HostReads in;
    in.words = words;
    in.word_offsets = word_offsets;
    in.lens = lens;
    in.layout = layout;
    if (!word_offsets || !lens) return fail(PA_ERR_INVALID_ARG, "null argument");
    return map_batch_host(idx, in, n_reads, allowed_mismatches, results, class_offsets, class_ids, nullptr, 0, nullptr);
}

// map_read_with_mismatch on a read the caller holds 2-bit packed (a DnaString): no ASCII round trip
int pa_map_read_packed(pa_index* idx, const uint64_t* words, uint32_t len, int layout, uint32_t allowed_mismatches, uint32_t* class_buf,
                       uint32_t class_cap, uint32_t* class_len, uint32_t* coverage, uint32_t* mismatches) {
    const uint64_t word_offsets[2] = {0, (len + 31) / 32};
    pa_read_result r;
    uint64_t co[2];
    const uint32_t* ids = nullptr;
    const int rc = pa_map_batch_packed(idx, words, word_offsets, &len, 1, layout, allowed_mismatches, &r, co, &ids);
    if (rc != PA_OK) return rc;
    if (class_len) *class_len = r.class_len;
    if (coverage) *coverage = r.coverage;
    if (mismatches) *mismatches = r.mismatches & ~PA_MAPPED_BIT;
    if (!(r.mismatches & PA_MAPPED_BIT)) return 0;
    if (r.class_len > class_cap) return fail(PA_ERR_INVALID_ARG, "class buffer too small: %u ids", r.class_len);
    if (r.class_len && class_buf) memcpy(class_buf, ids, r.class_len * 4ull);
    return 1;
}

int pa_map_read_with_mismatch(pa_index* idx, const uint8_t* ascii, uint32_t len, uint32_t allowed_mismatches, uint32_t* class_buf,
                              uint32_t class_cap, uint32_t* class_len, uint32_t* coverage, uint32_t* mismatches) {
    const uint64_t offsets[2] = {0, len};
    pa_read_result r;
    uint64_t co[2];
    const uint32_t* ids = nullptr;
    const int rc = pa_map_batch(idx, ascii, offsets, 1, allowed_mismatches, &r, co, &ids);
    if (rc != PA_OK) return rc;
    if (class_len) *class_len = r.class_len;
    if (coverage) *coverage = r.coverage;
    if (mismatches) *mismatches = r.mismatches & ~PA_MAPPED_BIT;
    if (!(r.mismatches & PA_MAPPED_BIT)) return 0;
    if (r.class_len > class_cap) return fail(PA_ERR_INVALID_ARG, "class buffer too small: %u ids", r.class_len);
    if (r.class_len && class_buf) memcpy(class_buf, ids, r.class_len * 4ull);
    return 1;
}

int pa_map_read(pa_index* idx, const uint8_t* ascii, uint32_t len, uint32_t* class_buf, uint32_t class_cap, uint32_t* class_len,
                uint32_t* coverage) {
    return pa_map_read_with_mismatch(idx, ascii, len, PA_DEFAULT_ALLOWED_MISMATCHES, class_buf, class_cap, class_len, coverage, nullptr);
}

int pa_map_read_to_nodes(pa_index* idx, const uint8_t* ascii, uint32_t len, uint32_t allowed_mismatches, uint32_t* node_buf,
                         uint32_t node_cap, uint32_t* num_nodes, uint32_t* coverage, uint32_t* mismatches) {
    const uint64_t offsets[2] = {0, len};
    pa_read_result r;
    uint32_t nn = 0;
    std::vector<uint32_t> tmp(node_cap ? node_cap : 1);
    HostReads in;
    in.ascii = ascii;
    in.offsets = offsets;
    const int rc = map_batch_host(idx, in, 1, allowed_mismatches, &r, nullptr, nullptr, tmp.data(), node_cap, &nn);
    if (rc != PA_OK) return rc;
    if (num_nodes) *num_nodes = nn;
    if (coverage) *coverage = r.coverage;
    if (mismatches) *mismatches = r.mismatches & ~PA_MAPPED_BIT;
    if (!(r.mismatches & PA_MAPPED_BIT)) return 0;
    if (nn > node_cap) return fail(PA_ERR_INVALID_ARG, "node buffer too small: %u nodes", nn);
    if (node_buf) memcpy(node_buf, tmp.data(), nn * 4ull);
    return 1;
}

// batch variant of the node trace (test surface)
int pa_map_batch_nodes(pa_index* idx, const uint8_t* ascii, const uint64_t* offsets, uint64_t n_reads, uint32_t allowed_mismatches,
                       pa_read_result* results, uint32_t* nodes_flat, uint32_t nodes_stride, uint32_t* nodes_len) {
    HostReads in;
    in.ascii = ascii;
    in.offsets = offsets;
    if (!offsets) return fail(PA_ERR_INVALID_ARG, "null argument");
    return map_batch_host(idx, in, n_reads, allowed_mismatches, results, nullptr, nullptr, nodes_flat, nodes_stride, nodes_len);
}

int pa_counts_by_barcode_device(pa_index* idx, const pa_read_result* d_results, const uint32_t* d_arena, const uint32_t* d_barcode, uint64_t n_reads,
                                uint32_t barcode_bits, uint64_t* d_keys, uint32_t* d_vals, uint64_t* n_entries, void* stream) {
    if (!idx || !n_entries || (n_reads && (!d_results || !d_arena || !d_barcode || !d_keys || !d_vals))) return fail(PA_ERR_INVALID_ARG, "null argument");
    HIP_TRY(hipSetDevice(idx->device));
    return barcode_counts(idx->dv, static_cast<const uint32_t*>(idx->d_class_table), idx->class_table_size, d_results, d_arena, d_barcode, n_reads,
                          barcode_bits, d_keys, d_vals, n_entries, static_cast<hipStream_t>(stream));
}

int pa_index_set_overflow(pa_index* idx, pa_overflow* ovf) {
    if (!idx) return fail(PA_ERR_INVALID_ARG, "null argument");
    if (ovf && overflow_device(ovf) != idx->device) return fail(PA_ERR_INVALID_ARG, "overflow table lives on device %d, the index on device %d", overflow_device(ovf), idx->device);
    std::lock_guard<std::mutex> g(idx->mu);
    idx->ovf = ovf;
    return PA_OK;
}

// ---- counts ----
uint64_t pa_counts_len(const pa_index* idx) { return idx ? (uint64_t)idx->stats.num_classes + 3 : 0; }

int pa_counts_accumulate_device(pa_index* idx, const pa_read_result* d_results, const uint32_t* d_arena, const uint32_t* d_colour,
                                uint64_t n_reads, uint64_t* d_counts, void* stream) {
    if (!idx || !d_results || !d_arena || !d_counts) return fail(PA_ERR_INVALID_ARG, "null argument");
    HIP_TRY(hipSetDevice(idx->device));
    const int e = launch_count(d_results, d_arena, d_colour, n_reads, idx->dv, static_cast<const uint32_t*>(idx->d_class_table),
                               idx->class_table_size, reinterpret_cast<unsigned long long*>(d_counts), static_cast<hipStream_t>(stream));
    if (e) return fail(PA_ERR_HIP, "count launch: %s", hipGetErrorString((hipError_t)e));
    return PA_OK;
}

// ---- synthetic reads on the device ----
struct pa_txome_device {
    int device;
    void *d_packed, *d_tx_start, *d_cum;
    uint32_t num_tx, read_len;
    uint64_t total;
};

int pa_txome_upload(const pa_txome* t, uint32_t read_len, int device, pa_txome_device** out) {
    if (!t || !out || read_len == 0 || read_len > PA_MAX_SIM_READ_LEN) return fail(PA_ERR_INVALID_ARG, "bad argument");
    int rc = use_device(device);
    if (rc != PA_OK) return rc;
    std::vector<uint64_t> cum;
    synth::build_cum(t->t.tx_start.data(), t->t.num_tx(), read_len, cum);
    if (cum.back() == 0) return fail(PA_ERR_INVALID_ARG, "no transcript is at least %u bases long", read_len);
    pa_txome_device* d = new pa_txome_device{device, nullptr, nullptr, nullptr, t->t.num_tx(), read_len, cum.back()};
    rc = upload(t->t.packed.data(), t->t.packed.size() * 8, &d->d_packed);
    if (rc == PA_OK) rc = upload(t->t.tx_start.data(), t->t.tx_start.size() * 8, &d->d_tx_start);
    if (rc == PA_OK) rc = upload(cum.data(), cum.size() * 8, &d->d_cum);
    if (rc != PA_OK) { pa_txome_device_destroy(d); return rc; }
    *out = d;
    return PA_OK;
}

void pa_txome_device_destroy(pa_txome_device* t) {
    if (!t) return;
    (void)hipSetDevice(t->device);
    for (void* p : {t->d_packed, t->d_tx_start, t->d_cum})
        if (p) (void)hipFree(p);
    delete t;
}

int pa_simulate_reads_device(const pa_txome_device* t, uint64_t seed, uint32_t sub_rate_ppm, uint64_t first_read, uint64_t n_reads,
                             uint32_t words_per_read, uint64_t* d_tiles, uint32_t* d_lens, void* stream) {
    if (!t || !d_tiles || !d_lens) return fail(PA_ERR_INVALID_ARG, "null argument");
    if (words_per_read < (t->read_len + 31) / 32) return fail(PA_ERR_INVALID_ARG, "words_per_read too small");
    HIP_TRY(hipSetDevice(t->device));
    const int e = launch_simulate(static_cast<const uint64_t*>(t->d_packed), static_cast<const uint64_t*>(t->d_tx_start),
                                  static_cast<const uint64_t*>(t->d_cum), t->num_tx, t->total, t->read_len, seed, sub_rate_ppm, first_read,
                                  n_reads, words_per_read, d_tiles, d_lens, static_cast<hipStream_t>(stream));
    if (e) return fail(PA_ERR_HIP, "simulate launch: %s", hipGetErrorString((hipError_t)e));
    return PA_OK;
}

// ---- plumbing for hosts without their own allocator / event API ----
int pa_event_create(void** ev) {
    if (!ev) return fail(PA_ERR_INVALID_ARG, "null argument");
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    *ev = e;
    return PA_OK;
}
int pa_event_record(void* ev, void* stream) { HIP_TRY(hipEventRecord(static_cast<hipEvent_t>(ev), static_cast<hipStream_t>(stream))); return PA_OK; }
int pa_event_elapsed_ms(void* start, void* stop, float* ms) {
    HIP_TRY(hipEventSynchronize(static_cast<hipEvent_t>(stop)));
    HIP_TRY(hipEventElapsedTime(ms, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)));
    return PA_OK;
}
int pa_event_destroy(void* ev) { HIP_TRY(hipEventDestroy(static_cast<hipEvent_t>(ev))); return PA_OK; }

int pa_device_malloc(int device, size_t bytes, void** out) {
    if (!out) return fail(PA_ERR_INVALID_ARG, "null argument");
    int rc = use_device(device);
    if (rc != PA_OK) return rc;
    hipError_t e = hipMalloc(out, bytes ? bytes : 16);
    if (e != hipSuccess) return fail(PA_ERR_OOM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    return PA_OK;
}
int pa_device_free(void* p) { if (p) HIP_TRY(hipFree(p)); return PA_OK; }
int pa_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)));
    return PA_OK;
}
int pa_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
    HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return PA_OK;
}
int pa_memset_device(void* dst, int value, size_t bytes, void* stream) {
    HIP_TRY(hipMemsetAsync(dst, value, bytes, static_cast<hipStream_t>(stream)));
    return PA_OK;
}
int pa_stream_synchronize(void* stream) { HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream))); return PA_OK; }

}  // extern "C"
