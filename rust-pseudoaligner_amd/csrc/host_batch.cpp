// The hot path HOST TO HOST (SURVEY.md §8d's literal metric: "reads resident in host pinned memory (2-bit packed)" to "per-read outputs +
// count table on host"): a batch that lies in host memory in the tile layout is mapped in CHUNKS that rotate over several streams of
// the index handle, so that the copy of chunk i + 1 to the GPU, the kernels of chunk i and the copy of chunk i - 1's outputs back overlap.
// What comes back is the compact form (compact.hip): 8 bytes per read and the packed classes that are no index classes — 0.9 GB per
// 100 M reads against the 4 GB of packed reads that travel the other way; the link (PCIe Gen5 x16), not the kernel, bounds this path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <new>
#include <vector>

#include "pa_common.hpp"

using namespace pa;

namespace {

#define HB_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(PA_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); } while (0)

constexpr int MAX_STREAMS = 8;

struct Stage {   // one chunk in flight
    hipStream_t stream = nullptr;
    void *d_tiles = nullptr, *d_lens = nullptr, *d_res = nullptr, *d_arena = nullptr, *d_compact = nullptr, *d_packed = nullptr, *d_pw = nullptr, *d_scr = nullptr;
    uint64_t cap_reads = 0, arena_cap = 0, tiles_words = 0;
    size_t scr_bytes = 0;
    hipEvent_t ev_in = nullptr, ev_out = nullptr, ev_back = nullptr;   // the chunk's tiles have arrived | its outputs are ready on the device | ... and on the host
    int64_t busy = -1;   // the chunk whose outputs are on their way
};

struct HostPipe {
    pa_index* idx = nullptr;
    int device = 0;
    Stage st[MAX_STREAMS];
    // dedicated copy streams (A/B in knobs builds: see run())
    hipStream_t s_in = nullptr, s_back = nullptr, s_in2 = nullptr;
    void* d_counts = nullptr;
    uint64_t counts_len = 0;
    unsigned long long* h_pw = nullptr;   // pinned: words of every stage's packed classes
    static void destroy(void* p) {
        HostPipe* h = static_cast<HostPipe*>(p);
        (void)hipSetDevice(h->device);
        for (hipStream_t q : {h->s_in, h->s_back, h->s_in2})
            if (q) { (void)hipStreamSynchronize(q); (void)hipStreamDestroy(q); }
        for (Stage& s : h->st) {
            for (hipEvent_t e : {s.ev_in, s.ev_out, s.ev_back})
                if (e) (void)hipEventDestroy(e);
            if (s.stream) { (void)hipStreamSynchronize(s.stream); if (h->idx) (void)pa_index_release_stream(h->idx, s.stream); (void)hipStreamDestroy(s.stream); }
            for (void* q : {s.d_tiles, s.d_lens, s.d_res, s.d_arena, s.d_compact, s.d_packed, s.d_pw, s.d_scr})
                if (q) (void)hipFree(q);
        }
        if (h->d_counts) (void)hipFree(h->d_counts);
        if (h->h_pw) (void)hipHostFree(h->h_pw);
        delete h;
    }
};

int stage_ensure(pa_index* idx, Stage& s, uint64_t chunk, uint32_t wpr, bool lens) {
    if (!s.stream) {
        HB_HIP(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        HB_HIP(hipEventCreateWithFlags(&s.ev_in, hipEventDisableTiming));
        HB_HIP(hipEventCreateWithFlags(&s.ev_out, hipEventDisableTiming));
        HB_HIP(hipEventCreateWithFlags(&s.ev_back, hipEventDisableTiming));
    }
    const uint64_t tw = pa_tiles_words(chunk, wpr);
    if (tw > s.tiles_words) {
        if (s.d_tiles) (void)hipFree(s.d_tiles);
        s.d_tiles = nullptr; s.tiles_words = 0;
        HB_HIP(hipMalloc(&s.d_tiles, tw * 8 + 64));
        s.tiles_words = tw;
    }
    if (chunk > s.cap_reads) {
        for (void** q : {&s.d_lens, &s.d_res, &s.d_compact, &s.d_scr}) { if (*q) (void)hipFree(*q); *q = nullptr; }
        s.cap_reads = 0;
        HB_HIP(hipMalloc(&s.d_lens, chunk * 4 + 64));
        HB_HIP(hipMalloc(&s.d_res, chunk * sizeof(pa_read_result)));
        HB_HIP(hipMalloc(&s.d_compact, chunk * 8));
        s.scr_bytes = pa_compact_scratch_bytes(chunk);
        HB_HIP(hipMalloc(&s.d_scr, s.scr_bytes));
        s.cap_reads = chunk;
    }
    (void)lens;
    const uint64_t hint = pa_map_arena_hint(idx, chunk);
    if (hint > s.arena_cap) {
        for (void** q : {&s.d_arena, &s.d_packed}) { if (*q) (void)hipFree(*q); *q = nullptr; }
        s.arena_cap = 0;
        HB_HIP(hipMalloc(&s.d_arena, hint * 4));
        HB_HIP(hipMalloc(&s.d_packed, hint * 4));
        s.arena_cap = hint;
    }
    if (!s.d_pw) HB_HIP(hipMalloc(&s.d_pw, 8));
    return PA_OK;
}

int run(pa_index* idx, HostPipe& hp, const uint64_t* h_tiles, const uint32_t* h_lens, uint32_t uniform_len, uint64_t n, uint32_t wpr, uint32_t allowed, uint64_t* h_compact,
        uint32_t* h_packed, uint64_t packed_cap, uint64_t* packed_words, uint64_t* h_counts, uint64_t chunk, int ns) {
    const uint64_t counts_len = pa_counts_len(idx);
    if (counts_len > hp.counts_len) {
        if (hp.d_counts) (void)hipFree(hp.d_counts);
        hp.d_counts = nullptr; hp.counts_len = 0;
        HB_HIP(hipMalloc(&hp.d_counts, counts_len * 8));
        hp.counts_len = counts_len;
    }
    if (!hp.h_pw) HB_HIP(hipHostMalloc((void**)&hp.h_pw, MAX_STREAMS * 8, hipHostMallocDefault));
    if (!hp.s_in) HB_HIP(hipStreamCreateWithFlags(&hp.s_in, hipStreamNonBlocking));
    if (!hp.s_back) HB_HIP(hipStreamCreateWithFlags(&hp.s_back, hipStreamNonBlocking));
    if (!hp.s_in2) HB_HIP(hipStreamCreateWithFlags(&hp.s_in2, hipStreamNonBlocking));
    // Where the copies run — measured (tools/bench_e2e.py, 100 M reads of config 3, same box, profiles/r06_e2e_copy_streams.txt): every copy on its chunk's OWN stream,
    // nothing else: 76 - 113 ms, different from call to call — with several copies of one direction queued at once the runtime runs some of them as blit KERNELS
    // (rocprofv3: 8 - 16 of the 50 tile copies, 1.4 - 4 ms each against 1.4 ms by DMA). The same streams with every copy BACK waiting for the one before it
    // (an event of the chunk before: one copy back at a time): 72.9 - 73.7 ms, every call (default: back_mode 2). Also the copies IN one at a time: 74.1 - 75.1 ms.
    // Tiles in on one dedicated copy stream 84 - 86 ms, on two alternating 101 ms; outputs back on a dedicated stream 109 - 113 ms. The other
    // arrangements stay selectable in knobs builds only.
    const int in_mode = knob_int("PA_HB_IN", 0), back_mode = knob_int("PA_HB_BACK", 2);   // in: 0 = the chunk's own stream, 1 = one copy stream, 2 = two alternating, 3 = own stream, one at a time; back: 0 = own stream, 1 = one copy stream, 2 = own stream, one at a time
    for (int k = 0; k < ns; ++k) {
        const int e = stage_ensure(idx, hp.st[k], chunk, wpr, h_lens != nullptr);
        if (e != PA_OK) return e;
        hp.st[k].busy = -1;
    }
    HB_HIP(hipMemsetAsync(hp.d_counts, 0, counts_len * 8, hp.st[0].stream));
    HB_HIP(hipStreamSynchronize(hp.st[0].stream));
    const uint64_t n_chunks = (n + chunk - 1) / chunk;
    uint64_t off = 0;
    int rc = PA_OK;
    for (uint64_t c = 0; c < n_chunks + (uint64_t)ns && rc == PA_OK; ++c) {
        const int k = (int)(c % (uint64_t)ns);
        Stage& s = hp.st[k];
        if (s.busy >= 0) {   // the chunk launched ns chunks ago: its packed classes follow its records to the host
            uint64_t used = 0, need = 0;
            if ((rc = pa_map_finish(idx, s.stream, &used, &need)) != PA_OK) break;
            HB_HIP(hipEventSynchronize(s.ev_back));   // (its records and the length of its packed stream have arrived)
            const uint64_t words = hp.h_pw[k];
            if (off + words > packed_cap || words > s.arena_cap) { rc = fail(PA_ERR_ARENA_FULL, "packed classes: %llu words so far, room for %llu", (unsigned long long)(off + words), (unsigned long long)packed_cap); break; }
            const hipStream_t sb0 = back_mode == 1 ? hp.s_back : s.stream;
            if (words) HB_HIP(hipMemcpyAsync(h_packed + off, s.d_packed, words * 4, hipMemcpyDeviceToHost, sb0));
            HB_HIP(hipEventRecord(s.ev_back, sb0));                 // (the stage's d_packed is rewritten only behind this copy)
            HB_HIP(hipStreamWaitEvent(s.stream, s.ev_back, 0));
            off += words;
            s.busy = -1;
        }
        if (c < n_chunks) {
            const uint64_t lo = c * chunk, nn = std::min<uint64_t>(chunk, n - lo);   // chunk is a multiple of 64: tile aligned
            const hipStream_t si = (in_mode == 0 || in_mode == 3) ? s.stream : (in_mode == 2 && (c & 1)) ? hp.s_in2 : hp.s_in;
            if (in_mode == 3 && c > 0) HB_HIP(hipStreamWaitEvent(si, hp.st[(int)((c - 1) % (uint64_t)ns)].ev_in, 0));   // one copy to the GPU at a time, each on its chunk's own stream
            HB_HIP(hipMemcpyAsync(s.d_tiles, h_tiles + (lo / 64) * wpr * 64, pa_tiles_words(nn, wpr) * 8, hipMemcpyHostToDevice, si));
            if (h_lens) HB_HIP(hipMemcpyAsync(s.d_lens, h_lens + lo, nn * 4, hipMemcpyHostToDevice, si));
            HB_HIP(hipEventRecord(s.ev_in, si));
            HB_HIP(hipStreamWaitEvent(s.stream, s.ev_in, 0));
            if (h_lens)
                rc = pa_map_count_batch_device(idx, (const uint64_t*)s.d_tiles, (const uint32_t*)s.d_lens, nn, wpr, allowed, (pa_read_result*)s.d_res, (uint32_t*)s.d_arena, s.arena_cap,
                                               (uint64_t*)hp.d_counts, s.stream);
            else
                rc = pa_map_count_batch_uniform_device(idx, (const uint64_t*)s.d_tiles, uniform_len, nn, wpr, allowed, (pa_read_result*)s.d_res, (uint32_t*)s.d_arena, s.arena_cap,
                                                       (uint64_t*)hp.d_counts, s.stream);
            if (rc != PA_OK) break;
            rc = pa_results_compact_device(idx, (const pa_read_result*)s.d_res, (const uint32_t*)s.d_arena, s.arena_cap, nn, (uint64_t*)s.d_compact, (uint32_t*)s.d_packed, s.arena_cap,
                                           (uint64_t*)s.d_pw, s.d_scr, s.scr_bytes, s.stream);
            if (rc != PA_OK) break;
            HB_HIP(hipEventRecord(s.ev_out, s.stream));
            const hipStream_t sb = back_mode == 1 ? hp.s_back : s.stream;
            HB_HIP(hipStreamWaitEvent(sb, s.ev_out, 0));
            if (back_mode == 2 && c > 0) HB_HIP(hipStreamWaitEvent(sb, hp.st[(int)((c - 1) % (uint64_t)ns)].ev_back, 0));   // one copy back at a time
            HB_HIP(hipMemcpyAsync(h_compact + lo, s.d_compact, nn * 8, hipMemcpyDeviceToHost, sb));
            HB_HIP(hipMemcpyAsync(hp.h_pw + k, s.d_pw, 8, hipMemcpyDeviceToHost, sb));
            HB_HIP(hipEventRecord(s.ev_back, sb));
            s.busy = (int64_t)c;
        }
    }
    for (hipStream_t q : {hp.s_in, hp.s_in2, hp.s_back})
        if (q) (void)hipStreamSynchronize(q);
    for (int k = 0; k < ns; ++k)
        if (hp.st[k].stream) (void)hipStreamSynchronize(hp.st[k].stream);   // (also on the error path: nothing of this call is in flight when it returns)
    if (rc != PA_OK) return rc;
    if (h_counts) {
        HB_HIP(hipMemcpyAsync(h_counts, hp.d_counts, counts_len * 8, hipMemcpyDeviceToHost, hp.st[0].stream));
        HB_HIP(hipStreamSynchronize(hp.st[0].stream));
    }
    if (packed_words) *packed_words = off;
    return PA_OK;
}

}  // namespace

extern "C" int pa_map_tiles_host(pa_index* idx, const uint64_t* h_tiles, const uint32_t* h_lens, uint32_t uniform_len, uint64_t n_reads, uint32_t words_per_read,
                                 uint32_t allowed_mismatches, uint64_t* h_compact, uint32_t* h_packed, uint64_t packed_cap, uint64_t* packed_words, uint64_t* h_counts,
                                 uint64_t chunk_reads, int n_streams) {
    if (!idx || (n_reads && (!h_tiles || !h_compact)) || (packed_cap && !h_packed) || words_per_read == 0) return fail(PA_ERR_INVALID_ARG, "null argument");
    if (!h_lens && (uniform_len == 0 || uniform_len > PA_MAX_READ_LEN || uniform_len > 32ull * words_per_read)) return fail(PA_ERR_INVALID_ARG, "no length array and no usable uniform length");
    if (packed_words) *packed_words = 0;
    if (chunk_reads == 0) chunk_reads = 1000000;   // (1 M reads x 4 streams 71.8 - 73.3 ms per 100 M reads; 2 M 73.9 - 74.7; 4 M 72.3 - 76; 3 or 8 streams slower)
    chunk_reads = std::max<uint64_t>(64, std::min<uint64_t>(chunk_reads, std::max<uint64_t>(n_reads, 64)) / 64 * 64);
    if (n_streams <= 0) n_streams = 4;
    n_streams = std::min(n_streams, MAX_STREAMS);
    int device = 0;
    const uint32_t *h_ec = nullptr, *h_ref = nullptr;
    index_host_classes(idx, &h_ec, &h_ref, &device);
    if (hipSetDevice(device) != hipSuccess) return fail(PA_ERR_HIP, "hipSetDevice(%d) failed", device);
    HostPipe* hp = static_cast<HostPipe*>(index_take_host_pipe(idx));
    if (!hp) hp = new (std::nothrow) HostPipe();
    if (!hp) return fail(PA_ERR_OOM, "out of memory");
    hp->idx = idx;
    hp->device = device;
    const int rc = run(idx, *hp, h_tiles, h_lens, uniform_len, n_reads, words_per_read, allowed_mismatches, h_compact, h_packed, packed_cap, packed_words, h_counts, chunk_reads, n_streams);
    if (rc == PA_OK) index_put_host_pipe(idx, hp, HostPipe::destroy);   // the next batch starts with warm streams and buffers
    else { const std::string why = last_error_ref(); HostPipe::destroy(hp); last_error_ref() = why; }
    return rc;
}

extern "C" int pa_host_alloc_pinned(size_t bytes, void** out) {
    if (!out) return fail(PA_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    const hipError_t e = hipHostMalloc(out, bytes ? bytes : 16, hipHostMallocDefault);
    if (e != hipSuccess) return fail(PA_ERR_OOM, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e));
    return PA_OK;
}
extern "C" int pa_host_free_pinned(void* p) {
    if (p && hipHostFree(p) != hipSuccess) return fail(PA_ERR_HIP, "hipHostFree failed");
    return PA_OK;
}
