// The stages of the host ingest pipeline that pa_process_reads (fastq.cpp: records inside a FASTQ file) and the record stream
// (record_stream.cpp: records pushed by the caller) share: worker pool, 2-bit packing into pinned tiles, the GPU leg of one
// batch (H2D -> pa_map_batch_device -> D2H, arena regrown on demand) and the rendering of the reference's Debug tuples
// (src/pseudoaligner.rs:455-461, :490). Pure host code around the C ABI's device entry points; header-only, internal.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "kernels.hpp"
#include "pa_common.hpp"

namespace pa {
namespace ingest {

#define PA_INGEST_HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(PA_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); } while (0)

// persistent worker threads; run(n, fn) executes fn(0..n-1) on them (and on the caller) and returns when all are done
class Pool {
public:
    explicit Pool(int threads) : nthreads_(threads < 1 ? 1 : threads) {
        for (int t = 1; t < nthreads_; ++t) workers_.emplace_back([this] { loop(); });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> g(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto& w : workers_) w.join();
    }
    int size() const { return nthreads_; }
    void run(int ntasks, const std::function<void(int)>& fn) {
        if (ntasks <= 0) return;
        { std::lock_guard<std::mutex> g(mu_); fn_ = &fn; ntasks_ = ntasks; next_ = 0; pending_ = ntasks; ++epoch_; }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> g(mu_);
        done_.wait(g, [this] { return pending_ == 0; });
        fn_ = nullptr;
    }

    // the same without the caller: begin() hands the tasks to the workers and returns, end() waits for them (pa_process_reads reads the next window of the
    // text while the caller launches the kernels of the one before). One job at a time: run() / begin() only after end(). Without workers begin() runs the tasks.
    void begin(int ntasks, std::function<void(int)> fn) {
        if (ntasks <= 0) return;
        if (workers_.empty()) { for (int t = 0; t < ntasks; ++t) fn(t); return; }
        { std::lock_guard<std::mutex> g(mu_); async_fn_ = std::move(fn); fn_ = &async_fn_; ntasks_ = ntasks; next_ = 0; pending_ = ntasks; ++epoch_; }
        cv_.notify_all();
    }
    void end() {
        std::unique_lock<std::mutex> g(mu_);
        done_.wait(g, [this] { return pending_ == 0; });
        fn_ = nullptr;
    }

private:
    void work() {
        for (;;) {
            int t;
            const std::function<void(int)>* fn;
            { std::lock_guard<std::mutex> g(mu_); if (!fn_ || next_ >= ntasks_) return; t = next_++; fn = fn_; }
            (*fn)(t);
            { std::lock_guard<std::mutex> g(mu_); if (--pending_ == 0) done_.notify_all(); }
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            { std::unique_lock<std::mutex> g(mu_); cv_.wait(g, [&] { return stop_ || epoch_ != seen; }); if (stop_) return; seen = epoch_; }
            work();
        }
    }
    int nthreads_;
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    const std::function<void(int)>* fn_ = nullptr;
    std::function<void(int)> async_fn_;
    int ntasks_ = 0, next_ = 0, pending_ = 0;
    uint64_t epoch_ = 0;
    bool stop_ = false;
};

// text of one batch: one growable byte buffer per formatting thread, reused from batch to batch (fresh memory would be
// page-faulted in again every time)
struct RawBytes {   // growable bytes that are NOT zero-filled when they grow (a std::vector would write every new byte once before the copy writes it again)
    char* p = nullptr;
    size_t cap = 0;
    RawBytes() = default;
    RawBytes(RawBytes&& o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
    RawBytes& operator=(RawBytes&& o) noexcept { if (this != &o) { free(p); p = o.p; cap = o.cap; o.p = nullptr; o.cap = 0; } return *this; }
    RawBytes(const RawBytes&) = delete;
    RawBytes& operator=(const RawBytes&) = delete;
    ~RawBytes() { free(p); }
    char* data() { return p; }
    const char* data() const { return p; }
    size_t size() const { return cap; }
    void grow(size_t want, size_t keep) {
        char* q = (char*)malloc(want);
        if (!q) throw std::bad_alloc();
        if (keep) memcpy(q, p, keep);
        free(p);
        p = q;
        cap = want;
    }
};
struct TextBuf {
    RawBytes mem;
    size_t len = 0;
    char* room(size_t need) {   // at least `need` more bytes
        if (len + need > mem.size()) mem.grow(std::max(mem.size() * 2, len + need + (1 << 16)), len);
        return mem.data() + len;
    }
};

struct Record {   // one record inside a text (the mapped FASTQ file, or the bytes a caller pushed): offsets into that text
    uint64_t id_off;
    uint32_t id_len, seq_len;
    uint64_t seq_off;
};


struct BatchCtx {   // pinned host buffers + device buffers of one batch in flight
    void *d_tiles = nullptr, *d_lens = nullptr, *d_results = nullptr, *d_arena = nullptr;
    // (a) record stream: the batch's sequences as the records hold them (ASCII, back to back) and their offsets: the 2-bit packing into
    // tiles runs on the GPU (pa_encode_reads_device), the host only gathers the bytes into pinned memory
    uint8_t* h_ascii = nullptr;
    uint64_t* h_soff = nullptr;
    void *d_ascii = nullptr, *d_soff = nullptr;
    size_t ascii_cap = 0, ascii_bytes = 0, soff_cap = 0;
    // ... and the ids (record.id(), :456) for the render kernels (render.hip), which write the batch's output tuples: lengths, offsets
    // (d_off[n] = bytes of the whole text), the text itself, and its copy in pinned memory
    uint8_t* h_ids = nullptr;
    uint64_t* h_idoff = nullptr;
    void *d_ids = nullptr, *d_idoff = nullptr, *d_len = nullptr, *d_off = nullptr, *d_scan = nullptr, *d_flag = nullptr, *d_text = nullptr;
    unsigned long long* h_tot = nullptr;   // pinned {text bytes, flagged reads by bucket [PA_RENDER_FLAG_BUCKETS]}: bucket 0 = the reads before flag_mark, bucket j = the j-th million behind it
    char* h_text = nullptr;
    size_t ids_cap = 0, ids_bytes = 0, scan_bytes = 0, text_cap = 0, text_bytes = 0;
    size_t text_guess = 0, spec_bytes = 0;   // the text's expected length (from the batch before) and what was rendered + fetched ahead of knowing it
    uint64_t flagged = 0, flag_mark = 0;
    hipEvent_t ev_text = nullptr;
    size_t tiles_bytes = 0, arena_entries = 0, reads_cap = 0;
    std::vector<Record> recs;
    uint64_t first = 0, n = 0;
    uint32_t wpr = 1;
    // (b) pa_process_reads: a WINDOW of the FASTQ text as the file holds it — pinned copy, copy in HBM — whose records are found where they
    // lie: by the GPU (fastq_scan.hip: d_chunk / d_first / d_ls scratch, d_rec the records, d_info -> h_info what the host needs to go on) or,
    // for text the GPU scan does not take (the end of the file, wrapped records), by the host's scan (h_rec). Sequences and ids are
    // then read in place: no gather, no second copy
    bool in_place = false;
    uint8_t* h_raw = nullptr;
    void* d_raw = nullptr;
    size_t raw_cap = 0;
    uint64_t raw_begin = 0, raw_end = 0;   // the window's bytes are [raw_begin, raw_end) of h_raw / d_raw
    void *d_chunk = nullptr, *d_first = nullptr, *d_fq_tmp = nullptr, *d_ls = nullptr, *d_rec = nullptr, *d_info = nullptr;
    size_t chunk_cap = 0, fq_tmp_bytes = 0, ls_cap = 0, rec_cap = 0, h_rec_cap = 0;
    uint4* h_rec = nullptr;
    FqInfo* h_info = nullptr;
    hipEvent_t ev_h2d = nullptr, ev_info = nullptr;
    // the tuples' way back on a stream of its own (the lane's; not owned here): the next window's kernels do not queue behind 14 MB of text going to the host
    hipStream_t back = nullptr;
    hipEvent_t ev_render = nullptr;
    bool text_on_back = false;
    void release() {
        for (void* p : {(void*)h_ascii, (void*)h_soff, (void*)h_ids, (void*)h_idoff, (void*)h_tot, (void*)h_text, (void*)h_raw, (void*)h_rec, (void*)h_info})
            if (p) (void)hipHostFree(p);
        for (hipEvent_t e : {ev_text, ev_h2d, ev_info, ev_render})
            if (e) (void)hipEventDestroy(e);
        for (void* p : {d_tiles, d_lens, d_results, d_arena, d_ascii, d_soff, d_ids, d_idoff, d_len, d_off, d_scan, d_flag, d_text, d_raw, d_chunk, d_first, d_fq_tmp, d_ls, d_rec, d_info})
            if (p) (void)hipFree(p);
        *this = BatchCtx();
    }
};

// pinned host + device buffers of a batch of n reads of wpr words (grow-only; cap_reads: the size to allocate when growing)
inline int batch_ensure(pa_index* idx, BatchCtx& c, uint64_t n, uint32_t wpr, uint64_t cap_reads) {
    cap_reads = std::max<uint64_t>(n, cap_reads);
    const size_t tb = pa_tiles_words(n, wpr) * 8 + 8;
    if (tb > c.tiles_bytes || !c.d_tiles) {   // (the tiles only exist on the device: pa_encode_reads_device writes them)
        const size_t want = pa_tiles_words(cap_reads, wpr) * 8 + 8;
        if (c.d_tiles) (void)hipFree(c.d_tiles);
        c.d_tiles = nullptr;
        c.tiles_bytes = 0;
        PA_INGEST_HIP_OK(hipMalloc(&c.d_tiles, want));
        c.tiles_bytes = want;
    }
    if (n + 64 > c.reads_cap) {
        const size_t cap = cap_reads + 64;
        for (void** q : {&c.d_lens, &c.d_results, &c.d_len, &c.d_off, &c.d_scan}) { if (*q) (void)hipFree(*q); *q = nullptr; }
        c.reads_cap = 0;
        PA_INGEST_HIP_OK(hipMalloc(&c.d_lens, cap * 4));
        PA_INGEST_HIP_OK(hipMalloc(&c.d_results, cap * sizeof(pa_read_result)));
        PA_INGEST_HIP_OK(hipMalloc(&c.d_len, (cap + 1) * 4));
        PA_INGEST_HIP_OK(hipMalloc(&c.d_off, (cap + 1) * 8));
        c.scan_bytes = render_scan_bytes(cap);
        PA_INGEST_HIP_OK(hipMalloc(&c.d_scan, c.scan_bytes ? c.scan_bytes : 16));
        c.reads_cap = cap;
    }
    if (!c.h_tot) {
        PA_INGEST_HIP_OK(hipHostMalloc((void**)&c.h_tot, (1 + PA_RENDER_FLAG_BUCKETS) * 8, hipHostMallocDefault));
        PA_INGEST_HIP_OK(hipMalloc(&c.d_flag, PA_RENDER_FLAG_BUCKETS * 8));
        PA_INGEST_HIP_OK(hipEventCreateWithFlags(&c.ev_text, hipEventDisableTiming | hipEventBlockingSync));
    }
    if (!c.in_place) {   // gathered ids and sequences (record stream)
        if (n + 64 > c.soff_cap) {
            const size_t cap = cap_reads + 64;
            if (c.h_soff) (void)hipHostFree(c.h_soff);
            if (c.h_idoff) (void)hipHostFree(c.h_idoff);
            for (void** q : {&c.d_soff, &c.d_idoff}) { if (*q) (void)hipFree(*q); *q = nullptr; }
            c.h_soff = nullptr; c.h_idoff = nullptr;
            c.soff_cap = 0;
            PA_INGEST_HIP_OK(hipHostMalloc((void**)&c.h_soff, (cap + 1) * 8, hipHostMallocDefault));
            PA_INGEST_HIP_OK(hipHostMalloc((void**)&c.h_idoff, (cap + 1) * 8, hipHostMallocDefault));
            PA_INGEST_HIP_OK(hipMalloc(&c.d_soff, (cap + 1) * 8));
            PA_INGEST_HIP_OK(hipMalloc(&c.d_idoff, (cap + 1) * 8));
            c.soff_cap = cap;
        }
        if (c.ids_bytes + 64 > c.ids_cap) {
            const size_t want = std::max<size_t>(c.ids_bytes + c.ids_bytes / 8 + 4096, (size_t)cap_reads * 16);
            if (c.h_ids) (void)hipHostFree(c.h_ids);
            if (c.d_ids) (void)hipFree(c.d_ids);
            c.h_ids = nullptr; c.d_ids = nullptr;
            c.ids_cap = 0;
            PA_INGEST_HIP_OK(hipHostMalloc((void**)&c.h_ids, want, hipHostMallocDefault));
            PA_INGEST_HIP_OK(hipMalloc(&c.d_ids, want));
            c.ids_cap = want;
        }
        if (c.ascii_bytes + 64 > c.ascii_cap) {   // (ascii_bytes: set by the caller before this call — the sum of the batch's sequence lengths)
            const size_t want = std::max<size_t>(c.ascii_bytes + c.ascii_bytes / 8 + 4096, (size_t)cap_reads * 32ull * wpr / 2);
            if (c.h_ascii) (void)hipHostFree(c.h_ascii);
            if (c.d_ascii) (void)hipFree(c.d_ascii);
            c.h_ascii = nullptr; c.d_ascii = nullptr;
            c.ascii_cap = 0;
            PA_INGEST_HIP_OK(hipHostMalloc((void**)&c.h_ascii, want, hipHostMallocDefault));
            PA_INGEST_HIP_OK(hipMalloc(&c.d_ascii, want));
            c.ascii_cap = want;
        }
    }
    const uint64_t hint = pa_map_arena_hint(idx, n);
    if (hint > c.arena_entries) {
        if (c.d_arena) (void)hipFree(c.d_arena);
        c.d_arena = nullptr;
        PA_INGEST_HIP_OK(hipMalloc(&c.d_arena, hint * 4));
        c.arena_entries = hint;
    }
    return PA_OK;
}

// The batch's sequences (at text + rec.seq_off) gathered back to back into pinned memory, with their offsets: what the GPU packs
// (DnaString::from_dna_string, :450 -> pa_encode_reads_device). batch_offsets first (the caller sizes the buffers by ascii_bytes).
inline void batch_offsets(Pool& pool, BatchCtx& c, std::vector<uint64_t>& part) {   // part: [sequence bytes | id bytes] before every task's records
    const int ntask = pool.size() * 4;
    part.assign(2 * ((size_t)ntask + 1), 0);
    uint64_t* ps = part.data();
    uint64_t* pi = part.data() + ntask + 1;
    pool.run(ntask, [&](int t) {
        uint64_t sum = 0, isum = 0;
        for (uint64_t i = c.n * (uint64_t)t / ntask; i < c.n * (uint64_t)(t + 1) / ntask; ++i) { sum += c.recs[i].seq_len; isum += c.recs[i].id_len; }
        ps[(size_t)t + 1] = sum;
        pi[(size_t)t + 1] = isum;
    });
    for (int t = 0; t < ntask; ++t) { ps[(size_t)t + 1] += ps[(size_t)t]; pi[(size_t)t + 1] += pi[(size_t)t]; }
    c.ascii_bytes = ps[(size_t)ntask];
    c.ids_bytes = pi[(size_t)ntask];
}
inline void batch_gather_ascii(Pool& pool, BatchCtx& c, const char* text, const std::vector<uint64_t>& part) {
    const int ntask = pool.size() * 4;
    const uint64_t* ps = part.data();
    const uint64_t* pi = part.data() + ntask + 1;
    pool.run(ntask, [&](int t) {
        uint64_t o = ps[(size_t)t], io = pi[(size_t)t];
        for (uint64_t i = c.n * (uint64_t)t / ntask; i < c.n * (uint64_t)(t + 1) / ntask; ++i) {
            const Record& rec = c.recs[i];
            c.h_soff[i] = o;
            memcpy(c.h_ascii + o, text + rec.seq_off, rec.seq_len);
            o += rec.seq_len;
            c.h_idoff[i] = io;
            memcpy(c.h_ids + io, text + rec.id_off, rec.id_len);
            io += rec.id_len;
        }
    });
    c.h_soff[c.n] = c.ascii_bytes;
    c.h_idoff[c.n] = c.ids_bytes;
}

// the GPU leg of a batch, asynchronous on `stream`: tiles H2D -> index.map_read for every read (:451) -> records D2H
struct RecPos {   // where a record lies in the text and what of it counts: found by the scan (the only stage that looks at every byte), read by the gather stage
    uint64_t start;      // of the '@'
    uint32_t hdr, seq;   // bytes of the header line and of the sequence line (without their line breaks; a CR before the break still counted)
    uint32_t id_len;     // record.id() (:456): header[1..] up to its first space, trailing white space trimmed first (bio 1.5)
    uint32_t seq_len;    // record.seq() (:449): the sequence line without a CR before its line break
};


struct IngestCache {   // the two batches in flight of a pa_process_reads call or a record stream; parked on the index in between (pa_common.hpp)
    BatchCtx ctx[4];   // (a record stream uses the first two; pa_process_reads keeps four windows per lane in flight: read | scan | map + render | write)
    std::vector<RecPos> rec_pos;   // 24 bytes per record of a host-scanned window: kept, or every call would page them in again
    std::vector<std::vector<uint32_t>> brk;   // the host scan's line-break lists (4 bytes per line), kept for the same reason
    hipStream_t copy_stream = nullptr;   // pa_process_reads: the windows' text goes to the GPU on a stream of its own, beside the kernels of the window before
    hipStream_t scan_stream = nullptr;   // ... their records are found on another (a window's scan waits for its text and the scan before, not for the kernels of the window before)
    hipStream_t back_stream = nullptr;   // ... and the tuples go to the host on a third
    // the stream the batches run on travels with the buffers: its launch context inside the index (2 GB of list-mode rows)
    // is then reused by the next call instead of being stranded behind a destroyed stream
    pa_index* idx = nullptr;
    hipStream_t stream = nullptr;
    static void destroy(void* p) {
        IngestCache* c = static_cast<IngestCache*>(p);
        for (BatchCtx& b : c->ctx) b.release();
        for (hipStream_t s : {c->copy_stream, c->scan_stream, c->back_stream})
            if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
        if (c->stream) {
            if (c->idx) (void)pa_index_release_stream(c->idx, c->stream);
            (void)hipStreamDestroy(c->stream);
        }
        delete c;
    }
};

// wall seconds of the host stages of this thread's last pa_process_reads call (pa_process_reads_stage_seconds)
inline double* last_stage_seconds() {
    static thread_local double st[PA_INGEST_STAGES] = {0, 0, 0, 0, 0, 0, 0, 0};
    return st;
}

// ---- windows of raw text (pa_process_reads) ----
constexpr uint64_t WINDOW_HEAD_ROOM = 1ull << 20;   // bytes in front of a window's own text for the unfinished last record of the window before

// pinned + device copy of a window of up to `bytes` bytes (grow-only; 64 spare bytes: the scan kernels load whole 16-byte groups)
inline int window_ensure_raw(BatchCtx& c, uint64_t bytes) {
    if (bytes + 64 <= c.raw_cap) return PA_OK;
    const size_t want = (size_t)bytes + 64;
    if (c.h_raw) (void)hipHostFree(c.h_raw);
    if (c.d_raw) (void)hipFree(c.d_raw);
    c.h_raw = nullptr; c.d_raw = nullptr;
    c.raw_cap = 0;
    PA_INGEST_HIP_OK(hipHostMalloc((void**)&c.h_raw, want, hipHostMallocDefault));
    PA_INGEST_HIP_OK(hipMalloc(&c.d_raw, want));
    c.raw_cap = want;
    return PA_OK;
}
inline int window_ensure_events(BatchCtx& c) {
    if (!c.ev_h2d) PA_INGEST_HIP_OK(hipEventCreateWithFlags(&c.ev_h2d, hipEventDisableTiming));
    // (blocking: the thread that waits for a window's scan sleeps — the pool's workers are reading the next window on every CPU of the quota, and a spinning
    // waiter on top of them gets the whole process throttled)
    if (!c.ev_info) PA_INGEST_HIP_OK(hipEventCreateWithFlags(&c.ev_info, hipEventDisableTiming | hipEventBlockingSync));
    if (!c.ev_render) PA_INGEST_HIP_OK(hipEventCreateWithFlags(&c.ev_render, hipEventDisableTiming));
    if (!c.h_info) {
        PA_INGEST_HIP_OK(hipHostMalloc((void**)&c.h_info, sizeof(FqInfo), hipHostMallocDefault));
        PA_INGEST_HIP_OK(hipMalloc(&c.d_info, sizeof(FqInfo)));
    }
    return PA_OK;
}
// records of a window: room for `recs` of them on the device (and in pinned memory when the host fills them in)
inline int window_ensure_recs(BatchCtx& c, uint64_t recs, bool host_side) {
    if (recs + 1 > c.rec_cap) {
        const size_t want = (size_t)(recs + recs / 8 + 1024);
        if (c.d_rec) (void)hipFree(c.d_rec);
        c.d_rec = nullptr;
        c.rec_cap = 0;
        PA_INGEST_HIP_OK(hipMalloc(&c.d_rec, want * sizeof(uint4)));
        c.rec_cap = want;
    }
    if (host_side && recs + 1 > c.h_rec_cap) {
        const size_t want = (size_t)(recs + recs / 8 + 1024);
        if (c.h_rec) (void)hipHostFree(c.h_rec);
        c.h_rec = nullptr;
        c.h_rec_cap = 0;
        PA_INGEST_HIP_OK(hipHostMalloc((void**)&c.h_rec, want * sizeof(uint4), hipHostMallocDefault));
        c.h_rec_cap = want;
    }
    return PA_OK;
}
// scratch of the GPU scan of the window [raw_begin, raw_end): chunk counts and prefixes, line starts for `lines` lines (0: a guess from the
// window's size — FASTQ of 150-base reads has a line break per 79 bytes, of 60-base reads per 36)
inline int window_ensure_scan(BatchCtx& c, uint64_t lines) {
    int e = window_ensure_events(c);
    if (e != PA_OK) return e;
    const uint32_t chunks = fq_chunks(c.raw_begin, c.raw_end);
    if ((size_t)chunks + 1 > c.chunk_cap) {
        const size_t want = (size_t)chunks + chunks / 8 + 64;
        for (void** q : {&c.d_chunk, &c.d_first, &c.d_fq_tmp}) { if (*q) (void)hipFree(*q); *q = nullptr; }
        c.chunk_cap = 0;
        PA_INGEST_HIP_OK(hipMalloc(&c.d_chunk, want * 4));
        PA_INGEST_HIP_OK(hipMalloc(&c.d_first, want * 4));
        c.fq_tmp_bytes = fq_scan_tmp_bytes((uint32_t)want);
        PA_INGEST_HIP_OK(hipMalloc(&c.d_fq_tmp, c.fq_tmp_bytes ? c.fq_tmp_bytes : 16));
        c.chunk_cap = want;
    }
    const uint64_t need = lines ? lines + 8 : (c.raw_end - c.raw_begin) / 32 + 1024;
    if (need > c.ls_cap) {
        const size_t want = (size_t)(need + need / 8);
        if (c.d_ls) (void)hipFree(c.d_ls);
        c.d_ls = nullptr;
        c.ls_cap = 0;
        PA_INGEST_HIP_OK(hipMalloc(&c.d_ls, want * 4));
        c.ls_cap = want;
    }
    return window_ensure_recs(c, c.ls_cap / 4, false);
}
// the window's records found on the GPU, asynchronous on `stream`: c.h_info is valid once c.ev_info has passed
inline int window_scan_enqueue(BatchCtx& c, bool rescan, hipStream_t stream) {
    const int k = launch_fq_scan((const uint8_t*)c.d_raw, c.raw_begin, c.raw_end, (uint32_t*)c.d_chunk, (uint32_t*)c.d_first, c.d_fq_tmp, c.fq_tmp_bytes, (uint32_t*)c.d_ls, c.ls_cap,
                                 (uint4*)c.d_rec, c.rec_cap, (FqInfo*)c.d_info, rescan, stream);
    if (k) return fail(PA_ERR_HIP, "FASTQ scan: %s", hipGetErrorString((hipError_t)k));
    PA_INGEST_HIP_OK(hipMemcpyAsync(c.h_info, c.d_info, sizeof(FqInfo), hipMemcpyDeviceToHost, stream));
    PA_INGEST_HIP_OK(hipEventRecord(c.ev_info, stream));
    return PA_OK;
}

// The batch's output tuples (:455-461, :490) rendered where the records, the class table and the novel class ids are (render.hip):
// lengths + scan (d_off[n] = the text's bytes, copied to h_tot with the number of flagged reads), then — when buffers exist — the
// bytes and their copy to pinned memory, c.spec_bytes of them: a guess from the batch before (the host only learns the exact length
// when the batch is finished; a text that turns out longer is rendered again by batch_finish)
inline const uint8_t* batch_id_bytes(const BatchCtx& c) { return (const uint8_t*)(c.in_place ? c.d_raw : c.d_ids); }
inline const uint4* batch_rec(const BatchCtx& c) { return c.in_place ? (const uint4*)c.d_rec : nullptr; }
inline int batch_render_enqueue(pa_index* idx, BatchCtx& c, hipStream_t stream) {
    const uint64_t* d_cls_off = nullptr;
    const uint8_t* d_cls_txt = nullptr;
    int e = index_device_class_text(idx, &d_cls_off, &d_cls_txt);
    if (e != PA_OK) return e;
    PA_INGEST_HIP_OK(hipMemsetAsync(c.d_flag, 0, PA_RENDER_FLAG_BUCKETS * 8, stream));
    int k = launch_render_len((const pa_read_result*)c.d_results, (const uint32_t*)c.d_arena, batch_id_bytes(c), (const uint64_t*)c.d_idoff, batch_rec(c), d_cls_off, d_cls_txt, c.n,
                              c.arena_entries, c.flag_mark, (uint32_t*)c.d_len, (uint64_t*)c.d_off, (unsigned long long*)c.d_flag, c.d_scan, c.scan_bytes, stream);
    if (k) return fail(PA_ERR_HIP, "render (lengths): %s", hipGetErrorString((hipError_t)k));
    PA_INGEST_HIP_OK(hipMemcpyAsync(c.h_tot, (const uint64_t*)c.d_off + c.n, 8, hipMemcpyDeviceToHost, stream));
    PA_INGEST_HIP_OK(hipMemcpyAsync(c.h_tot + 1, c.d_flag, PA_RENDER_FLAG_BUCKETS * 8, hipMemcpyDeviceToHost, stream));
    c.spec_bytes = 0;
    if (c.text_cap && c.text_guess) {
        c.spec_bytes = std::min(c.text_cap, c.text_guess);
        k = launch_render_write((const pa_read_result*)c.d_results, (const uint32_t*)c.d_arena, batch_id_bytes(c), (const uint64_t*)c.d_idoff, batch_rec(c), d_cls_off, d_cls_txt, c.n,
                                c.arena_entries, (const uint64_t*)c.d_off, (uint8_t*)c.d_text, c.spec_bytes, stream);
        if (k) return fail(PA_ERR_HIP, "render (text): %s", hipGetErrorString((hipError_t)k));
        c.text_on_back = c.back && c.ev_render;
        if (c.text_on_back) {
            PA_INGEST_HIP_OK(hipEventRecord(c.ev_render, stream));
            PA_INGEST_HIP_OK(hipStreamWaitEvent(c.back, c.ev_render, 0));
        }
        PA_INGEST_HIP_OK(hipMemcpyAsync(c.h_text, c.d_text, c.spec_bytes, hipMemcpyDeviceToHost, c.text_on_back ? c.back : stream));
    }
    return PA_OK;
}

inline int batch_launch(pa_index* idx, BatchCtx& c, hipStream_t stream) {
    if (c.in_place) {   // sequences and ids are read where they lie in the window's text (already in HBM, records in d_rec)
        const int k = launch_encode_rec((const uint8_t*)c.d_raw, (const uint4*)c.d_rec, c.n, c.wpr, (uint64_t*)c.d_tiles, (uint32_t*)c.d_lens, stream);   // :450
        if (k) return fail(PA_ERR_HIP, "encode launch: %s", hipGetErrorString((hipError_t)k));
    } else {
        PA_INGEST_HIP_OK(hipMemcpyAsync(c.d_ascii, c.h_ascii, c.ascii_bytes, hipMemcpyHostToDevice, stream));
        PA_INGEST_HIP_OK(hipMemcpyAsync(c.d_soff, c.h_soff, (c.n + 1) * 8, hipMemcpyHostToDevice, stream));
        PA_INGEST_HIP_OK(hipMemcpyAsync(c.d_ids, c.h_ids, c.ids_bytes, hipMemcpyHostToDevice, stream));
        PA_INGEST_HIP_OK(hipMemcpyAsync(c.d_idoff, c.h_idoff, (c.n + 1) * 8, hipMemcpyHostToDevice, stream));
        const int e0 = pa_encode_reads_device(idx, (const uint8_t*)c.d_ascii, (const uint64_t*)c.d_soff, c.n, c.wpr, (uint64_t*)c.d_tiles, (uint32_t*)c.d_lens, stream);   // :450
        if (e0 != PA_OK) return e0;
    }
    const int e = pa_map_batch_device(idx, (const uint64_t*)c.d_tiles, (const uint32_t*)c.d_lens, c.n, c.wpr, PA_DEFAULT_ALLOWED_MISMATCHES,
                                      (pa_read_result*)c.d_results, (uint32_t*)c.d_arena, c.arena_entries, nullptr, stream);
    if (e != PA_OK) return e;
    return batch_render_enqueue(idx, c, stream);   // (the records stay on the device: the render kernels read them there)
}

// waits for the batch; an arena that turned out too small is regrown and the batch mapped (and rendered) again. Afterwards the batch's
// tuples are in c.h_text[0 .. c.text_bytes) — or on their way there (batch_text_wait)
inline int batch_finish(pa_index* idx, BatchCtx& c, hipStream_t stream) {
    uint64_t used = 0, need = 0;
    int e = pa_map_finish(idx, stream, &used, &need);
    for (int attempt = 0; e == PA_ERR_ARENA_FULL && attempt < 3; ++attempt) {
        if (c.d_arena) (void)hipFree(c.d_arena);
        c.d_arena = nullptr;
        c.arena_entries = need + need / 8 + 4096;
        PA_INGEST_HIP_OK(hipMalloc(&c.d_arena, c.arena_entries * 4));
        e = batch_launch(idx, c, stream);
        if (e == PA_OK) e = pa_map_finish(idx, stream, &used, &need);
    }
    if (e != PA_OK) return e;
    // (pa_map_finish synchronised the stream: the lengths — and the speculative text, if any — have arrived)
    c.text_bytes = (size_t)c.h_tot[0];
    c.flagged = 0;
    for (uint32_t j = 0; j < PA_RENDER_FLAG_BUCKETS; ++j) c.flagged += c.h_tot[1 + j];
    c.text_guess = c.text_bytes + c.text_bytes / 8 + (64 << 10);
    bool on_back = c.text_on_back && c.spec_bytes != 0;
    if (c.text_bytes > c.spec_bytes) {   // no guess yet (first batches) or a text longer than guessed: size the buffers, write it, fetch it
        if (on_back) { PA_INGEST_HIP_OK(hipStreamSynchronize(c.back)); on_back = false; }   // (the copy of the guessed part is not left in flight beside this one)
        if (c.text_bytes + 64 > c.text_cap) {
            const size_t want = c.text_bytes + c.text_bytes / 4 + (1 << 20);
            if (c.h_text) (void)hipHostFree(c.h_text);
            if (c.d_text) (void)hipFree(c.d_text);
            c.h_text = nullptr; c.d_text = nullptr;
            c.text_cap = 0;
            PA_INGEST_HIP_OK(hipHostMalloc((void**)&c.h_text, want, hipHostMallocDefault));
            PA_INGEST_HIP_OK(hipMalloc(&c.d_text, want));
            c.text_cap = want;
        }
        const uint64_t* d_cls_off = nullptr;
        const uint8_t* d_cls_txt = nullptr;
        if ((e = index_device_class_text(idx, &d_cls_off, &d_cls_txt)) != PA_OK) return e;
        const int k = launch_render_write((const pa_read_result*)c.d_results, (const uint32_t*)c.d_arena, batch_id_bytes(c), (const uint64_t*)c.d_idoff, batch_rec(c), d_cls_off, d_cls_txt, c.n,
                                          c.arena_entries, (const uint64_t*)c.d_off, (uint8_t*)c.d_text, c.text_cap, stream);
        if (k) return fail(PA_ERR_HIP, "render (text): %s", hipGetErrorString((hipError_t)k));
        if (c.text_bytes) PA_INGEST_HIP_OK(hipMemcpyAsync(c.h_text, c.d_text, c.text_bytes, hipMemcpyDeviceToHost, stream));
    }
    PA_INGEST_HIP_OK(hipEventRecord(c.ev_text, on_back ? c.back : stream));
    return PA_OK;
}
// the batch's tuples have arrived in c.h_text[0 .. c.text_bytes)
inline int batch_text_wait(BatchCtx& c) {
    PA_INGEST_HIP_OK(hipEventSynchronize(c.ev_text));
    return PA_OK;
}

}  // namespace ingest
}  // namespace pa
