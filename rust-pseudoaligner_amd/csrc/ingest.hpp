// The stages of the host ingest pipeline that pa_process_reads (fastq.cpp: records inside a FASTQ file) and the record stream
// (record_stream.cpp: records pushed by the caller) share: worker pool, 2-bit packing into pinned tiles, the GPU leg of one
// batch (H2D -> pa_map_batch_device -> D2H, arena regrown on demand) and the rendering of the reference's Debug tuples
// (src/pseudoaligner.rs:455-461, :490). Pure host code around the C ABI's device entry points; header-only, internal.
#pragma once
#include <hip/hip_runtime.h>
#include <emmintrin.h>

#include <algorithm>
#include <condition_variable>
#include <cstdio>
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
typedef std::vector<TextBuf> TextSet;

// Rust `impl Debug for str`: quotes, backslash escapes for \t \r \n \\ \" and \u{..} for other control bytes; at most
// 6 * n + 2 bytes
inline char* debug_str(char* o, const char* s, size_t n) {
    *o++ = '"';
    for (size_t i = 0; i < n; ++i) {
        const unsigned char c = (unsigned char)s[i];
        if (c >= 0x20 && c != 0x7f && c != '\\' && c != '"') { *o++ = (char)c; continue; }
        *o++ = '\\';
        switch (c) {
            case '\t': *o++ = 't'; break;
            case '\r': *o++ = 'r'; break;
            case '\n': *o++ = 'n'; break;
            case '\\': *o++ = '\\'; break;
            case '"': *o++ = '"'; break;
            default: o += snprintf(o, 8, "u{%x}", c);
        }
    }
    *o++ = '"';
    return o;
}

// decimal digits two at a time from a 200-byte table
struct DigitPairs {
    char d[200];
    DigitPairs() { for (int i = 0; i < 100; ++i) { d[2 * i] = (char)('0' + i / 10); d[2 * i + 1] = (char)('0' + i % 10); } }
};
static const DigitPairs DIGIT_PAIRS;
inline char* put_u32(char* o, uint32_t v) {
    char b[10];
    int n = 10;
    while (v >= 100) { const uint32_t q = v / 100, r = v - q * 100; v = q; n -= 2; memcpy(b + n, DIGIT_PAIRS.d + 2 * r, 2); }
    if (v >= 10) { n -= 2; memcpy(b + n, DIGIT_PAIRS.d + 2 * v, 2); }
    else b[--n] = (char)('0' + v);
    memcpy(o, b + n, (size_t)(10 - n));
    return o + (10 - n);
}

template <size_t N>
inline char* put_lit(char* o, const char (&lit)[N]) {   // a string literal, copied with its known length
    memcpy(o, lit, N - 1);
    return o + (N - 1);
}
inline char* put_str(char* o, const char* s) {
    while (*s) *o++ = *s++;
    return o;
}

// does the id need no escaping at all (the usual case)? eight bytes at a time: no byte below 0x20, none of 0x7f \\ "
inline bool plain_text(const char* s, size_t n) {
    const uint64_t ones = 0x0101010101010101ull, high = 0x8080808080808080ull;
    auto haszero = [&](uint64_t x) { return (x - ones) & ~x & high; };
    size_t i = 0;
    uint64_t bad = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t x;
        memcpy(&x, s + i, 8);
        bad |= (x & high)                                   // bytes >= 0x80: left to the byte loop (which copies them)
               | ((x - 0x20 * ones) & ~x & high)             // a byte below 0x20
               | haszero(x ^ (0x7Full * ones)) | haszero(x ^ ((uint64_t)'\\' * ones)) | haszero(x ^ ((uint64_t)'"' * ones));
    }
    for (; i < n; ++i) {
        const unsigned char c = (unsigned char)s[i];
        bad |= (uint64_t)(c < 0x20 || c >= 0x7f || c == '\\' || c == '"');
    }
    return bad == 0;
}
inline char* debug_id(char* o, const char* s, size_t n) {
    if (!plain_text(s, n)) return debug_str(o, s, n);
    *o++ = '"';
    memcpy(o, s, n);
    o += n;
    *o++ = '"';
    return o;
}

// DnaString::from_dna_string (:450) as a table: A0 C1 G2 T3 in either case, anything else A
struct BaseLut {
    uint8_t v[256];
    BaseLut() { memset(v, 0, sizeof v); v['C'] = v['c'] = 1; v['G'] = v['g'] = 2; v['T'] = v['t'] = 3; }
};
static const BaseLut BASE_LUT;

struct Record {   // one record inside a text (the mapped FASTQ file, or the bytes a caller pushed): offsets into that text
    uint64_t id_off;
    uint32_t id_len, seq_len;
    uint64_t seq_off;
};


// sixteen bases -> 32 bits, the table of BaseLut in registers (SSE2): code = ((c >> 1) ^ (c >> 2)) & 3 is A0 C1 G2 T3 in either
// case; every other byte becomes A like in the table
inline uint32_t pack16(const uint8_t* p) {
    const __m128i v = _mm_loadu_si128((const __m128i*)p);
    __m128i t = _mm_and_si128(_mm_xor_si128(_mm_srli_epi16(v, 1), _mm_srli_epi16(v, 2)), _mm_set1_epi8(3));
    const __m128i u = _mm_and_si128(v, _mm_set1_epi8((char)0xDF));
    const __m128i ok = _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(u, _mm_set1_epi8('A')), _mm_cmpeq_epi8(u, _mm_set1_epi8('C'))),
                                    _mm_or_si128(_mm_cmpeq_epi8(u, _mm_set1_epi8('G')), _mm_cmpeq_epi8(u, _mm_set1_epi8('T'))));
    t = _mm_and_si128(t, ok);
    t = _mm_and_si128(_mm_or_si128(t, _mm_srli_epi16(t, 6)), _mm_set1_epi16(0x000F));    // 2 bases per 16-bit lane
    t = _mm_and_si128(_mm_or_si128(t, _mm_srli_epi32(t, 12)), _mm_set1_epi32(0x000000FF)); // 4 per 32-bit lane
    t = _mm_or_si128(t, _mm_srli_epi64(t, 24));                                             // 8 per 64-bit lane (low 16 bits)
    return ((uint32_t)_mm_cvtsi128_si32(t) & 0xFFFFu) | ((uint32_t)_mm_extract_epi16(t, 4) << 16);
}

struct BatchCtx {   // pinned host buffers + device buffers of one batch in flight
    uint64_t* h_tiles = nullptr;
    uint32_t* h_lens = nullptr;
    pa_read_result* h_results = nullptr;
    void *d_tiles = nullptr, *d_lens = nullptr, *d_results = nullptr, *d_arena = nullptr;
    // the batch's sequences as the records hold them (ASCII, back to back) and their offsets: the 2-bit packing into tiles runs on the
    // GPU (pa_encode_reads_device), the host only gathers the bytes into pinned memory
    uint8_t* h_ascii = nullptr;
    uint64_t* h_soff = nullptr;
    void *d_ascii = nullptr, *d_soff = nullptr;
    size_t ascii_cap = 0, ascii_bytes = 0;
    // ... and the ids (record.id(), :456) for the render kernels (render.hip), which write the batch's output tuples: lengths, offsets
    // (d_off[n] = bytes of the whole text), the text itself, and its copy in pinned memory
    uint8_t* h_ids = nullptr;
    uint64_t* h_idoff = nullptr;
    void *d_ids = nullptr, *d_idoff = nullptr, *d_len = nullptr, *d_off = nullptr, *d_scan = nullptr, *d_flag = nullptr, *d_text = nullptr;
    unsigned long long* h_tot = nullptr;   // pinned {text bytes, flagged reads}
    char* h_text = nullptr;
    size_t ids_cap = 0, ids_bytes = 0, scan_bytes = 0, text_cap = 0, text_bytes = 0;
    size_t text_guess = 0, spec_bytes = 0;   // the text's expected length (from the batch before) and what was rendered + fetched ahead of knowing it
    uint64_t flagged = 0;
    hipEvent_t ev_text = nullptr;
    size_t tiles_bytes = 0, arena_entries = 0, reads_cap = 0;
    std::vector<uint32_t> h_arena;
    std::vector<Record> recs;
    uint64_t first = 0, n = 0;
    uint64_t text_abs = ~0ull;   // pa_process_reads: where the batch's first record lies in the file's mapping (~0: not in one)
    uint32_t wpr = 1;
    void release() {
        if (h_tiles) (void)hipHostFree(h_tiles);
        if (h_lens) (void)hipHostFree(h_lens);
        if (h_results) (void)hipHostFree(h_results);
        if (h_ascii) (void)hipHostFree(h_ascii);
        if (h_soff) (void)hipHostFree(h_soff);
        if (h_ids) (void)hipHostFree(h_ids);
        if (h_idoff) (void)hipHostFree(h_idoff);
        if (h_tot) (void)hipHostFree(h_tot);
        if (h_text) (void)hipHostFree(h_text);
        if (ev_text) (void)hipEventDestroy(ev_text);
        for (void* p : {d_tiles, d_lens, d_results, d_arena, d_ascii, d_soff, d_ids, d_idoff, d_len, d_off, d_scan, d_flag, d_text})
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
        if (c.h_results) (void)hipHostFree(c.h_results);
        if (c.h_soff) (void)hipHostFree(c.h_soff);
        if (c.h_idoff) (void)hipHostFree(c.h_idoff);
        for (void** q : {&c.d_lens, &c.d_results, &c.d_soff, &c.d_idoff, &c.d_len, &c.d_off, &c.d_scan}) { if (*q) (void)hipFree(*q); *q = nullptr; }
        c.h_results = nullptr; c.h_soff = nullptr; c.h_idoff = nullptr;
        c.reads_cap = 0;
        PA_INGEST_HIP_OK(hipHostMalloc((void**)&c.h_results, cap * sizeof(pa_read_result), hipHostMallocDefault));
        PA_INGEST_HIP_OK(hipHostMalloc((void**)&c.h_soff, (cap + 1) * 8, hipHostMallocDefault));
        PA_INGEST_HIP_OK(hipHostMalloc((void**)&c.h_idoff, (cap + 1) * 8, hipHostMallocDefault));
        PA_INGEST_HIP_OK(hipMalloc(&c.d_lens, cap * 4));
        PA_INGEST_HIP_OK(hipMalloc(&c.d_results, cap * sizeof(pa_read_result)));
        PA_INGEST_HIP_OK(hipMalloc(&c.d_soff, (cap + 1) * 8));
        PA_INGEST_HIP_OK(hipMalloc(&c.d_idoff, (cap + 1) * 8));
        PA_INGEST_HIP_OK(hipMalloc(&c.d_len, (cap + 1) * 4));
        PA_INGEST_HIP_OK(hipMalloc(&c.d_off, (cap + 1) * 8));
        c.scan_bytes = render_scan_bytes(cap);
        PA_INGEST_HIP_OK(hipMalloc(&c.d_scan, c.scan_bytes ? c.scan_bytes : 16));
        c.reads_cap = cap;
    }
    if (!c.h_tot) {
        PA_INGEST_HIP_OK(hipHostMalloc((void**)&c.h_tot, 16, hipHostMallocDefault));
        PA_INGEST_HIP_OK(hipMalloc(&c.d_flag, 16));
        PA_INGEST_HIP_OK(hipEventCreateWithFlags(&c.ev_text, hipEventDisableTiming));
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
    BatchCtx ctx[2];
    std::vector<RecPos> rec_pos;   // 24 bytes per record of the file: kept, or every call would page 200 MB in again
    std::vector<std::vector<uint32_t>> brk;   // the scan's line-break lists (4 bytes per line), kept for the same reason
    // the stream the batches run on travels with the buffers: its launch context inside the index (2 GB of list-mode rows)
    // is then reused by the next call instead of being stranded behind a destroyed stream
    pa_index* idx = nullptr;
    hipStream_t stream = nullptr;
    static void destroy(void* p) {
        IngestCache* c = static_cast<IngestCache*>(p);
        for (BatchCtx& b : c->ctx) b.release();
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

// The batch's output tuples (:455-461, :490) rendered where the records, the class table and the novel class ids are (render.hip):
// lengths + scan (d_off[n] = the text's bytes, copied to h_tot with the number of flagged reads), then — when buffers exist — the
// bytes and their copy to pinned memory, c.spec_bytes of them: a guess from the batch before (the host only learns the exact length
// when the batch is finished; a text that turns out longer is rendered again by batch_finish)
inline int batch_render_enqueue(pa_index* idx, BatchCtx& c, hipStream_t stream) {
    const uint64_t* d_cls_off = nullptr;
    const uint8_t* d_cls_txt = nullptr;
    int e = index_device_class_text(idx, &d_cls_off, &d_cls_txt);
    if (e != PA_OK) return e;
    PA_INGEST_HIP_OK(hipMemsetAsync(c.d_flag, 0, 8, stream));
    int k = launch_render_len((const pa_read_result*)c.d_results, (const uint32_t*)c.d_arena, (const uint8_t*)c.d_ids, (const uint64_t*)c.d_idoff, d_cls_off, d_cls_txt, c.n, c.arena_entries,
                              (uint32_t*)c.d_len, (uint64_t*)c.d_off, (unsigned long long*)c.d_flag, c.d_scan, c.scan_bytes, stream);
    if (k) return fail(PA_ERR_HIP, "render (lengths): %s", hipGetErrorString((hipError_t)k));
    PA_INGEST_HIP_OK(hipMemcpyAsync(c.h_tot, (const uint64_t*)c.d_off + c.n, 8, hipMemcpyDeviceToHost, stream));
    PA_INGEST_HIP_OK(hipMemcpyAsync(c.h_tot + 1, c.d_flag, 8, hipMemcpyDeviceToHost, stream));
    c.spec_bytes = 0;
    if (c.text_cap && c.text_guess) {
        c.spec_bytes = std::min(c.text_cap, c.text_guess);
        k = launch_render_write((const pa_read_result*)c.d_results, (const uint32_t*)c.d_arena, (const uint8_t*)c.d_ids, (const uint64_t*)c.d_idoff, d_cls_off, d_cls_txt, c.n, c.arena_entries,
                                (const uint64_t*)c.d_off, (uint8_t*)c.d_text, c.spec_bytes, stream);
        if (k) return fail(PA_ERR_HIP, "render (text): %s", hipGetErrorString((hipError_t)k));
        PA_INGEST_HIP_OK(hipMemcpyAsync(c.h_text, c.d_text, c.spec_bytes, hipMemcpyDeviceToHost, stream));
    }
    return PA_OK;
}

inline int batch_launch(pa_index* idx, BatchCtx& c, hipStream_t stream) {
    PA_INGEST_HIP_OK(hipMemcpyAsync(c.d_ascii, c.h_ascii, c.ascii_bytes, hipMemcpyHostToDevice, stream));
    PA_INGEST_HIP_OK(hipMemcpyAsync(c.d_soff, c.h_soff, (c.n + 1) * 8, hipMemcpyHostToDevice, stream));
    PA_INGEST_HIP_OK(hipMemcpyAsync(c.d_ids, c.h_ids, c.ids_bytes, hipMemcpyHostToDevice, stream));
    PA_INGEST_HIP_OK(hipMemcpyAsync(c.d_idoff, c.h_idoff, (c.n + 1) * 8, hipMemcpyHostToDevice, stream));
    const int e0 = pa_encode_reads_device(idx, (const uint8_t*)c.d_ascii, (const uint64_t*)c.d_soff, c.n, c.wpr, (uint64_t*)c.d_tiles, (uint32_t*)c.d_lens, stream);   // :450
    if (e0 != PA_OK) return e0;
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
    c.flagged = c.h_tot[1];
    c.text_guess = c.text_bytes + c.text_bytes / 8 + (64 << 10);
    if (c.text_bytes > c.spec_bytes) {   // no guess yet (first batches) or a text longer than guessed: size the buffers, write it, fetch it
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
        const int k = launch_render_write((const pa_read_result*)c.d_results, (const uint32_t*)c.d_arena, (const uint8_t*)c.d_ids, (const uint64_t*)c.d_idoff, d_cls_off, d_cls_txt, c.n, c.arena_entries,
                                          (const uint64_t*)c.d_off, (uint8_t*)c.d_text, c.text_cap, stream);
        if (k) return fail(PA_ERR_HIP, "render (text): %s", hipGetErrorString((hipError_t)k));
        if (c.text_bytes) PA_INGEST_HIP_OK(hipMemcpyAsync(c.h_text, c.d_text, c.text_bytes, hipMemcpyDeviceToHost, stream));
    }
    PA_INGEST_HIP_OK(hipEventRecord(c.ev_text, stream));
    return PA_OK;
}
// the batch's tuples have arrived in c.h_text[0 .. c.text_bytes)
inline int batch_text_wait(BatchCtx& c) {
    PA_INGEST_HIP_OK(hipEventSynchronize(c.ev_text));
    return PA_OK;
}

}  // namespace ingest
}  // namespace pa
