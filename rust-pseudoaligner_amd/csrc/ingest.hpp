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
#include <thread>
#include <vector>

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
struct TextBuf {
    std::vector<char> mem;
    size_t len = 0;
    char* room(size_t need) {   // at least `need` more bytes
        if (len + need > mem.size()) mem.resize(std::max(mem.size() * 2, len + need + (1 << 16)));
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
    size_t tiles_bytes = 0, arena_entries = 0, reads_cap = 0;
    std::vector<uint32_t> h_arena;
    std::vector<Record> recs;
    uint64_t first = 0, n = 0;
    uint32_t wpr = 1;
    void release() {
        if (h_tiles) (void)hipHostFree(h_tiles);
        if (h_lens) (void)hipHostFree(h_lens);
        if (h_results) (void)hipHostFree(h_results);
        if (h_ascii) (void)hipHostFree(h_ascii);
        if (h_soff) (void)hipHostFree(h_soff);
        for (void* p : {d_tiles, d_lens, d_results, d_arena, d_ascii, d_soff})
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
        if (c.d_lens) (void)hipFree(c.d_lens);
        if (c.d_results) (void)hipFree(c.d_results);
        if (c.d_soff) (void)hipFree(c.d_soff);
        c.h_results = nullptr; c.h_soff = nullptr; c.d_lens = nullptr; c.d_results = nullptr; c.d_soff = nullptr;
        c.reads_cap = 0;
        PA_INGEST_HIP_OK(hipHostMalloc((void**)&c.h_results, cap * sizeof(pa_read_result), hipHostMallocDefault));
        PA_INGEST_HIP_OK(hipHostMalloc((void**)&c.h_soff, (cap + 1) * 8, hipHostMallocDefault));
        PA_INGEST_HIP_OK(hipMalloc(&c.d_lens, cap * 4));
        PA_INGEST_HIP_OK(hipMalloc(&c.d_results, cap * sizeof(pa_read_result)));
        PA_INGEST_HIP_OK(hipMalloc(&c.d_soff, (cap + 1) * 8));
        c.reads_cap = cap;
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
inline void batch_offsets(Pool& pool, BatchCtx& c, std::vector<uint64_t>& part) {
    const int ntask = pool.size() * 4;
    part.assign((size_t)ntask + 1, 0);
    pool.run(ntask, [&](int t) {
        uint64_t sum = 0;
        for (uint64_t i = c.n * (uint64_t)t / ntask; i < c.n * (uint64_t)(t + 1) / ntask; ++i) sum += c.recs[i].seq_len;
        part[(size_t)t + 1] = sum;
    });
    for (int t = 0; t < ntask; ++t) part[(size_t)t + 1] += part[(size_t)t];
    c.ascii_bytes = part[(size_t)ntask];
}
inline void batch_gather_ascii(Pool& pool, BatchCtx& c, const char* text, const std::vector<uint64_t>& part) {
    const int ntask = pool.size() * 4;
    pool.run(ntask, [&](int t) {
        uint64_t o = part[(size_t)t];
        for (uint64_t i = c.n * (uint64_t)t / ntask; i < c.n * (uint64_t)(t + 1) / ntask; ++i) {
            const Record& rec = c.recs[i];
            c.h_soff[i] = o;
            memcpy(c.h_ascii + o, text + rec.seq_off, rec.seq_len);
            o += rec.seq_len;
        }
    });
    c.h_soff[c.n] = c.ascii_bytes;
}

// the GPU leg of a batch, asynchronous on `stream`: tiles H2D -> index.map_read for every read (:451) -> records D2H
struct RecPos {   // where a record lies in the text: found by the scan, read by the pack stage
    uint64_t start;      // of the '@'
    uint32_t hdr, seq;   // bytes of the header line and of the sequence line (without their line breaks)
};


struct IngestCache {   // the two batches in flight of a pa_process_reads call or a record stream; parked on the index in between (pa_common.hpp)
    BatchCtx ctx[2];
    std::vector<RecPos> rec_pos;   // 16 bytes per record of the file: kept, or every call would page 256 MB in again
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

inline int batch_launch(pa_index* idx, BatchCtx& c, hipStream_t stream) {
    PA_INGEST_HIP_OK(hipMemcpyAsync(c.d_ascii, c.h_ascii, c.ascii_bytes, hipMemcpyHostToDevice, stream));
    PA_INGEST_HIP_OK(hipMemcpyAsync(c.d_soff, c.h_soff, (c.n + 1) * 8, hipMemcpyHostToDevice, stream));
    const int e0 = pa_encode_reads_device(idx, (const uint8_t*)c.d_ascii, (const uint64_t*)c.d_soff, c.n, c.wpr, (uint64_t*)c.d_tiles, (uint32_t*)c.d_lens, stream);   // :450
    if (e0 != PA_OK) return e0;
    const int e = pa_map_batch_device(idx, (const uint64_t*)c.d_tiles, (const uint32_t*)c.d_lens, c.n, c.wpr, PA_DEFAULT_ALLOWED_MISMATCHES,
                                      (pa_read_result*)c.d_results, (uint32_t*)c.d_arena, c.arena_entries, nullptr, stream);
    if (e != PA_OK) return e;
    PA_INGEST_HIP_OK(hipMemcpyAsync(c.h_results, c.d_results, c.n * sizeof(pa_read_result), hipMemcpyDeviceToHost, stream));
    return PA_OK;
}

// waits for the batch; an arena that turned out too small is regrown and the batch mapped again; the ids of the classes that
// are no index classes come to the host
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
    c.h_arena.resize(used + 1);
    if (used) PA_INGEST_HIP_OK(hipMemcpy(c.h_arena.data(), c.d_arena, used * 4, hipMemcpyDeviceToHost));
    return PA_OK;
}

// records [a, b) of a finished batch as the reference prints them (:490): (flag, "id", [ids], coverage), flag by the rule of
// :455; returns the number of flagged reads. Ids at text + rec.id_off; classes returned by reference come from the index's
// table of rendered classes (index_host_class_text).
inline uint64_t format_records(const BatchCtx& c, uint64_t a, uint64_t b, const char* text, const uint64_t* cls_off, const char* cls_txt, TextBuf& buf) {
    uint64_t nflag = 0;
    // a class returned by reference is copied from the index's table of rendered classes (one random read into tens of MB behind
    // an offset table that stays in the cache): prefetched a few reads ahead, or every read would wait for the miss
    constexpr uint64_t PF_REF = 16, PF_IDS = 8, PF_TEXT = 12;
    for (uint64_t i = a; i < b; ++i) {
        if (i + PF_TEXT < b) __builtin_prefetch(text + c.recs[i + PF_TEXT].id_off);   // the read's id: a line of a multi-GB text last touched by the pack stage
        if (i + PF_REF < b) {
            const uint32_t off = c.h_results[i + PF_REF].class_off;
            if (off & PA_CLASS_REF) __builtin_prefetch(cls_off + (off & ~PA_CLASS_REF));
        }
        if (i + PF_IDS < b) {
            const pa_read_result& q = c.h_results[i + PF_IDS];
            if (q.class_off & PA_CLASS_REF) {
                const char* t = cls_txt + cls_off[q.class_off & ~PA_CLASS_REF];
                __builtin_prefetch(t);
                if (q.class_len > 7) __builtin_prefetch(t + 64);
            } else if (q.class_len) __builtin_prefetch(c.h_arena.data() + q.class_off);
        }
        const pa_read_result& r = c.h_results[i];
        const bool mapped_read = r.mismatches & PA_MAPPED_BIT;
        const bool flag = mapped_read && r.coverage >= PA_READ_COVERAGE_THRESHOLD && r.class_len == 0;   // :455
        nflag += flag;
        char* const base = buf.room(6 * (size_t)c.recs[i].id_len + 12 * (size_t)r.class_len + 64);
        char* o = flag ? put_lit(base, "(true, ") : put_lit(base, "(false, ");
        o = debug_id(o, text + c.recs[i].id_off, c.recs[i].id_len);
        o = put_lit(o, ", [");
        if (r.class_off & PA_CLASS_REF) {
            const uint32_t cid = r.class_off & ~PA_CLASS_REF;
            const size_t n = (size_t)(cls_off[cid + 1] - cls_off[cid]);
            memcpy(o, cls_txt + cls_off[cid], n);
            o += n;
        } else {
            const uint32_t* ids = c.h_arena.data() + r.class_off;
            for (uint32_t j = 0; j < r.class_len; ++j) {
                if (j) { *o++ = ','; *o++ = ' '; }
                o = put_u32(o, ids[j]);
            }
        }
        o = put_lit(o, "], ");
        o = put_u32(o, mapped_read ? r.coverage : 0u);   // None -> (false, id, [], 0) (:461)
        *o++ = ')';
        *o++ = '\n';
        buf.len += (size_t)(o - base);
    }
    return nflag;
}

}  // namespace ingest
}  // namespace pa
