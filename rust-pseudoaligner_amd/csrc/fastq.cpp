// Host ingest pipeline mirroring process_reads (src/pseudoaligner.rs:420-514): FASTQ in, one Rust-Debug-formatted tuple
// per read out, in INPUT order. The reference's per-record mutex around the reader (src/utils.rs:152-157), bounded
// channel (:430,:464) and serial println consumer (:490) become a batch pipeline whose stages overlap:
//
//   scan     the FASTQ file is memory-mapped; all host threads count line breaks in their byte range, then record the
//            start of every 4-line record (no thread ever parses a byte twice, nobody takes a lock)
//   pack     per batch: threads 2-bit pack their reads straight into pinned tiles (the layout the kernel reads)
//   GPU      one stream: tiles H2D -> pa_map_batch_device -> results D2H; runs while the host formats the previous batch
//            and packs the next one
//   format   threads render "(flag, id, [ids], coverage)" (:455-461,:490) into per-thread buffers; classes returned by
//            reference are read from the host copy of the class table
//   write    a writer thread streams the buffers to the output in order
//
// The flag rule of :455 is kept as is (true iff coverage >= 32 and the class is EMPTY).
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <emmintrin.h>
#include <sys/mman.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <condition_variable>
#include <cstdio>
#include <chrono>
#include <cstdlib>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

#include "ingest.hpp"
#include "pa_common.hpp"

using namespace pa;
using namespace pa::ingest;

namespace {

constexpr uint64_t DEFAULT_BATCH_READS = 2u << 20;   // (4 Mi reads left 1.3 GB of the mapping and 0.23 GB of text to the last, unoverlapped batch: 33 ms of tear-down per 8 M reads against 12)   // PA_INGEST_BATCH overrides (tests exercise the batch seams with small values)


// in-order writer: pieces of text (the batches' rendered tuples, in pinned memory) are written by a dedicated thread; the owner of a
// piece waits for its job before it overwrites the bytes
class Writer {
public:
    explicit Writer(FILE* f) : f_(f), th_([this] { loop(); }) {}
    uint64_t push(const char* p, size_t n) {   // returns the job's number (1, 2, ...)
        std::lock_guard<std::mutex> g(mu_);
        q_.push_back({p, n});
        cv_.notify_one();
        return ++pushed_;
    }
    void wait(uint64_t job) {                  // until job `job` has been written
        std::unique_lock<std::mutex> g(mu_);
        room_.wait(g, [&] { return written_ >= job; });
    }
    bool finish() {
        { std::lock_guard<std::mutex> g(mu_); done_ = true; }
        cv_.notify_one();
        th_.join();
        return ok_;
    }

private:
    void loop() {
        for (;;) {
            std::pair<const char*, size_t> job;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [this] { return done_ || !q_.empty(); });
                if (q_.empty()) return;
                job = q_.front();
                q_.pop_front();
            }
            if (ok_ && job.second && fwrite(job.first, 1, job.second, f_) != job.second) ok_ = false;
            { std::lock_guard<std::mutex> g(mu_); ++written_; }
            room_.notify_all();
        }
    }
    FILE* f_;
    std::mutex mu_;
    std::condition_variable cv_, room_;
    std::deque<std::pair<const char*, size_t>> q_;
    uint64_t pushed_ = 0, written_ = 0;
    bool done_ = false, ok_ = true;
    std::thread th_;
};


const char* line_end(const char* p, const char* end) {
    const char* e = (const char*)memchr(p, '\n', (size_t)(end - p));
    return e ? e : end;
}

// Line breaks, 64 bytes at a time (SSE2, part of every x86-64): FASTQ lines are short — a header, a '+' — and one memchr call per
// line costs more than the bytes it looks at.
inline uint64_t nl_mask64(const char* p) {
    const __m128i nl = _mm_set1_epi8('\n');
    const uint64_t m0 = (uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i*)p), nl));
    const uint64_t m1 = (uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i*)(p + 16)), nl));
    const uint64_t m2 = (uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i*)(p + 32)), nl));
    const uint64_t m3 = (uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i*)(p + 48)), nl));
    return m0 | (m1 << 16) | (m2 << 32) | (m3 << 48);
}
uint64_t count_newlines(const char* d, uint64_t a, uint64_t b) {
    uint64_t c = 0, i = a;
    for (; i + 64 <= b; i += 64) c += (uint64_t)__builtin_popcountll(nl_mask64(d + i));
    for (; i < b; ++i) c += d[i] == '\n';
    return c;
}
// fn(position of a line break) for every line break in [from, to), in order, until fn returns false
template <class F>
void for_each_newline(const char* d, uint64_t from, uint64_t to, F&& fn) {
    uint64_t i = from;
    for (; i + 64 <= to; i += 64)
        for (uint64_t m = nl_mask64(d + i); m; m &= m - 1)
            if (!fn(i + (uint64_t)__builtin_ctzll(m))) return;
    for (; i < to; ++i)
        if (d[i] == '\n' && !fn(i)) return;
}

#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(PA_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); } while (0)

// The parallel scan finds records by counting lines, four to a record. A file that does not have that shape — sequence or
// qualities wrapped over several lines, which bio's fastq::Reader (the reference's reader) accepts — is first rewritten
// into it by this sequential reader: header line '@...', sequence lines up to the line that starts with '+', then as many
// quality lines as there were sequence lines (what bio 1.5's Reader::read does). Returns false (with the 0-based record
// number) when the text is no FASTQ at all; trailing blank lines are tolerated as everywhere in this file.
bool normalize_fastq(const char* d, uint64_t n, std::vector<char>& out, uint64_t& bad_rec) {
    out.clear();
    out.reserve(n + 16);
    const char* p = d;
    const char* const end = d + n;
    auto next_line = [&](const char*& b, const char*& e) -> bool {   // [b, e) without the line break; false at the end of the text
        if (p >= end) return false;
        b = p;
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        e = nl ? nl : end;
        p = nl ? nl + 1 : end;
        if (e > b && e[-1] == '\r') --e;
        return true;
    };
    uint64_t rec = 0;
    const char *b, *e;
    for (;;) {
        if (!next_line(b, e)) return true;
        if (b == e) {   // blank line: fine only if nothing but blank lines follows
            while (next_line(b, e))
                if (b != e) { bad_rec = rec; return false; }
            return true;
        }
        if (*b != '@') { bad_rec = rec; return false; }
        out.insert(out.end(), b, e);
        out.push_back('\n');
        uint64_t seq_lines = 0;
        bool plus = false;
        while (next_line(b, e)) {
            if (b != e && *b == '+') { plus = true; break; }
            out.insert(out.end(), b, e);
            ++seq_lines;
        }
        if (!plus) { bad_rec = rec; return false; }   // the text ends inside a record
        out.push_back('\n');
        out.push_back('+');
        out.push_back('\n');
        for (uint64_t i = 0; i < seq_lines; ++i) {
            if (!next_line(b, e)) break;   // (bio leaves the qualities short; the reference never looks at them)
            out.insert(out.end(), b, e);
        }
        out.push_back('\n');
        ++rec;
    }
}

struct FastqText {   // the text of a FASTQ file as the scan sees it: the mapped file, an inflated gzip stream, or the rewritten records
    const char* data = nullptr;
    uint64_t fsize = 0;
    bool mapped = false;
    std::vector<char> inflated;     // a gzip'ed FASTQ (utils::open_with_gz, src/utils.rs:45-57) is inflated into memory first
    std::vector<char> normalized;   // the text rewritten into four-line records, if it did not have that shape
    uint64_t off = 0;               // text before this offset has been handed out as records (windowed scan of pa_process_reads)
    const char* map_base = nullptr; // the mapping as mmap returned it (data moves on when the rest of a file is rewritten)
    uint64_t map_size = 0;
    void release() {
        if (mapped) munmap((void*)map_base, map_size);
        mapped = false;
        data = nullptr;
        fsize = 0;
    }
};

int open_fastq(const char* fastq_path, FastqText& t) {
    const char*& data = t.data;
    uint64_t& fsize = t.fsize;
    bool& mapped = t.mapped;
    std::vector<char>& inflated = t.inflated;
    // ---- map the file ----
    const int fd = open(fastq_path, O_RDONLY);
    if (fd < 0) return fail(PA_ERR_IO, "cannot open %s: %s", fastq_path, strerror(errno));
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return fail(PA_ERR_IO, "cannot stat %s: %s", fastq_path, strerror(errno)); }
    fsize = (uint64_t)st.st_size;
    unsigned char magic[2] = {0, 0};
    const bool gz = fsize >= 2 && pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
    if (gz) {
        gzFile g = gzdopen(dup(fd), "rb");   // every member of a multi-member file, like flate2's MultiGzDecoder
        if (!g) { close(fd); return fail(PA_ERR_IO, "cannot read %s as gzip", fastq_path); }
        (void)gzbuffer(g, 1 << 20);
        inflated.resize(std::max<uint64_t>(fsize * 4, 1 << 20));
        uint64_t have = 0;
        for (;;) {
            if (have == inflated.size()) inflated.resize(inflated.size() * 2);
            const int got = gzread(g, inflated.data() + have, (unsigned)std::min<uint64_t>(inflated.size() - have, 1u << 30));
            if (got < 0) { int e = 0; const char* why = gzerror(g, &e); gzclose(g); close(fd); return fail(PA_ERR_FORMAT, "%s: corrupt gzip stream: %s", fastq_path, why); }
            if (got == 0) break;
            have += (uint64_t)got;
        }
        {
            int e = Z_OK;
            const char* why = gzerror(g, &e);   // a truncated member hands out what it has and reports Z_BUF_ERROR
            if (e != Z_OK && e != Z_STREAM_END) { const int rc_ = fail(PA_ERR_FORMAT, "%s: corrupt gzip stream: %s", fastq_path, why); gzclose(g); close(fd); return rc_; }
        }
        gzclose(g);
        fsize = have;
        data = inflated.data();
    } else if (fsize) {
        void* m = mmap(nullptr, fsize, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { close(fd); return fail(PA_ERR_IO, "cannot map %s: %s", fastq_path, strerror(errno)); }
        (void)madvise(m, fsize, MADV_SEQUENTIAL);
        data = (const char*)m;
        mapped = true;
        t.map_base = data;
        t.map_size = fsize;
    }
    close(fd);
    return PA_OK;
}

constexpr int SCAN_NOT_FOUR_LINE = 1, SCAN_WINDOW_TOO_SMALL = 2;   // (what a window's scan may answer besides a pa_status)

// Records of the text t.data[t.off, t.off + avail): line breaks per byte range, then what every line is (scan of pa_process_reads;
// pa_fastq_scan_host runs it alone, over the whole text). rec_pos[i] = where record i lies, counted from t.data + t.off.
//   last   the text ends with these bytes: trailing blank lines are tolerated, a last record may lack its quality line, and a text that
//          is not in four-line shape is rewritten once (t.data / t.fsize / t.off then describe the rewritten text) and scanned again
//   !last  a WINDOW of a longer text: only whole records are taken, *consumed = the bytes they are (the next window starts behind them).
//          SCAN_NOT_FOUR_LINE: a record of the window lacks its '@' or '+' — the caller scans the rest of the file as one text;
//          SCAN_WINDOW_TOO_SMALL: the window holds no whole record
// rec_base: records before this text (error messages count records from the file's first).
int scan_fastq(const char* fastq_path, FastqText& t, uint64_t avail, bool last, uint64_t rec_base, Pool& pool, std::vector<RecPos>& rec_pos, uint64_t& nrec,
               std::vector<std::vector<uint32_t>>& brk, uint64_t* consumed) {
    bool& mapped = t.mapped;
    std::vector<char>& inflated = t.inflated;
    std::vector<char>& normalized = t.normalized;
    const int T = pool.size();
    int rc = PA_OK;
    nrec = 0;
    if (consumed) *consumed = 0;
    // ---- scan: line breaks per byte range, then the start of every fourth line ----
    for (int attempt = 0; attempt < 2; ++attempt) {
        const char* const data = t.data + t.off;
        const uint64_t fsize = attempt == 0 ? avail : t.fsize - t.off;
        std::atomic<uint64_t> odd_record{~0ull};   // first record whose first line lacks the '@' or whose third the '+'
        rc = PA_OK;
        // ONE pass over the text: every range notes where its line breaks are (32-bit offsets from the range's start: ranges are
        // kept below 2 GiB); what every line is follows from these lists alone, once the prefix sum has given each range its first
        // line number (a second pass over the text cost as much as the first: 2.5 GB per 8 M reads)
        const int R = (int)std::min<uint64_t>(std::max<uint64_t>((uint64_t)T * 4, fsize / (1ull << 30) + 1), fsize / (1 << 16) + 1);
        std::vector<uint64_t> nl((size_t)R + 1, 0);
        if (brk.size() < (size_t)R) brk.resize((size_t)R);   // (the caller keeps the lists between calls: 128 MB per 8 M reads that would otherwise be paged in again)
        auto range = [&](int r, uint64_t& a, uint64_t& b) { a = fsize * (uint64_t)r / R; b = fsize * (uint64_t)(r + 1) / R; };
        pool.run(R, [&](int r) {
            uint64_t a, b;
            range(r, a, b);
#ifdef MADV_POPULATE_READ
            // a fresh mapping of a file in the page cache costs a minor fault per 4 KiB page on first touch (0.6 M of them for 8 M reads):
            // let the kernel fill this range's page table entries in one call instead (Linux >= 5.14; elsewhere the faults simply happen)
            if (mapped && attempt == 0) {
                const uint64_t pa_ = (uint64_t)(data + a - t.map_base) & ~4095ull;   // (page-aligned in the mapping)
                (void)madvise((void*)(t.map_base + pa_), (size_t)((uint64_t)(data + b - t.map_base) - pa_), MADV_POPULATE_READ);
            }
#endif
            std::vector<uint32_t>& v = brk[(size_t)r];
            v.clear();
            v.reserve((size_t)((b - a) / 64 + 16));   // (FASTQ of 150-base reads: one line break per ~79 bytes)
            for_each_newline(data, a, b, [&](uint64_t e) { v.push_back((uint32_t)(e - a)); return true; });
            nl[(size_t)r + 1] = v.size();
        });
        for (int r = 0; r < R; ++r) nl[(size_t)r + 1] += nl[(size_t)r];
        uint64_t content_lines = 0;
        if (!last) {   // a window: the whole records among its lines; the next window starts behind their last line break
            nrec = nl[(size_t)R] / 4;
            if (nrec == 0) return SCAN_WINDOW_TOO_SMALL;
            content_lines = 4 * nrec;
            int r = 0;
            while (nl[(size_t)r + 1] < content_lines) ++r;
            uint64_t a, b;
            range(r, a, b);
            *consumed = a + brk[(size_t)r][(size_t)(content_lines - 1 - nl[(size_t)r])] + 1;
        } else {
        // trailing empty lines are tolerated: lines = line breaks before the last content byte + 1
        uint64_t tail = fsize, trailing_nl = 0;
        while (tail > 0 && (data[tail - 1] == '\n' || data[tail - 1] == '\r')) { trailing_nl += data[tail - 1] == '\n'; --tail; }
        content_lines = tail ? nl[(size_t)R] - trailing_nl + 1 : 0;
        if (content_lines % 4 == 3) {
            // a last record with an EMPTY sequence: its empty quality line looks like a trailing blank line (or is missing
            // altogether when the file ends after the '+'; bio's reader reads nothing there and hands the record out). Taken as
            // that record when the text ends "...\n<empty line>\n+..."
            uint64_t ls = tail;                                            // start of the last content line
            while (ls > 0 && data[ls - 1] != '\n') --ls;
            if (data[ls] == '+' && ls >= 2) {
                uint64_t pe = ls - 1;                                      // the line break that ends the sequence line
                if (pe > 0 && data[pe - 1] == '\r') --pe;
                if (pe > 0 && data[pe - 1] == '\n') content_lines += 1;    // the sequence line is empty
            }
        }
        if (content_lines % 4 != 0)
            rc = fail(PA_ERR_FORMAT, "%s: malformed FASTQ record %llu (file ends inside a record)", fastq_path, (unsigned long long)(rec_base + content_lines / 4));
        nrec = content_lines / 4;
        if (consumed) *consumed = fsize;
        }
        if (rc == PA_OK && nrec) {
            rec_pos.resize(nrec);   // (every field is written below: each line starts in exactly one range)
            RecPos* const rp = rec_pos.data();
            pool.run(R, [&](int r) {
                uint64_t a, b;
                range(r, a, b);
                auto odd = [&](uint64_t rec) {
                    uint64_t cur = odd_record.load();
                    while (rec < cur && !odd_record.compare_exchange_weak(cur, rec)) {}
                };
                // line li = bytes [p, e) (e = its line break, or the end of the text)
                auto line = [&](uint64_t p, uint64_t e, uint64_t li) {
                    const uint64_t rec = li >> 2;
                    if (rec >= nrec) return;
                    switch (li & 3) {
                        case 0: {
                            rp[rec].start = p;
                            rp[rec].hdr = (uint32_t)std::min<uint64_t>(e - p, 0xFFFFFFFFull);
                            if (data[p] != '@') odd(rec);
                            // record.id() (:456) = header[1..].trim_end().splitn(2, ' ').next() in bio 1.5: cut at the first SPACE only (a tab stays
                            // part of the id), after trailing white space was trimmed. Found here, while the header's bytes are in the cache
                            uint64_t hend = e, ide = p + 1;
                            while (hend > p + 1 && (data[hend - 1] == '\r' || data[hend - 1] == ' ' || data[hend - 1] == '\t' || data[hend - 1] == '\n')) --hend;
                            while (ide < hend && data[ide] != ' ') ++ide;
                            rp[rec].id_len = e > p ? (uint32_t)std::min<uint64_t>(ide - (p + 1), 0xFFFFFFFFull) : 0u;
                            break;
                        }
                        case 1: {
                            const uint64_t len = e - p;
                            rp[rec].seq = (uint32_t)std::min<uint64_t>(len, 0xFFFFFFFFull);
                            rp[rec].seq_len = (uint32_t)std::min<uint64_t>(len && data[e - 1] == '\r' ? len - 1 : len, 0xFFFFFFFFull);
                            break;
                        }
                        case 2: if (data[p] != '+') odd(rec); break;
                        default: break;
                    }
                };
                // the lines that START in [a, b): the first one begins after the first line break at or after a - 1. Their ends are
                // this range's line breaks and, for the last of them, the first line break of the ranges behind it
                uint64_t li = nl[(size_t)r], p = a;
                bool started = a == 0 || data[a - 1] == '\n';
                if (!started) li += 1;   // (the line break that ends the straddling line is counted in this range or a later one)
                bool open = true;        // a line that started in this range still waits for its end
                for (int q = r; q < R && open; ++q) {
                    uint64_t qa, qb;
                    range(q, qa, qb);
                    for (const uint32_t rel : brk[(size_t)q]) {
                        const uint64_t e = qa + rel;
                        if (!started) { started = true; p = e + 1; if (p >= b) { open = false; break; } continue; }
                        line(p, e, li);
                        p = e + 1;
                        ++li;
                        if (p >= b) { open = false; break; }
                    }
                }
                if (open && started && p < b && p < fsize) line(p, fsize, li);   // a last line without a line break
            });
        }
        if (rc == PA_OK && odd_record.load() == ~0ull) break;   // four lines to a record, markers in place
        if (!last) return SCAN_NOT_FOUR_LINE;
        if (attempt == 1) {
            if (rc == PA_OK) rc = fail(PA_ERR_FORMAT, "%s: malformed FASTQ record %llu", fastq_path, (unsigned long long)(rec_base + odd_record.load()));
            break;
        }
        // not that shape: wrapped sequence / quality lines? rewrite and scan again
        uint64_t bad = 0;
        std::vector<char> rewritten;
        if (!normalize_fastq(data, fsize, rewritten, bad)) {
            rc = fail(PA_ERR_FORMAT, "%s: malformed FASTQ record %llu (no '@' header, or the file ends inside the record)", fastq_path, (unsigned long long)(rec_base + bad));
            break;
        }
        t.release();   // (the mapping, if the text was one)
        inflated = std::vector<char>();
        normalized.swap(rewritten);
        t.data = normalized.data();
        t.fsize = normalized.size();
        t.off = 0;
        nrec = 0;
    }

    return rc;
}

// The text in WINDOWS: a mapped file is scanned a window at a time (PA_INGEST_WINDOW bytes), so that pa_process_reads has its first batch on
// the GPU while the rest of the file is still being scanned; a window ends behind its last whole record. A text held in memory (gzip), a
// window that is not in four-line shape (then: the rest of the file as one text, rewritten) and the last window are scanned as one text.
struct WindowScan {
    FastqText& text;
    uint64_t window = 256ull << 20;
    bool windowed, done = false;
    uint64_t nrec = 0;            // records of the current window
    const char* base = nullptr;   // the window's text: rec_pos counts from here
    uint64_t size = 0;            // its bytes
    uint64_t abs = ~0ull;         // its offset in the file's mapping (~0: not part of one)
    explicit WindowScan(FastqText& t) : text(t), windowed(t.mapped) {
        if (const char* v = getenv("PA_INGEST_WINDOW")) { const long long x = atoll(v); if (x >= 1) window = (uint64_t)x; }
    }
    // the next window with records in it (nrec = 0: the text has ended). records_before: records of the windows before (error messages)
    int next(const char* fastq_path, uint64_t records_before, Pool& pool, std::vector<RecPos>& rec_pos, std::vector<std::vector<uint32_t>>& brk) {
        nrec = 0;
        while (!done) {
            const uint64_t rest = text.fsize - text.off;
            if (rest == 0) { done = true; break; }
            const uint64_t avail = windowed ? std::min<uint64_t>(window, rest) : rest;
            const bool last = avail == rest;
            uint64_t consumed = 0;
            const int r = scan_fastq(fastq_path, text, avail, last, records_before, pool, rec_pos, nrec, brk, &consumed);
            if (r == SCAN_WINDOW_TOO_SMALL) { window *= 2; continue; }   // (a record longer than the window)
            if (r == SCAN_NOT_FOUR_LINE) { windowed = false; continue; }
            if (r != PA_OK) return r;
            base = text.data + text.off;
            size = consumed;
            abs = text.mapped ? (uint64_t)(base - text.map_base) : ~0ull;
            text.off += consumed;
            if (last) done = true;
            if (nrec) break;
        }
        return PA_OK;
    }
};

}  // namespace

// The scan stage of pa_process_reads by itself (no GPU): how many records the text holds and where their header and sequence
// lines lie. Offsets refer to the text as scanned: the file itself (*text_kind 0), the inflated gzip stream (1) or the text
// rewritten into four-line records (2, wrapped input).
extern "C" int pa_fastq_scan_host(const char* fastq_path, int num_threads, uint64_t* n_records, uint64_t* starts, uint32_t* header_len,
                                  uint32_t* seq_len, uint64_t capacity, int* text_kind) {
    if (!fastq_path || !n_records) return fail(PA_ERR_INVALID_ARG, "null argument");
    *n_records = 0;
    FastqText text;
    int rc = open_fastq(fastq_path, text);
    if (rc != PA_OK) return rc;
    const bool was_gz = !text.mapped && text.fsize != 0;
    Pool pool(num_threads < 1 ? 1 : num_threads);
    std::vector<RecPos> rec_pos;
    std::vector<std::vector<uint32_t>> brk;
    // window by window, as pa_process_reads walks the text. Offsets refer to the text the LAST window was part of: when a window turns out
    // not to be in four-line shape the rest of the file is rewritten, and records from there on lie in the rewritten text (*text_kind 2)
    uint64_t nrec = 0;
    WindowScan ws(text);
    while (rc == PA_OK) {
        rc = ws.next(fastq_path, nrec, pool, rec_pos, brk);
        if (rc != PA_OK || ws.nrec == 0) break;
        const uint64_t at = (uint64_t)(ws.base - text.data);   // of the window in the text it belongs to
        for (uint64_t i = 0; i < ws.nrec && nrec + i < capacity; ++i) {
            if (starts) starts[nrec + i] = at + rec_pos[i].start;
            if (header_len) header_len[nrec + i] = rec_pos[i].hdr;
            if (seq_len) seq_len[nrec + i] = rec_pos[i].seq_len;   // a CR before the line break is not sequence
        }
        nrec += ws.nrec;
    }
    if (rc == PA_OK) {
        *n_records = nrec;
        if (text_kind) *text_kind = !text.normalized.empty() ? 2 : was_gz ? 1 : 0;
    }
    text.release();
    return rc;
}

extern "C" int pa_process_reads(pa_index* idx, const char* fastq_path, const char* out_path, int num_threads, uint64_t* n_reads_out,
                                uint64_t* n_flagged_out) {
    if (!idx || !fastq_path || !out_path) return fail(PA_ERR_INVALID_ARG, "null argument");
    const double t_enter = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    if (num_threads < 1) num_threads = 1;
    if (n_reads_out) *n_reads_out = 0;
    if (n_flagged_out) *n_flagged_out = 0;
    const uint32_t *h_ec = nullptr, *h_class_ref = nullptr;
    int device = 0;
    index_host_classes(idx, &h_ec, &h_class_ref, &device);
    HIP_OK(hipSetDevice(device));

    // ---- map the file ----
    FastqText text;
    {
        const int orc = open_fastq(fastq_path, text);
        if (orc != PA_OK) return orc;
    }
    FILE* out = strcmp(out_path, "-") == 0 ? stdout : fopen(out_path, "wb");
    if (!out) { text.release(); return fail(PA_ERR_IO, "cannot create %s: %s", out_path, strerror(errno)); }
    // a private 4 MiB stdio buffer only for a file this function opened (and closes before the buffer dies); the process-wide
    // stdout keeps its own buffering: handing it a function-local buffer would leave it dangling after the return
    std::vector<char> obuf(out != stdout ? (size_t)1 << 22 : 0);
    if (out != stdout) setvbuf(out, obuf.data(), _IOFBF, obuf.size());

    uint64_t BATCH_READS = DEFAULT_BATCH_READS;
    if (const char* v = getenv("PA_INGEST_BATCH")) { const long long x = atoll(v); if (x >= 64) BATCH_READS = (uint64_t)x / 64 * 64; }
    const bool verbose = getenv("PA_VERBOSE") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_text_wait = 0, t_scan = 0, t_pack = 0, t_finish = 0, t_launch = 0, t_format = 0, t_pack_rec = 0, t_pack_alloc = 0, t_pack_tiles = 0, t_push = 0;
    const double t_begin = now();
    Pool pool(num_threads);
    const int T = pool.size();
    int rc = PA_OK;
    uint64_t nrec = 0;
    IngestCache* cache = static_cast<IngestCache*>(index_take_ingest_cache(idx));   // buffers of the previous call, if any
    if (!cache) cache = new IngestCache();
    std::vector<RecPos>& rec_pos = cache->rec_pos;

    // ---- the text in windows (WindowScan): the scan of a window runs inside the pack stage of its first batch ----
    WindowScan ws(text);
    uint64_t wnext = 0;   // the next record of the window to be packed
    // ---- batches ----
    BatchCtx* const ctx = cache->ctx;
    cache->idx = idx;
    if (rc == PA_OK && !cache->stream && hipStreamCreate(&cache->stream) != hipSuccess) { cache->stream = nullptr; rc = fail(PA_ERR_HIP, "hipStreamCreate failed"); }
    const hipStream_t stream = cache->stream;
    Writer writer(out);
    uint64_t flagged = 0, next_report = 1000000, reported = 0;

    // buffers are sized once, for the batches this FILE will need — a whole batch when the text goes on behind the window (the record
    // stream shares the parked buffers and pushes whole batches: sized by the window alone they were regrown there, pinned memory and all)
    auto ensure = [&](BatchCtx& c, uint64_t n, uint32_t wpr) -> int {
        const uint64_t est = ws.size ? (uint64_t)((double)ws.nrec * (double)(text.fsize - (uint64_t)(ws.base - text.data)) / (double)ws.size) : ws.nrec;
        return batch_ensure(idx, c, n, wpr, std::min<uint64_t>(BATCH_READS, std::max<uint64_t>(ws.nrec, est)));
    };

    // the next batch of the text into c: its records (of the current window, or the next one: scanned now), parsed + packed (parallel over
    // whole tiles). c.n = 0 at the end of the text
    auto pack = [&](BatchCtx& c) -> int {
        c.n = 0;
        if (wnext == ws.nrec) {
            const double ts = now();
            const int wr = ws.next(fastq_path, nrec, pool, rec_pos, cache->brk);
            t_scan += now() - ts;
            wnext = 0;
            if (wr != PA_OK) return wr;
            if (ws.nrec == 0) return PA_OK;
        }
        const char* const data = ws.base;
        const uint64_t fsize = ws.size;
        c.first = wnext;
        c.n = std::min<uint64_t>(BATCH_READS, ws.nrec - wnext);
        wnext += c.n;
        nrec += c.n;
        c.text_abs = ws.abs == ~0ull ? ~0ull : ws.abs + rec_pos[c.first].start;
        double t0 = now();
        c.recs.resize(c.n);
        std::vector<uint32_t> tmax((size_t)T * 4, 0);
        const int ntask = T * 4;
        pool.run(ntask, [&](int t) {   // records + lengths: from what the scan noted, no byte of the text is touched here
            uint32_t mx = 0;
            for (uint64_t i = c.n * (uint64_t)t / ntask; i < c.n * (uint64_t)(t + 1) / ntask; ++i) {
                const RecPos& rp = rec_pos[c.first + i];
                Record& rec = c.recs[i];
                rec.id_off = rp.start + 1;
                rec.id_len = rp.id_len;
                rec.seq_off = std::min<uint64_t>(rp.start + rp.hdr + 1, fsize);
                rec.seq_len = (uint32_t)std::min<uint64_t>(rp.seq_len, fsize - rec.seq_off);
                mx = std::max(mx, rec.seq_len);
            }
            tmax[(size_t)t] = mx;
        });
        uint32_t maxlen = 1;
        for (uint32_t m : tmax) maxlen = std::max(maxlen, m);
        if (maxlen > PA_MAX_READ_LEN) return fail(PA_ERR_UNSUPPORTED, "read longer than %u bases", PA_MAX_READ_LEN);
        c.wpr = pa_words_per_read(maxlen);
        std::vector<uint64_t> part;
        batch_offsets(pool, c, part);
        t_pack_rec += now() - t0; t0 = now();
        const int e = ensure(c, c.n, c.wpr);
        if (e != PA_OK) return e;
        t_pack_alloc += now() - t0; t0 = now();
        batch_gather_ascii(pool, c, data, part);   // the bytes of record.seq() (:449) for the GPU's DnaString::from_dna_string (:450)
        t_pack_tiles += now() - t0;
        return PA_OK;
    };

    auto launch = [&](BatchCtx& c) -> int { return batch_launch(idx, c, stream); };   // index.map_read (:451) for the whole batch
    auto finish = [&](BatchCtx& c) -> int { return batch_finish(idx, c, stream); };

    uint64_t unmapped_to = 0;   // bytes of the mapping already given back (page-aligned)
    // The text of the batches already written goes back to the kernel piece by piece (unmapping 5 GB of page-cache mapping is 0.09 s of
    // one thread's time): on a helper thread, so that no stage of the pipeline waits for it.
    std::thread unmapper;
    auto unmap_async = [&](uint64_t from, uint64_t to) {
        if (unmapper.joinable()) unmapper.join();
        const char* base = text.map_base;
        unmapper = std::thread([base, from, to] { (void)munmap((void*)(base + from), to - from); });
    };
    int format_rc = PA_OK;
    uint64_t text_job[2] = {0, 0};   // the writer's job that reads ctx[k].h_text (0: none)
    auto format = [&](BatchCtx& c, int k) {
        // the batch's tuples were rendered on the GPU (batch_finish -> render.hip) and are on their way to pinned memory: wait for them and
        // hand them to the writer where they are (finish() of the batch after next waits for that job before the buffer is written again)
        double tw = now();
        if ((format_rc = batch_text_wait(c)) != PA_OK) return;
        t_text_wait += now() - tw;
        const uint64_t keep_from = (text.mapped && c.text_abs != ~0ull) ? (c.text_abs & ~4095ull) : 0;   // nothing before this batch is read again
        if (keep_from > unmapped_to) { unmap_async(unmapped_to, keep_from); unmapped_to = keep_from; }
        flagged += c.flagged;
        reported += c.n;
        while (reported >= next_report) {   // :497-503
            fprintf(stderr, "\rDone Mapping %llu reads w/ Rate: %g", (unsigned long long)next_report,
                    (double)((float)flagged * 100.0f / (float)reported));
            next_report += 1000000;
        }
        text_job[k] = writer.push(c.h_text, c.text_bytes);
    };

    // pack(b) — and the scan of its window — overlaps GPU(b-1); format(b-1) overlaps GPU(b)
    bool have_prev = false;
    for (uint64_t b = 0; rc == PA_OK; ++b) {
        BatchCtx& cur = ctx[b & 1];
        BatchCtx& prev = ctx[(b + 1) & 1];
        const int kp = (int)((b + 1) & 1);
        double t0 = now();
        const double scan_before = t_scan;
        rc = pack(cur);
        t_pack += now() - t0 - (t_scan - scan_before); t0 = now();
        const bool have = rc == PA_OK && cur.n != 0;
        if (rc == PA_OK && have_prev) rc = finish(prev);
        t_finish += now() - t0; t0 = now();
        if (rc == PA_OK && have) {
            // the launch ends with the speculative copy of this batch's tuples into the context's pinned text buffer (batch_render_enqueue):
            // the writer must be done with what the buffer held before — the text of batch b - 2
            const int kc = (int)(b & 1);
            if (text_job[kc]) { const double tw = now(); writer.wait(text_job[kc]); t_push += now() - tw; text_job[kc] = 0; }
            rc = launch(cur);
        }
        t_launch += now() - t0; t0 = now();
        if (rc == PA_OK && have_prev) { format(prev, kp); rc = format_rc; }
        t_format += now() - t0;
        have_prev = have;
        if (!have) break;
    }
    {
        double* st = pa::ingest::last_stage_seconds();
        st[0] = t_scan; st[1] = t_pack; st[2] = t_finish; st[3] = t_launch; st[4] = t_format; st[5] = t_push; st[6] = now() - t_begin; st[7] = (double)nrec;
    }
    if (verbose)
        fprintf(stderr, "\n[pa ingest] %llu reads, %d threads: scan %.3f s, pack %.3f s (records %.3f, alloc %.3f, tiles %.3f), wait GPU %.3f s, launch %.3f s, text %.3f s (waiting for the GPU's tuples %.3f; writer wait %.3f), total %.3f s\n",
                (unsigned long long)nrec, T, t_scan, t_pack, t_pack_rec, t_pack_alloc, t_pack_tiles, t_finish, t_launch, t_format, t_text_wait, t_push, now() - t_begin);
    double t0 = now();
    if (stream) (void)hipStreamSynchronize(stream);   // (the stream stays with the parked buffers; IngestCache::destroy releases both)
    const double t_stream = now() - t0; t0 = now();
    const bool wrote = writer.finish();
    const double t_writer = now() - t0; t0 = now();
    if (rc == PA_OK && !wrote) rc = fail(PA_ERR_IO, "short write to %s", out_path);
    if (cache->rec_pos.capacity() > ((size_t)64 << 20)) { std::vector<RecPos>().swap(cache->rec_pos); std::vector<std::vector<uint32_t>>().swap(cache->brk); }   // (do not park more than 1 GB of it)
    if (rc == PA_OK) index_put_ingest_cache(idx, cache, IngestCache::destroy);   // the next call starts with warm buffers
    else IngestCache::destroy(cache);
    if (unmapper.joinable()) unmapper.join();
    if (text.mapped && text.map_size > unmapped_to) {   // the rest of the mapping (the last batch's text): nobody reads it any more; given back without making the caller wait
        const char* base = text.map_base;
        const uint64_t from = unmapped_to, to = text.map_size;
        text.mapped = false;
        std::thread([base, from, to] { (void)munmap((void*)(base + from), to - from); }).detach();
    }
    const double t_unmap = now() - t0; t0 = now();
    if (out != stdout) { if (fclose(out) != 0 && rc == PA_OK) rc = fail(PA_ERR_IO, "close %s: %s", out_path, strerror(errno)); }
    else fflush(stdout);
    if (verbose) {
        struct rusage ru;
        getrusage(RUSAGE_SELF, &ru);
        fprintf(stderr, "[pa ingest] process CPU so far: user %.2f s, system %.2f s, minor faults %ld\n", ru.ru_utime.tv_sec + 1e-6 * ru.ru_utime.tv_usec,
                ru.ru_stime.tv_sec + 1e-6 * ru.ru_stime.tv_usec, ru.ru_minflt);
    }
    if (verbose)
        fprintf(stderr, "[pa ingest] teardown: stream %.3f s, writer %.3f s, unmap %.3f s, close %.3f s; before the scan %.3f s\n", t_stream, t_writer, t_unmap,
                now() - t0, t_begin - t_enter);
    if (n_reads_out) *n_reads_out = reported;
    if (n_flagged_out) *n_flagged_out = flagged;
    return rc;
}

extern "C" int pa_process_reads_stage_seconds(double out[PA_INGEST_STAGES]) {
    if (!out) return fail(PA_ERR_INVALID_ARG, "null argument");
    memcpy(out, pa::ingest::last_stage_seconds(), sizeof(double) * PA_INGEST_STAGES);
    return PA_OK;
}
