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
#include <cmath>
#include <string>
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
    int fd = -1;                    // of a mapped file (kept open for the call)
    void release() {
        if (mapped) munmap((void*)map_base, map_size);   // (the whole mapping: nobody gives parts of it back any more)
        if (fd >= 0) close(fd);
        fd = -1;
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
        t.fd = fd;
        return PA_OK;
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

namespace {

// f32 as Rust's `{}` prints it (the progress line of :497-503): the shortest digits that read back as the same float, never an exponent
std::string rust_f32(float v) {
    if (v != v) return "NaN";
    if (v == 0.0f) return std::signbit(v) ? "-0" : "0";
    if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
    char buf[64];
    int prec = 0;
    for (; prec < 9; ++prec) {
        snprintf(buf, sizeof buf, "%.*e", prec, (double)v);
        if (strtof(buf, nullptr) == v) break;
    }
    // buf = [-]d[.ddd]e[+-]xx  ->  digits and a decimal exponent
    std::string digits;
    const char* q = buf;
    const bool neg = *q == '-';
    if (neg) ++q;
    for (; *q && *q != 'e'; ++q)
        if (*q != '.') digits.push_back(*q);
    const int exp10 = atoi(q + 1);
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    std::string out = neg ? "-" : "";
    const int nd = (int)digits.size();
    if (exp10 < 0) {
        out += "0.";
        out.append((size_t)(-exp10 - 1), '0');
        out += digits;
    } else if (exp10 + 1 >= nd) {
        out += digits;
        out.append((size_t)(exp10 + 1 - nd), '0');
    } else {
        out += digits.substr(0, (size_t)exp10 + 1) + "." + digits.substr((size_t)exp10 + 1);
    }
    return out;
}

// ---- process_reads over WINDOWS of raw text ----
// The host does not look at the text: worker threads copy a window of the file into pinned memory (out of the file's mapping, page tables
// filled and dropped piece by piece: TextPipe::read_piece), the window goes to HBM as it is on a copy stream, the GPU finds its records (fastq_scan.hip) and the encode /
// map / render kernels read sequences and ids where they lie. A lane has FOUR streams: copy (text in), scan (a window's records: waits for its
// text and the scan before, not for the kernels of the window before), the kernels' stream, and back (the tuples' 14 MB per window to the host):
// on one stream the chain scan | encode | map | render | copy back + two host round trips was as long as a window's copy, and every hiccup a gap on the link. A window ends where the file offset says, not where a record does: the
// scan reports how many bytes its whole records take, and the unfinished record is read once more as the HEAD of the next window.
// Windows are dealt round-robin to LANES — one per index handle (pa_process_reads_multi: the GPUs of a node; the same handle twice
// gives two streams on one GPU) — and their tuples are written in input order. What the GPU scan does not take goes through the
// host's tolerant scan (scan_fastq) and the same in-place kernels: the last piece of the text (a missing final line break, trailing
// blank lines, an empty last record) and text that is not in four-line shape (wrapped records: rewritten first).
constexpr int LANE_SLOTS = 4;            // windows of a lane in flight: read | scan | map + render | write
constexpr uint32_t FLAG_BUCKETS = PA_RENDER_FLAG_BUCKETS;

struct Lane {
    pa_index* idx = nullptr;
    int device = 0;
    IngestCache* cache = nullptr;
    hipStream_t stream = nullptr, copy = nullptr, scan = nullptr, back = nullptr;
    hipEvent_t last_h2d = nullptr;       // behind the lane's last window copy (an event of one of its slots)
    int64_t unfinished = -1;             // the window whose kernels were launched last on `stream` and have not been waited for
    uint64_t text_job[LANE_SLOTS] = {0, 0, 0, 0};   // the writer's job that reads the slot's pinned text (0: none)
};

struct Win {
    uint64_t id = 0;
    int lane = 0, slot = 0;
    uint64_t from = 0;          // text offset of its first record
    uint64_t first_read = 0;    // number of the reads before it
    bool launched = false;      // its kernels are on the lane's stream
};

struct TextPipe {
    const char* fastq_path;
    FastqText& text;
    Pool& pool;
    Writer& writer;
    std::vector<Lane>& lanes;
    uint64_t batch_reads;
    std::deque<Win> wins;       // launched or about to be, in order; the front is written first
    uint64_t next_id = 0, launched_reads = 0, reported = 0, flagged = 0, next_report = 1000000;
    double t_scan = 0, t_read = 0, t_wait = 0, t_launch = 0, t_text = 0, t_push = 0;
    uint64_t gpu_windows = 0, host_windows = 0, rescans = 0;

    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    int L() const { return (int)lanes.size(); }
    Lane& lane_of(uint64_t id) { return lanes[(size_t)(id % (uint64_t)L())]; }
    int slot_of(uint64_t id) const { return (int)((id / (uint64_t)L()) % LANE_SLOTS); }
    BatchCtx& ctx_of(const Win& w) { return lanes[(size_t)w.lane].cache->ctx[w.slot]; }
    int use(const Lane& l) { return hipSetDevice(l.device) == hipSuccess ? PA_OK : fail(PA_ERR_HIP, "hipSetDevice(%d) failed", l.device); }

    // bytes [off, off + len) of the text into dst (pinned): out of the file's mapping (read_piece), memcpy for text in memory (inflated gzip).
    // read_begin hands the pieces to the worker pool and returns; read_end waits for them (small reads are done at once by the caller)
    std::atomic<int> read_bad{0};
    bool read_async = false;
    // bytes of the mapping into a pinned window with streaming stores: the window is read next by the copy engine, never by this CPU, so no line of it
    // has to be fetched for ownership or kept in a cache
    static void copy_streaming(uint8_t* d, const uint8_t* s, size_t n) {
        size_t head = (64 - ((uintptr_t)d & 63)) & 63;
        if (head > n) head = n;
        memcpy(d, s, head);
        d += head; s += head; n -= head;
        const size_t body = n & ~(size_t)63;
        for (size_t i = 0; i < body; i += 64) {
            const __m128i v0 = _mm_loadu_si128((const __m128i*)(s + i)), v1 = _mm_loadu_si128((const __m128i*)(s + i + 16));
            const __m128i v2 = _mm_loadu_si128((const __m128i*)(s + i + 32)), v3 = _mm_loadu_si128((const __m128i*)(s + i + 48));
            _mm_stream_si128((__m128i*)(d + i), v0); _mm_stream_si128((__m128i*)(d + i + 16), v1);
            _mm_stream_si128((__m128i*)(d + i + 32), v2); _mm_stream_si128((__m128i*)(d + i + 48), v3);
        }
        _mm_sfence();
        memcpy(d + body, s + body, n - body);
    }
    bool use_pread = false;   // (knobs builds: the windows through pread, as before; tools/microbench/host_read.cpp has both side by side)
    void read_piece(uint64_t off, uint64_t len, uint8_t* dst, int t, int ntask) {
        const uint64_t a = off + len * (uint64_t)t / (uint64_t)ntask, b = off + len * (uint64_t)(t + 1) / (uint64_t)ntask;
        if (text.mapped && text.data == text.map_base && !use_pread) {
            // A file: out of its MAPPING. pread copies at 65 - 75 GB/s on 16 threads of the target host (one copy_to_user per page, the file's page-cache
            // lock) — 1.2 x the link, no margin — the same bytes out of the mapping at 127 GB/s INCLUDING the page tables of the piece, which are
            // filled in one call before the copy (MADV_POPULATE_READ, Linux 5.14; without it the copy faults them in: 115 GB/s) and dropped behind it
            // (a 100 GB file would otherwise keep 25 M entries mapped until the call ends)
            constexpr uint64_t PAGE = 4096;
            const bool big = b - a >= (256u << 10);
            if (big) {
                const uint64_t pa = a & ~(PAGE - 1), pb = std::min<uint64_t>((b + PAGE - 1) & ~(PAGE - 1), text.map_size);
#ifdef MADV_POPULATE_READ
                (void)madvise((void*)(text.map_base + pa), (size_t)(pb - pa), MADV_POPULATE_READ);
#else
                (void)madvise((void*)(text.map_base + pa), (size_t)(pb - pa), 22);
#endif
            }
            copy_streaming(dst + (a - off), (const uint8_t*)text.data + a, (size_t)(b - a));
            if (big) {
                const uint64_t qa = (a + PAGE - 1) & ~(PAGE - 1), qb = b & ~(PAGE - 1);
                if (qb > qa) (void)madvise((void*)(text.map_base + qa), (size_t)(qb - qa), MADV_DONTNEED);
            }
        } else if (text.mapped && text.fd >= 0 && text.data == text.map_base) {
            uint64_t p = a;
            while (p < b) {
                const ssize_t got = pread(text.fd, dst + (p - off), (size_t)(b - p), (off_t)p);
                if (got < 0 && errno == EINTR) continue;
                if (got <= 0) { read_bad.store(1); return; }
                p += (uint64_t)got;
            }
        } else memcpy(dst + (a - off), text.data + a, (size_t)(b - a));
    }
    void read_begin(uint64_t off, uint64_t len, uint8_t* dst) {
        read_async = false;
        if (len == 0) return;
        const uint64_t PIECE = 2ull << 20;
        const int ntask = (int)std::min<uint64_t>((len + PIECE - 1) / PIECE, 1u << 20);
        if (ntask == 1) { read_piece(off, len, dst, 0, 1); return; }
        read_async = true;
        pool.begin(ntask, [this, off, len, dst, ntask](int t) { read_piece(off, len, dst, t, ntask); });
    }
    int read_end() {
        if (read_async) { pool.end(); read_async = false; }
        return read_bad.exchange(0) ? fail(PA_ERR_IO, "%s: read failed: %s", fastq_path, strerror(errno)) : PA_OK;
    }
    int read_text(uint64_t off, uint64_t len, uint8_t* dst) {
        read_begin(off, len, dst);
        return read_end();
    }
    int read_small(uint64_t off, uint64_t len, uint8_t* dst) {   // by the caller itself, whatever the pool is doing (a window's head: <= 1 MiB)
        if (len) read_piece(off, len, dst, 0, 1);
        return read_bad.load() ? fail(PA_ERR_IO, "%s: read failed: %s", fastq_path, strerror(errno)) : PA_OK;
    }

    // the kernels of the window launched before on this lane's stream: waited for (they share the stream's launch context inside the index)
    int finish_lane(Lane& l) {
        if (l.unfinished < 0) return PA_OK;
        for (Win& w : wins)
            if ((int64_t)w.id == l.unfinished) {
                const double t0 = now();
                int e = use(l);
                if (e == PA_OK) e = batch_finish(l.idx, ctx_of(w), l.stream);
                t_wait += now() - t0;
                l.unfinished = -1;
                return e;
            }
        l.unfinished = -1;
        return PA_OK;
    }

    // the tuples of the windows whose kernels have been waited for: to the writer, in order, with the progress line of :497-503.
    // upto: also wait for the kernels of every window with id < upto (all of them at the end of the text)
    int retire_finished(uint64_t upto) {
        while (!wins.empty()) {
            Win& w = wins.front();
            if (!w.launched) break;   // (the window being launched right now)
            Lane& l = lanes[(size_t)w.lane];
            if (l.unfinished == (int64_t)w.id) {
                if (w.id >= upto) break;
                const int e = finish_lane(l);
                if (e != PA_OK) return e;
            }
            BatchCtx& c = ctx_of(w);
            const double t0 = now();
            int e = use(l);
            if (e == PA_OK) e = batch_text_wait(c);
            t_text += now() - t0;
            if (e != PA_OK) return e;
            uint64_t cum = 0, bucket = 0;
            while (next_report <= w.first_read + c.n) {   // :497-503: the counts of exactly the first 10^6 m reads
                if (bucket < FLAG_BUCKETS) cum += c.h_tot[1 + bucket];
                ++bucket;
                fprintf(stderr, "\rDone Mapping %llu reads w/ Rate: %s", (unsigned long long)next_report,
                        rust_f32((float)(flagged + cum) * 100.0f / (float)next_report).c_str());
                fflush(stderr);
                next_report += 1000000;
            }
            flagged += c.flagged;
            reported += c.n;
            l.text_job[w.slot] = writer.push(c.h_text, c.text_bytes);
            wins.pop_front();
        }
        return PA_OK;
    }

    // the slot of window `id`: the window that had it before (LANE_SLOTS windows of this lane earlier) has been handed to the writer
    int acquire(uint64_t id, BatchCtx** out) {
        const uint64_t span = (uint64_t)L() * LANE_SLOTS;
        if (id >= span) {
            const int e = retire_finished(id - span + 1);
            if (e != PA_OK) return e;
        }
        Lane& l = lane_of(id);
        BatchCtx& c = l.cache->ctx[slot_of(id)];
        int e = use(l);
        if (e != PA_OK) return e;
        if ((e = window_ensure_events(c)) != PA_OK) return e;
        c.back = l.back;
        *out = &c;
        return PA_OK;
    }

    // index.map_read (:451) for the window's records (c.n of them, c.wpr words each, found by the GPU scan or filled in by the host)
    int launch(Win& w) {
        Lane& l = lanes[(size_t)w.lane];
        BatchCtx& c = ctx_of(w);
        int e = finish_lane(l);
        if (e != PA_OK) return e;
        if ((e = retire_finished(0)) != PA_OK) return e;
        double t0 = now();
        if (l.text_job[w.slot]) { writer.wait(l.text_job[w.slot]); l.text_job[w.slot] = 0; }   // (the launch ends with the speculative copy of the tuples into the slot's pinned text)
        t_push += now() - t0; t0 = now();
        if ((e = use(l)) != PA_OK) return e;
        c.in_place = true;
        w.first_read = launched_reads;
        c.flag_mark = 1000000 - launched_reads % 1000000;
        if ((e = batch_ensure(l.idx, c, c.n, c.wpr, std::min<uint64_t>(std::max<uint64_t>(c.n + c.n / 8, 1 << 16), std::max<uint64_t>(batch_reads, c.n)))) != PA_OK) return e;
        if ((e = batch_launch(l.idx, c, l.stream)) != PA_OK) return e;
        l.unfinished = (int64_t)w.id;
        w.launched = true;
        launched_reads += c.n;
        t_launch += now() - t0;
        return PA_OK;
    }
};

constexpr int WIN_OK = 0, WIN_ODD = 1, WIN_EMPTY = 2;

int process_reads_impl(pa_index* const* idxs, int nidx, const char* fastq_path, const char* out_path, int num_threads, uint64_t* n_reads_out, uint64_t* n_flagged_out) {
    if (!idxs || nidx < 1 || !fastq_path || !out_path) return fail(PA_ERR_INVALID_ARG, "null argument");
    for (int i = 0; i < nidx; ++i)
        if (!idxs[i]) return fail(PA_ERR_INVALID_ARG, "null index handle");
    const double t_enter = TextPipe::now();
    if (num_threads < 1) num_threads = 1;
    if (n_reads_out) *n_reads_out = 0;
    if (n_flagged_out) *n_flagged_out = 0;
    {
        pa_index_stats s0, si;
        if (pa_index_get_stats(idxs[0], &s0) != PA_OK) return PA_ERR_INVALID_ARG;
        for (int i = 1; i < nidx; ++i) {
            if (pa_index_get_stats(idxs[i], &si) != PA_OK) return PA_ERR_INVALID_ARG;
            if (si.k != s0.k || si.num_nodes != s0.num_nodes || si.num_classes != s0.num_classes || si.num_kmers != s0.num_kmers)
                return fail(PA_ERR_INVALID_ARG, "handle %d is not a replica of handle 0 (k / nodes / classes / k-mers differ)", i);
        }
    }

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
    uint64_t W = 64ull << 20;   // bytes of a window the GPU scans (64 MiB: 128 MiB leaves more of the first read and the last kernels unoverlapped, 16 MiB costs launches) (PA_INGEST_WINDOW; never more than 256 bytes per read of a batch: the tests' small batches give small windows)
    if (const char* v = getenv("PA_INGEST_WINDOW")) { const long long x = atoll(v); if (x >= 1) W = (uint64_t)x; }
    W = std::min<uint64_t>(std::min<uint64_t>(W, BATCH_READS * 256), 1ull << 31);
    const bool verbose = getenv("PA_VERBOSE") != nullptr;
    const bool lane_serial = knob_int("PA_LANE_SERIAL", 1) != 0;   // (knobs builds: A/B of the one-window-at-a-time rule for lanes that share a GPU)
    const bool host_only = getenv("PA_INGEST_HOST_SCAN") != nullptr;   // (diagnosis: every window through the host's scan)
    const double t_begin = TextPipe::now();
    Pool pool(num_threads);
    int rc = PA_OK;

    // ---- lanes: one per handle; buffers and streams of an earlier call are taken from the handle ----
    std::vector<Lane> lanes((size_t)nidx);
    for (int i = 0; i < nidx && rc == PA_OK; ++i) {
        Lane& l = lanes[(size_t)i];
        l.idx = idxs[i];
        const uint32_t *h_ec = nullptr, *h_class_ref = nullptr;
        index_host_classes(l.idx, &h_ec, &h_class_ref, &l.device);
        if (hipSetDevice(l.device) != hipSuccess) { rc = fail(PA_ERR_HIP, "hipSetDevice(%d) failed", l.device); break; }
        l.cache = static_cast<IngestCache*>(index_take_ingest_cache(l.idx));
        if (!l.cache) l.cache = new IngestCache();
        l.cache->idx = l.idx;
        if (!l.cache->stream && hipStreamCreateWithFlags(&l.cache->stream, hipStreamNonBlocking) != hipSuccess) { l.cache->stream = nullptr; rc = fail(PA_ERR_HIP, "hipStreamCreate failed"); break; }
        if (!l.cache->copy_stream && hipStreamCreateWithFlags(&l.cache->copy_stream, hipStreamNonBlocking) != hipSuccess) { l.cache->copy_stream = nullptr; rc = fail(PA_ERR_HIP, "hipStreamCreate failed"); break; }
        for (hipStream_t* sp : {&l.cache->scan_stream, &l.cache->back_stream})
            if (rc == PA_OK && !*sp && hipStreamCreateWithFlags(sp, hipStreamNonBlocking) != hipSuccess) { *sp = nullptr; rc = fail(PA_ERR_HIP, "hipStreamCreate failed"); }
        if (rc != PA_OK) break;
        l.stream = l.cache->stream;
        l.copy = l.cache->copy_stream;
        l.scan = l.cache->scan_stream;
        l.back = l.cache->back_stream;
    }

    Writer writer(out);
    TextPipe tp{fastq_path, text, pool, writer, lanes, BATCH_READS};
    tp.use_pread = knob_int("PA_INGEST_PREAD", 0) != 0;
    const uint64_t fsize0 = text.fsize;
    const uint64_t KEEP = std::max<uint64_t>(4096, std::min<uint64_t>(W / 4, 1ull << 20));   // the end of the text is the host's: its rules for the last record live there
    uint64_t rec_start = 0;   // text offset of the first record no window has taken yet (known once the window before has been scanned)
    uint64_t read_to = 0;     // text read so far
    bool gpu_mode = rc == PA_OK && !host_only && fsize0 > KEEP;
    bool have_pending = false;
    Win pending;              // the window whose records the GPU is finding
    hipEvent_t vt0[8] = {nullptr}, vt1[8] = {nullptr};   // PA_VERBOSE: how long the windows' copies to the GPU took (eight windows back)
    uint64_t vbytes[8] = {0};
    double v_h2d_ms = 0, v_h2d_bytes = 0;

    // the pending window's scan: waited for; its records are launched, the next window's first record is known
    auto resolve = [&]() -> int {
        Lane& l = lanes[(size_t)pending.lane];
        BatchCtx& c = l.cache->ctx[pending.slot];
        have_pending = false;
        int e = tp.use(l);
        if (e != PA_OK) return e;
        for (int attempt = 0;; ++attempt) {
            const double t0 = TextPipe::now();
            if (hipEventSynchronize(c.ev_info) != hipSuccess) return fail(PA_ERR_HIP, "waiting for the FASTQ scan failed");
            tp.t_wait += TextPipe::now() - t0;
            if (!c.h_info->overflow) break;
            if (attempt == 2) return fail(PA_ERR_INTERNAL, "FASTQ scan: line table too small after regrowing");
            ++tp.rescans;   // more lines than guessed (short reads): grow the line table, fill it again from the counts already there
            if ((e = window_ensure_scan(c, c.h_info->lines)) != PA_OK) return e;
            if ((e = window_scan_enqueue(c, true, l.scan)) != PA_OK) return e;
        }
        if (c.h_info->odd) return WIN_ODD;
        if (c.h_info->n == 0) return WIN_EMPTY;
        if (c.h_info->max_seq > PA_MAX_READ_LEN) return fail(PA_ERR_UNSUPPORTED, "read longer than %u bases", PA_MAX_READ_LEN);
        c.n = c.h_info->n;
        c.wpr = pa_words_per_read(c.h_info->max_seq ? c.h_info->max_seq : 1);
        rec_start = pending.from + c.h_info->consumed;
        tp.wins.push_back(pending);
        ++tp.gpu_windows;
        return tp.launch(tp.wins.back());
    };

    try {
    // ---- windows the GPU scans ----
    // The text of window w + 1 is read (by the pool's workers, asynchronously) while this thread waits for window w - 1's scan, launches its kernels and
    // enqueues window w's scan: the reads follow each other without a gap, and so do the copies to the GPU behind them.
    struct Pre { bool active = false; uint64_t id = 0, main_from = 0, main_len = 0; BatchCtx* c = nullptr; Lane* l = nullptr; };
    auto start_read = [&](Pre& p, uint64_t id) -> int {
        p.active = false;
        if (!gpu_mode || read_to + KEEP >= text.fsize) return PA_OK;
        p.main_len = std::min<uint64_t>(W, text.fsize - KEEP - read_to);
        p.main_from = read_to;
        p.id = id;
        int e = tp.acquire(id, &p.c);
        if (e != PA_OK) return e;
        p.l = &tp.lane_of(id);
        if ((e = window_ensure_raw(*p.c, WINDOW_HEAD_ROOM + p.main_len)) != PA_OK) return e;
        tp.read_begin(read_to, p.main_len, p.c->h_raw + WINDOW_HEAD_ROOM);
        read_to += p.main_len;
        p.active = true;
        return PA_OK;
    };
    Pre cur, nxt;
    if (rc == PA_OK) rc = start_read(cur, tp.next_id);
    while (rc == PA_OK && cur.active) {
        const uint64_t id = cur.id, main_len = cur.main_len, main_from = cur.main_from;
        BatchCtx& c = *cur.c;
        Lane& l = *cur.l;
        double t0 = TextPipe::now();
        if ((rc = tp.read_end()) != PA_OK) break;                       // this window's text is in pinned memory
        tp.t_read += TextPipe::now() - t0;
        if ((rc = tp.use(l)) != PA_OK) break;
        if (verbose) {
            if (!vt0[id % 8]) { (void)hipEventCreate(&vt0[id % 8]); (void)hipEventCreate(&vt1[id % 8]); }
            else { float ms = 0; if (hipEventElapsedTime(&ms, vt0[id % 8], vt1[id % 8]) == hipSuccess) { v_h2d_ms += ms; v_h2d_bytes += vbytes[id % 8]; } }
            (void)hipEventRecord(vt0[id % 8], l.copy);
            vbytes[id % 8] = main_len;
        }
        // lanes that share a GPU (a handle listed twice) send their windows one at a time: with two copies of one direction queued at once the runtime
        // runs one of them as a blit kernel, at a fraction of the DMA engine's rate (host_batch.cpp has the measurement)
        for (size_t o = 0; o < lanes.size() && lane_serial; ++o)
            if (&lanes[o] != &l && lanes[o].device == l.device && lanes[o].last_h2d && hipStreamWaitEvent(l.copy, lanes[o].last_h2d, 0) != hipSuccess) { rc = fail(PA_ERR_HIP, "hipStreamWaitEvent failed"); break; }
        if (rc != PA_OK) break;
        if (hipMemcpyAsync((uint8_t*)c.d_raw + WINDOW_HEAD_ROOM, c.h_raw + WINDOW_HEAD_ROOM, main_len, hipMemcpyHostToDevice, l.copy) != hipSuccess ||
            hipEventRecord(c.ev_h2d, l.copy) != hipSuccess) { rc = fail(PA_ERR_HIP, "copy of a text window to the GPU failed"); break; }
        l.last_h2d = c.ev_h2d;
        if (verbose) (void)hipEventRecord(vt1[id % 8], l.copy);
        if ((rc = start_read(nxt, id + 1)) != PA_OK) break;             // the next window's text starts to arrive
        bool discard = false;
        if (have_pending) {
            const int r = resolve();
            if (r == WIN_ODD) { gpu_mode = false; discard = true; rec_start = pending.from; tp.next_id = pending.id; }               // not four-line text from here on: the host's scan takes over
            else if (r == WIN_EMPTY) { W = std::max<uint64_t>(2 * W, 2 * (main_from - pending.from)); discard = true; rec_start = pending.from; tp.next_id = pending.id; }   // no whole record in the window: a longer one
            else if (r != PA_OK) { rc = r; break; }
        }
        if ((rc = tp.use(l)) != PA_OK) break;
        const uint64_t head = main_from - rec_start;   // the unfinished record of the window before
        if (!discard && head > WINDOW_HEAD_ROOM) { W = std::max<uint64_t>(W, 2 * head); discard = true; }
        if (discard) {   // this window's text (and what was being read behind it) is read again, from the first record not yet taken
            if (nxt.active) { (void)tp.read_end(); nxt.active = false; }
            (void)hipStreamSynchronize(l.copy);
            read_to = rec_start;
            if (W > (1ull << 31)) gpu_mode = false;   // (a record of gigabytes: the host's scan says what it is)
            if ((rc = start_read(cur, tp.next_id)) != PA_OK) break;
            continue;
        }
        t0 = TextPipe::now();
        if (head) {
            if ((rc = tp.read_small(rec_start, head, c.h_raw + WINDOW_HEAD_ROOM - head)) != PA_OK) break;
            if (hipMemcpyAsync((uint8_t*)c.d_raw + WINDOW_HEAD_ROOM - head, c.h_raw + WINDOW_HEAD_ROOM - head, head, hipMemcpyHostToDevice, l.scan) != hipSuccess) { rc = fail(PA_ERR_HIP, "copy of a window's head failed"); break; }
        }
        tp.t_read += TextPipe::now() - t0; t0 = TextPipe::now();
        c.raw_begin = WINDOW_HEAD_ROOM - head;
        c.raw_end = WINDOW_HEAD_ROOM + main_len;
        if ((rc = window_ensure_scan(c, 0)) != PA_OK) break;
        if (hipStreamWaitEvent(l.scan, c.ev_h2d, 0) != hipSuccess) { rc = fail(PA_ERR_HIP, "hipStreamWaitEvent failed"); break; }
        if ((rc = window_scan_enqueue(c, false, l.scan)) != PA_OK) break;
        tp.t_launch += TextPipe::now() - t0;
        pending = Win();
        pending.id = id;
        pending.lane = (int)(id % (uint64_t)nidx);
        pending.slot = tp.slot_of(id);
        pending.from = rec_start;
        have_pending = true;
        tp.next_id = id + 1;
        if ((rc = tp.retire_finished(0)) != PA_OK) break;
        cur = nxt;
        nxt.active = false;
    }
    if (cur.active || nxt.active) (void)tp.read_end();   // (an error path: nothing of the pool's job is left behind)
    gpu_mode = false;
    if (rc == PA_OK && have_pending) {
        const int r = resolve();
        if (r == WIN_ODD || r == WIN_EMPTY) { rec_start = pending.from; tp.next_id = pending.id; }
        else if (r != PA_OK) rc = r;
    }

    // ---- the rest of the text (its end; all of it when it is not in four-line shape): the host's scan, the same kernels ----
    if (rc == PA_OK) {
        IngestCache* const hc = lanes[0].cache;   // (the scan's lists are parked with lane 0's buffers)
        text.off = rec_start;
        WindowScan ws(text);
        uint64_t records_before = tp.launched_reads;
        for (;;) {
            double t0 = TextPipe::now();
            rc = ws.next(fastq_path, records_before, pool, hc->rec_pos, hc->brk);
            tp.t_scan += TextPipe::now() - t0;
            if (rc != PA_OK || ws.nrec == 0) break;
            records_before += ws.nrec;
            const RecPos* const rp = hc->rec_pos.data();
            for (uint64_t i0 = 0; i0 < ws.nrec && rc == PA_OK;) {
                // a batch of whole records whose text fits a window of 2 GiB (offsets into it are 32 bits)
                uint64_t i1 = std::min<uint64_t>(ws.nrec, i0 + BATCH_READS);
                const uint64_t first = rp[i0].start;
                auto end_of = [&](uint64_t i) { return i < ws.nrec ? rp[i].start : ws.size; };
                while (i1 > i0 + 1 && end_of(i1) - first > (1ull << 31)) i1 = i0 + (i1 - i0) / 2;
                const uint64_t bytes = end_of(i1) - first, n = i1 - i0;
                if (bytes > (3ull << 30)) { rc = fail(PA_ERR_UNSUPPORTED, "%s: record %llu is longer than 3 GiB", fastq_path, (unsigned long long)(tp.launched_reads)); break; }
                const uint64_t id = tp.next_id;
                BatchCtx* cp = nullptr;
                if ((rc = tp.acquire(id, &cp)) != PA_OK) break;
                BatchCtx& c = *cp;
                Lane& l = tp.lane_of(id);
                if ((rc = window_ensure_raw(c, WINDOW_HEAD_ROOM + bytes)) != PA_OK) break;
                if ((rc = window_ensure_recs(c, n, true)) != PA_OK) break;
                t0 = TextPipe::now();
                {   // the batch's text and where its records lie in it
                    const char* const src = ws.base + first;
                    const uint64_t PIECE = 2ull << 20;
                    const int ntask = (int)std::max<uint64_t>(1, std::min<uint64_t>((bytes + PIECE - 1) / PIECE, 1u << 20));
                    pool.run(ntask, [&](int t) {
                        const uint64_t a = bytes * (uint64_t)t / (uint64_t)ntask, b = bytes * (uint64_t)(t + 1) / (uint64_t)ntask;
                        memcpy(c.h_raw + WINDOW_HEAD_ROOM + a, src + a, (size_t)(b - a));
                    });
                }
                const int T4 = pool.size() * 4;
                std::vector<uint32_t> tmax((size_t)T4, 0);
                pool.run(T4, [&](int t) {
                    uint32_t mx = 0;
                    for (uint64_t i = n * (uint64_t)t / (uint64_t)T4; i < n * (uint64_t)(t + 1) / (uint64_t)T4; ++i) {
                        const RecPos& r = rp[i0 + i];
                        const uint64_t seq_off = std::min<uint64_t>(r.start + r.hdr + 1, ws.size);
                        const uint32_t seq_len = (uint32_t)std::min<uint64_t>(r.seq_len, ws.size - seq_off);
                        c.h_rec[i] = make_uint4((uint32_t)(WINDOW_HEAD_ROOM + r.start + 1 - first), r.id_len, (uint32_t)(WINDOW_HEAD_ROOM + seq_off - first), seq_len);
                        mx = std::max(mx, seq_len);
                    }
                    tmax[(size_t)t] = mx;
                });
                uint32_t maxlen = 1;
                for (uint32_t m : tmax) maxlen = std::max(maxlen, m);
                tp.t_read += TextPipe::now() - t0;
                if (maxlen > PA_MAX_READ_LEN) { rc = fail(PA_ERR_UNSUPPORTED, "read longer than %u bases", PA_MAX_READ_LEN); break; }
                // (the copies ride on the lane's kernel stream: this path is bound by the host's scan, not by the link)
                if (hipMemcpyAsync((uint8_t*)c.d_raw + WINDOW_HEAD_ROOM, c.h_raw + WINDOW_HEAD_ROOM, bytes, hipMemcpyHostToDevice, l.stream) != hipSuccess ||
                    hipMemcpyAsync(c.d_rec, c.h_rec, n * sizeof(uint4), hipMemcpyHostToDevice, l.stream) != hipSuccess) { rc = fail(PA_ERR_HIP, "copy of a text window to the GPU failed"); break; }
                c.raw_begin = WINDOW_HEAD_ROOM;
                c.raw_end = WINDOW_HEAD_ROOM + bytes;
                c.n = n;
                c.wpr = pa_words_per_read(maxlen);
                Win w;
                w.id = id;
                w.lane = (int)(id % (uint64_t)nidx);
                w.slot = tp.slot_of(id);
                w.from = 0;
                tp.next_id = id + 1;
                tp.wins.push_back(w);
                ++tp.host_windows;
                rc = tp.launch(tp.wins.back());
                i0 = i1;
            }
            if (rc != PA_OK) break;
        }
    }
    if (rc == PA_OK) rc = tp.retire_finished(~0ull);
    } catch (const std::bad_alloc&) {
        rc = fail(PA_ERR_OOM, "out of host memory in pa_process_reads");
    } catch (const std::exception& ex) {
        rc = fail(PA_ERR_INTERNAL, "pa_process_reads: %s", ex.what());
    }
    {
        double* st = pa::ingest::last_stage_seconds();
        st[0] = tp.t_scan; st[1] = tp.t_read; st[2] = tp.t_wait; st[3] = tp.t_launch; st[4] = tp.t_text; st[5] = tp.t_push; st[6] = TextPipe::now() - t_begin; st[7] = (double)tp.reported;
    }
    if (verbose)
        fprintf(stderr, "\n[pa ingest] %llu reads, %d threads, %d lane(s): %llu windows scanned on the GPU (%llu scanned twice), %llu batches by the host; host scan %.3f s, read %.3f s, wait GPU %.3f s, launch %.3f s, wait text %.3f s, wait writer %.3f s, total %.3f s (before the first window %.3f s)\n",
                (unsigned long long)tp.reported, pool.size(), nidx, (unsigned long long)tp.gpu_windows, (unsigned long long)tp.rescans, (unsigned long long)tp.host_windows, tp.t_scan, tp.t_read, tp.t_wait,
                tp.t_launch, tp.t_text, tp.t_push, TextPipe::now() - t_begin, t_begin - t_enter);
    if (verbose && v_h2d_ms > 0) fprintf(stderr, "[pa ingest] windows to the GPU: %.1f MB in %.2f ms of copies = %.1f GB/s\n", v_h2d_bytes / 1e6, v_h2d_ms, v_h2d_bytes / v_h2d_ms / 1e6);
    for (int i = 0; i < 8; ++i) { if (vt0[i]) (void)hipEventDestroy(vt0[i]); if (vt1[i]) (void)hipEventDestroy(vt1[i]); }
    if (tp.reported >= 1000000) fputc('\n', stderr);   // (`eprintln!()` behind the progress line, :508)
    for (Lane& l : lanes) {
        if (!l.cache) continue;
        (void)hipSetDevice(l.device);
        for (hipStream_t s : {l.copy, l.scan, l.back})
            if (s) (void)hipStreamSynchronize(s);
        if (l.stream) (void)hipStreamSynchronize(l.stream);   // (the streams stay with the parked buffers; IngestCache::destroy releases them)
    }
    bool wrote = true;
    try { wrote = writer.finish(); } catch (...) { wrote = false; }
    if (rc == PA_OK && !wrote) rc = fail(PA_ERR_IO, "short write to %s", out_path);
    const std::string why = rc != PA_OK ? last_error_ref() : std::string();
    for (Lane& l : lanes) {
        if (!l.cache) continue;
        (void)hipSetDevice(l.device);
        if (l.cache->rec_pos.capacity() > ((size_t)64 << 20)) { std::vector<RecPos>().swap(l.cache->rec_pos); std::vector<std::vector<uint32_t>>().swap(l.cache->brk); }   // (do not park more than 1 GB of it)
        if (rc == PA_OK) index_put_ingest_cache(l.idx, l.cache, IngestCache::destroy);   // the next call starts with warm buffers
        else IngestCache::destroy(l.cache);
        l.cache = nullptr;
    }
    text.release();
    if (out != stdout) { if (fclose(out) != 0 && rc == PA_OK) rc = fail(PA_ERR_IO, "close %s: %s", out_path, strerror(errno)); }
    else fflush(stdout);
    if (rc != PA_OK && !why.empty()) last_error_ref() = why;
    if (n_reads_out) *n_reads_out = tp.reported;
    if (n_flagged_out) *n_flagged_out = tp.flagged;
    return rc;
}

}  // namespace

extern "C" int pa_process_reads(pa_index* idx, const char* fastq_path, const char* out_path, int num_threads, uint64_t* n_reads_out,
                                uint64_t* n_flagged_out) {
    pa_index* one[1] = {idx};
    try {
        return process_reads_impl(one, 1, fastq_path, out_path, num_threads, n_reads_out, n_flagged_out);
    } catch (const std::bad_alloc&) {
        return fail(PA_ERR_OOM, "out of host memory in pa_process_reads");
    } catch (const std::exception& ex) {   // (thread creation: std::system_error) — nothing crosses the C ABI
        return fail(PA_ERR_INTERNAL, "pa_process_reads: %s", ex.what());
    }
}

extern "C" int pa_process_reads_multi(pa_index* const* idx, int n_idx, const char* fastq_path, const char* out_path, int num_threads, uint64_t* n_reads_out,
                                      uint64_t* n_flagged_out) {
    try {
        return process_reads_impl(idx, n_idx, fastq_path, out_path, num_threads, n_reads_out, n_flagged_out);
    } catch (const std::bad_alloc&) {
        return fail(PA_ERR_OOM, "out of host memory in pa_process_reads_multi");
    } catch (const std::exception& ex) {
        return fail(PA_ERR_INTERNAL, "pa_process_reads_multi: %s", ex.what());
    }
}

extern "C" int pa_process_reads_stage_seconds(double out[PA_INGEST_STAGES]) {
    if (!out) return fail(PA_ERR_INVALID_ARG, "null argument");
    memcpy(out, pa::ingest::last_stage_seconds(), sizeof(double) * PA_INGEST_STAGES);
    return PA_OK;
}
