// process_reads (src/pseudoaligner.rs:420-514) for a caller that HOLDS the reader: the reference's signature consumes an open
// fastq::Reader (:421), so a drop-in replacement cannot ask for a path. The caller pushes the records it reads — ids as
// record.id() gives them (:456), sequences as record.seq() (:449) — and pulls the reference's Debug tuples (:490) in push order.
// Behind the two calls runs the batch pipeline of fastq.cpp with its stages overlapped (ingest.hpp):
//
//   push     records are copied into the batch being filled; a full batch is 2-bit packed into pinned tiles by the worker
//            pool and launched (H2D -> pa_map_batch_device -> D2H on the stream's own HIP stream), then the PREVIOUS batch —
//            whose GPU leg ran while this one was being filled — is rendered into text by the pool
//   pull     copies rendered text out, whole lines, never waits for the GPU
//   flush    launches what is left, waits, renders: afterwards pull drains everything pushed so far
//
// The GPU leg of batch b therefore overlaps the caller's reading + the packing of batch b + 1 and the rendering of batch b - 1,
// exactly as in pa_process_reads; output order is input order (the reference's is completion order, :490).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <new>
#include <stdexcept>

#include "ingest.hpp"
#include "pa_common.hpp"

using namespace pa;
using namespace pa::ingest;

// bytes of a batch's records: grows without initialising what the copy is about to overwrite (a std::vector would zero it first)
struct RawText {
    char* p = nullptr;
    size_t n = 0, cap = 0;
    ~RawText() { free(p); }
    bool grow(size_t extra) {
        if (n + extra <= cap) return true;
        size_t want = cap ? cap : (size_t)1 << 20;
        while (want < n + extra) want *= 2;
        char* q = (char*)realloc(p, want);
        if (!q) return false;
        p = q;
        cap = want;
        return true;
    }
    const char* data() const { return p; }
    void clear() { n = 0; }
};

struct RsLane {   // one index handle (one GPU) of a record stream: its parked buffers (two batches) and its HIP stream
    pa_index* idx = nullptr;
    int device = 0;
    IngestCache* cache = nullptr;   // taken from the handle (warm, if a pa_process_reads call or another record stream left it there) and parked there again at the end
    hipStream_t stream = nullptr;
};

struct pa_record_stream {
    std::vector<RsLane> lanes;     // pa_record_stream_create_multi: batches go round-robin to the lanes, their tuples come back in push order
    std::unique_ptr<Pool> pool;
    uint64_t batch_reads = 2u << 20;
    // slot = 2 * lane + half: the batch being filled, and per lane the batch in flight before it
    std::vector<RawText> text;     // ids and sequences of the slot's records (Record offsets point into it)
    std::vector<uint32_t> maxlen;
    std::vector<char> inflight;
    uint64_t seq = 0;              // batches submitted so far: batch b uses lane b % L, half (b / L) % 2
    std::deque<TextBuf> outq;      // rendered text in order; out_off = bytes of the front buffer already pulled
    std::vector<TextBuf> spare;    // buffers the caller has pulled empty: the next batches' text goes into them (fresh memory is paged in again every time)
    size_t out_off = 0;
    uint64_t n_reads = 0, n_flagged = 0;
    double stage[PA_INGEST_STAGES] = {0, 0, 0, 0, 0, 0, 0, 0};   // pa_record_stream_stage_seconds
    int rc = PA_OK;                // sticky: after a failure every call reports it
    std::string why;
    int L() const { return (int)lanes.size(); }
    int slot_of(uint64_t b) const { return 2 * (int)(b % (uint64_t)L()) + (int)((b / (uint64_t)L()) & 1u); }
    RsLane& lane_of_slot(int k) { return lanes[(size_t)(k >> 1)]; }
    BatchCtx& ctx(int k) { return lanes[(size_t)(k >> 1)].cache->ctx[k & 1]; }
    int cur() const { return slot_of(seq); }
};

namespace {

int fail_sticky(pa_record_stream* s, int rc) {
    if (s->rc == PA_OK) { s->rc = rc; s->why = last_error_ref(); }
    return rc;
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int render(pa_record_stream* s, int k) {
    // the batch's tuples were rendered on the GPU (batch_finish -> render.hip): wait for their copy in pinned memory and move it into the
    // output queue as ONE buffer (it ends with a line break; pull hands out whole lines), the pool sharing the copy
    const double t_render = now_s();
    BatchCtx& c = s->ctx(k);
    if (hipSetDevice(s->lane_of_slot(k).device) != hipSuccess) return fail(PA_ERR_HIP, "hipSetDevice failed");
    const int rc = batch_text_wait(c);
    if (rc != PA_OK) return rc;
    if (c.text_bytes) {
        TextBuf buf;
        if (!s->spare.empty()) { buf = std::move(s->spare.back()); s->spare.pop_back(); buf.len = 0; }
        char* dst = buf.room(c.text_bytes);
        const int P = s->pool->size() * 2;
        const size_t total = c.text_bytes;
        s->pool->run(P, [&](int t) {
            const size_t a = total * (size_t)t / P, b = total * (size_t)(t + 1) / P;
            memcpy(dst + a, c.h_text + a, b - a);
        });
        buf.len = total;
        s->outq.push_back(std::move(buf));
    }
    s->n_flagged += c.flagged;
    s->n_reads += c.n;
    s->inflight[(size_t)k] = 0;
    c.recs.clear();
    c.n = 0;
    s->text[(size_t)k].clear();
    s->maxlen[(size_t)k] = 0;
    s->stage[4] += now_s() - t_render;
    s->stage[7] = (double)s->n_reads;
    return PA_OK;
}

// The batch being filled goes to its lane's GPU; the batch launched on that lane before it is waited for and rendered. (The two batches of a lane share
// its stream and with it ONE launch context inside the index — its control block says how much of the arena a launch used — so the previous batch is
// finished before the next is launched; the GPU then works on batch b while the host renders batch b - L and packs b + 1.) With L lanes the batch waited
// for is b - L: the oldest one not yet rendered, so the tuples come out in push order.
int submit(pa_record_stream* s) {
    const int k = s->cur(), o = k ^ 1;
    RsLane& lane = s->lane_of_slot(k);
    BatchCtx& c = s->ctx(k);
    c.n = c.recs.size();
    if (c.n == 0) return PA_OK;
    if (s->maxlen[(size_t)k] > PA_MAX_READ_LEN) return fail(PA_ERR_UNSUPPORTED, "read longer than %u bases", PA_MAX_READ_LEN);
    if (hipSetDevice(lane.device) != hipSuccess) return fail(PA_ERR_HIP, "hipSetDevice(%d) failed", lane.device);
    c.wpr = pa_words_per_read(s->maxlen[(size_t)k] ? s->maxlen[(size_t)k] : 1);
    double t0 = now_s();
    std::vector<uint64_t> part;
    batch_offsets(*s->pool, c, part);
    int rc = batch_ensure(lane.idx, c, c.n, c.wpr, s->batch_reads);
    if (rc != PA_OK) return rc;
    batch_gather_ascii(*s->pool, c, s->text[(size_t)k].data(), part);
    s->stage[1] += now_s() - t0; t0 = now_s();
    if (s->inflight[(size_t)o] && (rc = batch_finish(lane.idx, s->ctx(o), lane.stream)) != PA_OK) return rc;
    s->stage[2] += now_s() - t0; t0 = now_s();
    if ((rc = batch_launch(lane.idx, c, lane.stream)) != PA_OK) return rc;
    s->stage[3] += now_s() - t0;
    s->inflight[(size_t)k] = 1;
    if (s->inflight[(size_t)o] && (rc = render(s, o)) != PA_OK) return rc;
    ++s->seq;
    return PA_OK;
}

}  // namespace

extern "C" {

int pa_record_stream_create_multi(pa_index* const* idxs, int n_idx, int num_threads, uint64_t batch_reads, pa_record_stream** out) {
    if (!idxs || n_idx < 1 || !out) return fail(PA_ERR_INVALID_ARG, "null argument");
    for (int i = 0; i < n_idx; ++i)
        if (!idxs[i]) return fail(PA_ERR_INVALID_ARG, "null index handle");
    {
        pa_index_stats s0, si;
        if (pa_index_get_stats(idxs[0], &s0) != PA_OK) return PA_ERR_INVALID_ARG;
        for (int i = 1; i < n_idx; ++i) {
            if (pa_index_get_stats(idxs[i], &si) != PA_OK) return PA_ERR_INVALID_ARG;
            if (si.k != s0.k || si.num_nodes != s0.num_nodes || si.num_classes != s0.num_classes || si.num_kmers != s0.num_kmers)
                return fail(PA_ERR_INVALID_ARG, "handle %d is not a replica of handle 0 (k / nodes / classes / k-mers differ)", i);
        }
    }
    pa_record_stream* s = new (std::nothrow) pa_record_stream();
    if (!s) return fail(PA_ERR_OOM, "out of memory");
    if (batch_reads) s->batch_reads = std::max<uint64_t>(64, batch_reads / 64 * 64);
    int T = num_threads > 0 ? num_threads : usable_threads();
    if (T < 1) T = 1;
    try {
        s->pool.reset(new Pool(T));
        s->lanes.resize((size_t)n_idx);
        s->text.resize(2 * (size_t)n_idx);
        s->maxlen.assign(2 * (size_t)n_idx, 0);
        s->inflight.assign(2 * (size_t)n_idx, 0);
    } catch (const std::exception& ex) {
        delete s;
        return fail(PA_ERR_OOM, "record stream: %s", ex.what());
    }
    for (int i = 0; i < n_idx; ++i) {
        RsLane& l = s->lanes[(size_t)i];
        l.idx = idxs[i];
        const uint32_t *h_ec = nullptr, *h_ref = nullptr;
        index_host_classes(l.idx, &h_ec, &h_ref, &l.device);
        l.cache = static_cast<IngestCache*>(index_take_ingest_cache(l.idx));   // buffers of an earlier call, if any: allocating them costs more than packing a batch
        if (!l.cache) l.cache = new (std::nothrow) IngestCache();
        if (!l.cache) { pa_record_stream_destroy(s); return fail(PA_ERR_OOM, "out of memory"); }
        l.cache->idx = l.idx;
        for (int k = 0; k < 2; ++k) { BatchCtx& c = l.cache->ctx[k]; c.recs.clear(); c.n = 0; c.first = 0; c.in_place = false; c.flag_mark = 0; c.back = nullptr; c.text_on_back = false; }   // (a parked set still names its last batch — or a window of pa_process_reads)
        if (hipSetDevice(l.device) != hipSuccess) {   // (a stream the parked cache brought along stays with it: destroy releases it and its launch context)
            pa_record_stream_destroy(s);
            return fail(PA_ERR_HIP, "hipSetDevice(%d) failed", l.device);
        }
        if (!l.cache->stream && hipStreamCreateWithFlags(&l.cache->stream, hipStreamNonBlocking) != hipSuccess) {
            l.cache->stream = nullptr;
            pa_record_stream_destroy(s);
            return fail(PA_ERR_HIP, "hipStreamCreate failed");
        }
        l.stream = l.cache->stream;
    }
    *out = s;
    return PA_OK;
}

int pa_record_stream_create(pa_index* idx, int num_threads, uint64_t batch_reads, pa_record_stream** out) {
    pa_index* one[1] = {idx};
    return pa_record_stream_create_multi(one, 1, num_threads, batch_reads, out);
}

void pa_record_stream_destroy(pa_record_stream* s) {
    if (!s) return;
    for (RsLane& l : s->lanes) {
        if (!l.cache) continue;
        (void)hipSetDevice(l.device);
        if (l.stream) (void)hipStreamSynchronize(l.stream);
        for (int k = 0; k < 2; ++k) { l.cache->ctx[k].recs.clear(); l.cache->ctx[k].n = 0; }
        if (s->rc == PA_OK && l.stream) index_put_ingest_cache(l.idx, l.cache, IngestCache::destroy);   // the next stream / pa_process_reads call starts warm
        else IngestCache::destroy(l.cache);   // (releases the stream's launch context inside the index and the stream)
        l.cache = nullptr;
    }
    delete s;
}

// (std::bad_alloc out of the growable buffers, std::system_error out of the worker pool: nothing crosses the C ABI)
#define PA_RS_GUARD(call) \
    try { return call; } \
    catch (const std::bad_alloc&) { return s ? fail_sticky(s, fail(PA_ERR_OOM, "out of host memory in the record stream")) : PA_ERR_OOM; } \
    catch (const std::exception& ex) { return s ? fail_sticky(s, fail(PA_ERR_INTERNAL, "record stream: %s", ex.what())) : PA_ERR_INTERNAL; }
static int records_push_impl(pa_record_stream* s, const uint8_t* ids, const uint64_t* id_offsets, const uint8_t* seqs, const uint64_t* seq_offsets, uint64_t n);
static int records_flush_impl(pa_record_stream* s);
int pa_records_push(pa_record_stream* s, const uint8_t* ids, const uint64_t* id_offsets, const uint8_t* seqs, const uint64_t* seq_offsets, uint64_t n) {
    PA_RS_GUARD(records_push_impl(s, ids, id_offsets, seqs, seq_offsets, n))
}
int pa_records_flush(pa_record_stream* s) {
    PA_RS_GUARD(records_flush_impl(s))
}
static int records_push_impl(pa_record_stream* s, const uint8_t* ids, const uint64_t* id_offsets, const uint8_t* seqs, const uint64_t* seq_offsets, uint64_t n) {
    if (!s || (n && (!id_offsets || !seq_offsets))) return fail(PA_ERR_INVALID_ARG, "null argument");
    if (s->rc != PA_OK) return fail(s->rc, "%s", s->why.c_str());
    // every record of the call is checked before the first one is taken: a call that fails has appended (and submitted) nothing,
    // so the caller may correct it and push the same records again
    for (uint64_t i = 0; i < n; ++i) {
        if (id_offsets[i + 1] < id_offsets[i] || seq_offsets[i + 1] < seq_offsets[i]) return fail(PA_ERR_INVALID_ARG, "offsets not monotone at record %llu", (unsigned long long)i);
        const uint64_t il = id_offsets[i + 1] - id_offsets[i], sl = seq_offsets[i + 1] - seq_offsets[i];
        if (il > 0xFFFFFFFFull || sl > 0xFFFFFFFFull || (il && !ids) || (sl && !seqs)) return fail(PA_ERR_INVALID_ARG, "record %llu: bad id or sequence", (unsigned long long)i);
    }
    // the records go into the batch being filled as two blocks — all their ids, then all their sequences — copied and indexed
    // by the worker pool (the caller's buffers are laid out like that already: a per-record copy cost 70 ns per record)
    for (uint64_t i = 0; i < n;) {
        const int k = s->cur();
        BatchCtx& c = s->ctx(k);
        const uint64_t have = c.recs.size(), m = std::min<uint64_t>(n - i, s->batch_reads - have);
        const uint64_t ib = id_offsets[i + m] - id_offsets[i], sb = seq_offsets[i + m] - seq_offsets[i];
        RawText& t = s->text[(size_t)k];
        if (!t.grow(ib + sb)) return fail_sticky(s, fail(PA_ERR_OOM, "out of memory for %llu bytes of records", (unsigned long long)(ib + sb)));
        const uint64_t id_base = t.n, seq_base = t.n + ib;
        t.n += ib + sb;
        c.recs.resize(have + m);
        const int P = s->pool->size();
        std::vector<uint32_t> tmax((size_t)P, 0);
        s->pool->run(P, [&](int w) {
            const uint64_t a = m * (uint64_t)w / P, b = m * (uint64_t)(w + 1) / P;
            if (a == b) return;
            if (ids) memcpy(t.p + id_base + (id_offsets[i + a] - id_offsets[i]), ids + id_offsets[i + a], id_offsets[i + b] - id_offsets[i + a]);
            if (seqs) memcpy(t.p + seq_base + (seq_offsets[i + a] - seq_offsets[i]), seqs + seq_offsets[i + a], seq_offsets[i + b] - seq_offsets[i + a]);
            uint32_t mx = 0;
            for (uint64_t j = a; j < b; ++j) {
                Record& r = c.recs[have + j];
                r.id_off = id_base + (id_offsets[i + j] - id_offsets[i]);
                r.id_len = (uint32_t)(id_offsets[i + j + 1] - id_offsets[i + j]);
                r.seq_off = seq_base + (seq_offsets[i + j] - seq_offsets[i]);
                r.seq_len = (uint32_t)(seq_offsets[i + j + 1] - seq_offsets[i + j]);
                mx = std::max(mx, r.seq_len);
            }
            tmax[(size_t)w] = mx;
        });
        for (uint32_t v : tmax) s->maxlen[(size_t)k] = std::max(s->maxlen[(size_t)k], v);
        i += m;
        if (c.recs.size() >= s->batch_reads) {
            const int rc = submit(s);
            if (rc != PA_OK) return fail_sticky(s, rc);
        }
    }
    return PA_OK;
}

static int records_flush_impl(pa_record_stream* s) {
    if (!s) return fail(PA_ERR_INVALID_ARG, "null argument");
    if (s->rc != PA_OK) return fail(s->rc, "%s", s->why.c_str());
    int rc = submit(s);   // what is left of the batch being filled (renders the batch launched on its lane before it)
    if (rc != PA_OK) return fail_sticky(s, rc);
    // the batches still in flight, oldest first: the last L submitted (one per lane)
    const uint64_t L = (uint64_t)s->L();
    for (uint64_t b = s->seq >= L ? s->seq - L : 0; b < s->seq; ++b) {
        const int k = s->slot_of(b);
        if (!s->inflight[(size_t)k]) continue;
        RsLane& lane = s->lane_of_slot(k);
        if (hipSetDevice(lane.device) != hipSuccess) return fail_sticky(s, fail(PA_ERR_HIP, "hipSetDevice failed"));
        const double t0 = now_s();
        if ((rc = batch_finish(lane.idx, s->ctx(k), lane.stream)) != PA_OK) return fail_sticky(s, rc);
        s->stage[2] += now_s() - t0;
        if ((rc = render(s, k)) != PA_OK) return fail_sticky(s, rc);
    }
    return PA_OK;
}

int pa_records_pull(pa_record_stream* s, char* buf, size_t cap, size_t* n_bytes) {
    if (!s || !n_bytes || (cap && !buf)) return fail(PA_ERR_INVALID_ARG, "null argument");
    *n_bytes = 0;
    if (s->rc != PA_OK) return fail(s->rc, "%s", s->why.c_str());
    size_t got = 0;
    while (!s->outq.empty() && got < cap) {
        TextBuf& f = s->outq.front();
        const size_t left = f.len - s->out_off, room = cap - got;
        size_t take = left;
        if (take > room) {   // whole lines only
            const void* nl = memrchr(f.mem.data() + s->out_off, '\n', room);
            take = nl ? (size_t)((const char*)nl - (f.mem.data() + s->out_off)) + 1 : 0;
            if (take == 0) {
                if (got == 0) {   // not even the next tuple fits: say how much it needs (the caller grows its buffer and pulls again)
                    const void* end = memchr(f.mem.data() + s->out_off, '\n', left);
                    if (n_bytes) *n_bytes = end ? (size_t)((const char*)end - (f.mem.data() + s->out_off)) + 1 : left;
                    return fail(PA_ERR_BUFFER_TOO_SMALL, "buffer of %zu bytes is smaller than one tuple", cap);
                }
                break;
            }
        }
        if (take >= ((size_t)4 << 20)) {   // a big piece: copied by the pool (it is idle while the caller pulls; one thread copies 8 GB/s, the tuples of 4 M reads are 0.23 GB)
            const int P = s->pool->size();
            const char* src = f.mem.data() + s->out_off;
            char* dst = buf + got;
            s->pool->run(P, [&](int w) {
                const size_t a = take * (size_t)w / P, b = take * (size_t)(w + 1) / P;
                if (b > a) memcpy(dst + a, src + a, b - a);
            });
        } else memcpy(buf + got, f.mem.data() + s->out_off, take);
        got += take;
        s->out_off += take;
        if (s->out_off == f.len) {
            if (s->spare.size() < 3) s->spare.push_back(std::move(f));
            s->outq.pop_front();
            s->out_off = 0;
        }
        else break;   // the buffer is full up to a line boundary
    }
    *n_bytes = got;
    return PA_OK;
}

int pa_record_stream_stage_seconds(const pa_record_stream* s, double out[PA_INGEST_STAGES]) {
    if (!s || !out) return fail(PA_ERR_INVALID_ARG, "null argument");
    memcpy(out, s->stage, sizeof s->stage);
    out[6] = out[1] + out[2] + out[3] + out[4];
    return PA_OK;
}

int pa_record_stream_stats(const pa_record_stream* s, uint64_t* n_reads, uint64_t* n_flagged) {
    if (!s) return fail(PA_ERR_INVALID_ARG, "null argument");
    if (n_reads) *n_reads = s->n_reads;
    if (n_flagged) *n_flagged = s->n_flagged;
    return PA_OK;
}

}  // extern "C"
