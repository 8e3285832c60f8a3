// The reference's output record on the GPU: `println!("{:?}", (flag, record.id(), eq_class, coverage))`
// (src/pseudoaligner.rs:455-461, :490) for every read of a finished batch — "(false, "id", [1, 5, 9], 150)\n" — rendered where
// the records, the class table and the novel class ids already are, so that the host neither walks class tables nor converts
// integers: it receives text.
//
//   pa_render_len_kernel    bytes of every read's tuple (one thread per read) + the number of flagged reads (:455)
//   rocPRIM exclusive scan  where every tuple starts; the last entry is the text's length
//   pa_render_write_kernel  the bytes
//
// A class that IS an index class (pa_read_result.class_off & PA_CLASS_REF) is copied from the index's table of rendered classes
// (device copy of index_host_class_text: every class as "1, 5, 9", built once per index); any other class is rendered from its
// ids in the arena. Ids follow Rust's `impl Debug for str` for ASCII: \t \r \n \\ \" \0 escaped, other control bytes as \u{hex},
// everything from 0x20 on (but 0x7f) copied — bytes from 0x80 on too: the few non-printable code points beyond ASCII (U+0085, U+00A0 ...),
// which Rust prints as \u{..}, are copied as they are (documented limitation).
#include <hip/hip_runtime.h>
#include <rocprim/device/device_scan.hpp>

#include "kernels.hpp"
#include "pa_common.hpp"

namespace pa {
namespace {

__device__ __forceinline__ uint32_t dec_digits(uint32_t v) {
    return 1u + (v >= 10u) + (v >= 100u) + (v >= 1000u) + (v >= 10000u) + (v >= 100000u) + (v >= 1000000u) + (v >= 10000000u) + (v >= 100000000u) +
           (v >= 1000000000u);
}
__device__ __forceinline__ uint8_t* put_dec(uint8_t* o, uint32_t v) {
    uint8_t b[10];
    int k = 10;
    do { b[--k] = (uint8_t)('0' + v % 10u); v /= 10u; } while (v);
    for (; k < 10; ++k) *o++ = b[k];
    return o;
}
// bytes one byte of an id takes inside the quotes
__device__ __forceinline__ uint32_t esc_len(uint8_t c) {
    if (c >= 0x20 && c != 0x7f && c != '\\' && c != '"') return 1;
    if (c == '\t' || c == '\r' || c == '\n' || c == '\\' || c == '"' || c == 0) return 2;   // (char::escape_debug prints NUL as \0)
    return c < 0x10 ? 5u : 6u;   // \u{f} / \u{1f}
}
__device__ __forceinline__ uint8_t* put_esc(uint8_t* o, uint8_t c) {
    if (c >= 0x20 && c != 0x7f && c != '\\' && c != '"') { *o++ = c; return o; }
    *o++ = '\\';
    switch (c) {
        case '\t': *o++ = 't'; break;
        case '\r': *o++ = 'r'; break;
        case '\n': *o++ = 'n'; break;
        case '\\': *o++ = '\\'; break;
        case '"': *o++ = '"'; break;
        case 0: *o++ = '0'; break;
        default: {
            *o++ = 'u'; *o++ = '{';
            const uint8_t hi = c >> 4, lo = c & 15;
            if (hi) *o++ = (uint8_t)(hi < 10 ? '0' + hi : 'a' + hi - 10);
            *o++ = (uint8_t)(lo < 10 ? '0' + lo : 'a' + lo - 10);
            *o++ = '}';
        }
    }
    return o;
}

struct RenderArgs {
    const pa_read_result* results;
    const uint32_t* arena;
    const uint8_t* ids;           // the reads' ids back to back
    const uint64_t* id_off;       // [n + 1]; or
    const uint4* rec;             // [n] {id offset, id length, -, -} into `ids` (a window's text, fastq_scan.hip) when not null
    const uint64_t* cls_off;      // [num_classes + 1] into cls_txt
    const uint8_t* cls_txt;
    uint64_t n;
    uint64_t flag_mark;           // flagged reads among the first flag_mark count in n_flagged[0], those of the j-th million behind them in n_flagged[j]
    uint64_t arena_cap;           // entries of `arena`: a launch whose arena overflowed leaves records that point beyond it (the host maps that batch
                                  // again with a larger one, PA_ERR_ARENA_FULL); such a class is rendered as empty here, never read
};
__device__ __forceinline__ void id_range(const RenderArgs& a, uint64_t i, uint64_t& b, uint64_t& e) {
    if (a.rec) { const uint4 q = a.rec[i]; b = q.x; e = (uint64_t)q.x + q.y; }
    else { b = a.id_off[i]; e = a.id_off[i + 1]; }
}
__device__ __forceinline__ uint32_t arena_len(const RenderArgs& a, const pa_read_result& r) {
    return (uint64_t)r.class_off + r.class_len <= a.arena_cap ? r.class_len : 0u;
}

__device__ __forceinline__ bool flag_of(const pa_read_result& r) {
    return (r.mismatches & PA_MAPPED_BIT) && r.coverage >= PA_READ_COVERAGE_THRESHOLD && r.class_len == 0;   // :455
}

__global__ __launch_bounds__(256) void pa_render_len_kernel(const RenderArgs a, uint32_t* __restrict__ len, unsigned long long* __restrict__ n_flagged) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t flagged = 0;
    if (i <= a.n) {
        uint32_t l = 0;
        if (i < a.n) {
            const pa_read_result r = a.results[i];
            const bool mapped = r.mismatches & PA_MAPPED_BIT;
            const bool flag = flag_of(r);
            flagged = flag;
            l = (flag ? 7u : 8u) + 2u;                                        // "(true, " / "(false, " and the id's quotes
            uint64_t b, e;
            id_range(a, i, b, e);
            for (uint64_t j = b; j < e; ++j) l += esc_len(a.ids[j]);
            l += 3u;                                                          // ", ["
            if (r.class_off & PA_CLASS_REF) {
                const uint32_t c = r.class_off & ~PA_CLASS_REF;
                l += (uint32_t)(a.cls_off[c + 1] - a.cls_off[c]);
            } else if (const uint32_t cl = arena_len(a, r)) {
                const uint32_t* ids = a.arena + r.class_off;
                l += 2u * (cl - 1);
                for (uint32_t j = 0; j < cl; ++j) l += dec_digits(ids[j]);
            }
            l += 3u + dec_digits(mapped ? r.coverage : 0u) + 2u;              // "], " coverage ")\n"  (None -> (false, id, [], 0), :461)
        }
        len[i] = l;                                                           // (len[n] = 0: the scan's last entry is the total)
    }
    // flagged reads -> their bucket (a wave's 64 consecutive reads touch at most two): one atomic per wave and bucket
    const uint32_t bucket = i < a.flag_mark ? 0u : (uint32_t)min((uint64_t)PA_RENDER_FLAG_BUCKETS - 1, 1 + (i - a.flag_mark) / 1000000ull);
    const uint32_t b0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)bucket);
    const unsigned long long m0 = __ballot(flagged && bucket == b0), m1 = __ballot(flagged && bucket != b0);
    if ((threadIdx.x & 63u) == 0) {
        if (m0) atomicAdd(n_flagged + b0, (unsigned long long)__popcll(m0));
        if (m1) atomicAdd(n_flagged + min(b0 + 1, PA_RENDER_FLAG_BUCKETS - 1), (unsigned long long)__popcll(m1));
    }
}

__global__ __launch_bounds__(256) void pa_render_write_kernel(const RenderArgs a, const uint64_t* __restrict__ off, uint8_t* __restrict__ text, uint64_t cap) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n || off[a.n] > cap) return;   // (a text that does not fit the buffer is not written at all: the host sees its length and renders it again)
    const pa_read_result r = a.results[i];
    const bool mapped = r.mismatches & PA_MAPPED_BIT;
    uint8_t* o = text + off[i];
    if (flag_of(r)) { const char s[] = "(true, "; for (int k = 0; k < 7; ++k) *o++ = (uint8_t)s[k]; }
    else { const char s[] = "(false, "; for (int k = 0; k < 8; ++k) *o++ = (uint8_t)s[k]; }
    *o++ = '"';
    uint64_t ib, ie;
    id_range(a, i, ib, ie);
    for (uint64_t j = ib; j < ie; ++j) o = put_esc(o, a.ids[j]);
    *o++ = '"'; *o++ = ','; *o++ = ' '; *o++ = '[';
    if (r.class_off & PA_CLASS_REF) {
        const uint32_t c = r.class_off & ~PA_CLASS_REF;
        const uint8_t* t = a.cls_txt + a.cls_off[c];
        const uint32_t n = (uint32_t)(a.cls_off[c + 1] - a.cls_off[c]);
        for (uint32_t j = 0; j < n; ++j) *o++ = t[j];
    } else {
        const uint32_t* ids = a.arena + r.class_off;
        for (uint32_t j = 0, cl = arena_len(a, r); j < cl; ++j) {
            if (j) { *o++ = ','; *o++ = ' '; }
            o = put_dec(o, ids[j]);
        }
    }
    *o++ = ']'; *o++ = ','; *o++ = ' ';
    o = put_dec(o, mapped ? r.coverage : 0u);
    *o++ = ')'; *o++ = '\n';
}

}  // namespace

// bytes of rocPRIM scratch the scan over n + 1 lengths needs
size_t render_scan_bytes(uint64_t n) {
    size_t bytes = 0;
    (void)rocprim::exclusive_scan(nullptr, bytes, (const uint32_t*)nullptr, (uint64_t*)nullptr, (uint64_t)0, (size_t)(n + 1), rocprim::plus<uint64_t>(), (hipStream_t) nullptr);
    return bytes;
}

// lengths + offsets of the tuples of a finished batch: d_len[n + 1], d_off[n + 1] (d_off[n] = bytes of the whole text), *d_flagged += flagged reads
int launch_render_len(const pa_read_result* d_results, const uint32_t* d_arena, const uint8_t* d_ids, const uint64_t* d_id_off, const uint4* d_rec, const uint64_t* d_cls_off,
                      const uint8_t* d_cls_txt, uint64_t n, uint64_t arena_cap, uint64_t flag_mark, uint32_t* d_len, uint64_t* d_off, unsigned long long* d_flagged, void* d_tmp,
                      size_t tmp_bytes, hipStream_t stream) {
    const RenderArgs a{d_results, d_arena, d_ids, d_id_off, d_rec, d_cls_off, d_cls_txt, n, flag_mark, arena_cap};
    hipLaunchKernelGGL(pa_render_len_kernel, dim3((uint32_t)((n + 1 + 255) / 256)), dim3(256), 0, stream, a, d_len, d_flagged);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    e = rocprim::exclusive_scan(d_tmp, tmp_bytes, (const uint32_t*)d_len, d_off, (uint64_t)0, (size_t)(n + 1), rocprim::plus<uint64_t>(), stream);
    return (int)e;
}

int launch_render_write(const pa_read_result* d_results, const uint32_t* d_arena, const uint8_t* d_ids, const uint64_t* d_id_off, const uint4* d_rec, const uint64_t* d_cls_off,
                        const uint8_t* d_cls_txt, uint64_t n, uint64_t arena_cap, const uint64_t* d_off, uint8_t* d_text, uint64_t text_cap, hipStream_t stream) {
    if (n == 0) return 0;
    const RenderArgs a{d_results, d_arena, d_ids, d_id_off, d_rec, d_cls_off, d_cls_txt, n, 0, arena_cap};
    hipLaunchKernelGGL(pa_render_write_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, a, d_off, d_text, text_cap);
    return (int)hipGetLastError();
}

}  // namespace pa
