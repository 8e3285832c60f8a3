// Record finding of process_reads on the GPU (src/pseudoaligner.rs:420-514 reads records through bio's fastq::Reader behind a mutex,
// src/utils.rs:152-157): the host only moves raw FASTQ text into pinned windows; the window is copied to HBM as it is and these
// kernels find its records in place.
//
//   pa_fq_count_kernel     line breaks per 4 KiB chunk (one workgroup per chunk, 16 bytes per lane, coalesced dwordx4 loads)
//   rocPRIM exclusive scan every chunk's first line number; the last entry is the window's number of line breaks
//   pa_fq_fill_kernel      line_start[l + 1] = position behind line break l (the chunk's prefix + a workgroup scan of the lanes' counts)
//   pa_fq_records_kernel   one lane per record (four lines): '@' / '+' in place (anything else marks the window ODD: the host then
//                          scans it with the tolerant rules of fastq.cpp), record.id() (bio 1.5: header[1..] trimmed at its end, cut at
//                          the first space), record.seq() without a CR, the longest sequence, and the bytes the whole records take
//
// Only whole records count: a window ends behind its last fourth line break, the rest is the next window's head. The encode and
// render kernels then read sequences and ids where they lie in the window (rec = {id offset, id length, sequence offset, length}).
#include <hip/hip_runtime.h>
#include <rocprim/device/device_scan.hpp>

#include "kernels.hpp"
#include "pa_common.hpp"

namespace pa {
namespace {

constexpr uint32_t FQ_CHUNK = 4096;   // bytes per workgroup: 256 lanes x 16 bytes

// 0x80 in every byte of v that equals '\n' (exact per byte: no borrow runs into a neighbour)
__device__ __forceinline__ uint32_t nl_flags(uint32_t v) {
    const uint32_t x = v ^ 0x0A0A0A0Au;
    return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);
}

// the 16 bytes of lane `t` of chunk `c`: flags of the line breaks among them that lie in [begin, end)
__device__ __forceinline__ void lane_flags(const uint8_t* __restrict__ text, uint64_t base16, uint64_t begin, uint64_t end, uint32_t f[4], uint64_t& pos) {
    pos = base16 + (uint64_t)blockIdx.x * FQ_CHUNK + 16ull * threadIdx.x;
    f[0] = f[1] = f[2] = f[3] = 0;
    if (pos >= end || pos + 16 <= begin) return;
    const uint4 v = *reinterpret_cast<const uint4*>(text + pos);
    f[0] = nl_flags(v.x); f[1] = nl_flags(v.y); f[2] = nl_flags(v.z); f[3] = nl_flags(v.w);
    if (pos < begin || pos + 16 > end) {   // the two lanes at the window's ends: bytes outside it do not count
        for (uint32_t j = 0; j < 16; ++j) {
            const uint64_t p = pos + j;
            if (p < begin || p >= end) f[j >> 2] &= ~(0x80u << (8 * (j & 3)));
        }
    }
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void pa_fq_count_kernel(const uint8_t* __restrict__ text, uint64_t base16, uint64_t begin, uint64_t end, uint32_t* __restrict__ chunk_count,
                                                          uint32_t n_chunks) {
    __shared__ uint32_t ws[4];
    uint32_t f[4];
    uint64_t pos;
    lane_flags(text, base16, begin, end, f, pos);
    const uint32_t c = wave_sum((uint32_t)(__popc(f[0]) + __popc(f[1]) + __popc(f[2]) + __popc(f[3])));
    if ((threadIdx.x & 63u) == 0) ws[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        chunk_count[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
        if (blockIdx.x == 0) chunk_count[n_chunks] = 0;   // (the scan's last entry is the total)
    }
}

__global__ __launch_bounds__(256) void pa_fq_fill_kernel(const uint8_t* __restrict__ text, uint64_t base16, uint64_t begin, uint64_t end, const uint32_t* __restrict__ chunk_first,
                                                         uint32_t* __restrict__ line_start, uint64_t cap_lines, FqInfo* __restrict__ info) {
    __shared__ uint32_t ws[4];
    uint32_t f[4];
    uint64_t pos;
    lane_flags(text, base16, begin, end, f, pos);
    const uint32_t mine = (uint32_t)(__popc(f[0]) + __popc(f[1]) + __popc(f[2]) + __popc(f[3]));
    uint32_t incl = mine;   // inclusive scan over the wave
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o, 64);
        if ((int)(threadIdx.x & 63u) >= o) incl += up;
    }
    if ((threadIdx.x & 63u) == 63u) ws[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t before = incl - mine;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) before += ws[w];
    uint64_t li = (uint64_t)chunk_first[blockIdx.x] + before;   // number of this lane's first line break
    if (blockIdx.x == 0 && threadIdx.x == 0) line_start[0] = 0;
    bool over = false;
    for (uint32_t d = 0; d < 4; ++d)
        for (uint32_t m = f[d]; m; m &= m - 1) {
            const uint32_t byte = d * 4 + ((uint32_t)__ffs((int)m) - 1) / 8;
            if (li + 1 < cap_lines) line_start[li + 1] = (uint32_t)(pos + byte + 1 - begin);
            else over = true;
            ++li;
        }
    if (over) atomicOr(&info->overflow, 1u);
}

__global__ __launch_bounds__(256) void pa_fq_records_kernel(const uint8_t* __restrict__ text, uint64_t begin, const uint32_t* __restrict__ chunk_first, uint32_t n_chunks,
                                                            const uint32_t* __restrict__ line_start, uint64_t cap_lines, uint4* __restrict__ rec, uint64_t cap_recs,
                                                            FqInfo* __restrict__ info) {
    const uint64_t lines = chunk_first[n_chunks];
    const uint64_t n = lines / 4;
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r == 0) {
        info->lines = lines;
        info->n = n;
        info->consumed = 4 * n < cap_lines ? line_start[4 * n] : 0;
        if (4 * n >= cap_lines || n > cap_recs) atomicOr(&info->overflow, 1u);
    }
    uint32_t seq_len = 0;
    bool odd = false;
    if (r < n && 4 * r + 3 < cap_lines && r < cap_recs) {
        const uint8_t* const t = text + begin;
        const uint32_t p0 = line_start[4 * r], p1 = line_start[4 * r + 1], p2 = line_start[4 * r + 2];
        const uint32_t e0 = p1 - 1, e1 = p2 - 1;            // the line breaks that end the header and the sequence line
        odd = t[p0] != '@' || t[p2] != '+';
        // record.id() (:456): header[1..] without trailing white space, up to its first space (a tab stays part of the id)
        uint32_t hend = e0, ide = p0 + 1;
        while (hend > p0 + 1 && (t[hend - 1] == '\r' || t[hend - 1] == ' ' || t[hend - 1] == '\t' || t[hend - 1] == '\n')) --hend;
        while (ide < hend && t[ide] != ' ') ++ide;
        const uint32_t id_len = e0 > p0 ? ide - (p0 + 1) : 0u;
        const uint32_t len = e1 - p1;
        seq_len = (len && t[e1 - 1] == '\r') ? len - 1 : len;   // record.seq() (:449): a CR before the line break is not sequence
        rec[r] = make_uint4((uint32_t)(begin + p0 + 1), id_len, (uint32_t)(begin + p1), seq_len);
    }
    uint32_t mx = seq_len;
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_down(mx, o, 64));
    if ((threadIdx.x & 63u) == 0 && mx) atomicMax(&info->max_seq, mx);
    if (__ballot(odd) && (threadIdx.x & 63u) == 0) atomicOr(&info->odd, 1u);
}

// DnaString::from_dna_string (:450) for sequences that lie in the window's text: one thread = one 64-bit word of one read
__global__ __launch_bounds__(256) void pa_encode_rec_kernel(const uint8_t* __restrict__ text, const uint4* __restrict__ rec, uint64_t n_reads, uint32_t wpr,
                                                            uint64_t* __restrict__ tiles, uint32_t* __restrict__ lens) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t ntiles = (n_reads + 63) >> 6;
    if (gid >= ntiles * wpr * 64) return;
    const uint32_t r = (uint32_t)(gid & 63);
    const uint64_t tw = gid >> 6;
    const uint32_t w = (uint32_t)(tw % wpr);
    const uint64_t rid = (tw / wpr) * 64 + r;
    uint64_t v = 0;
    if (rid < n_reads) {
        const uint4 q = rec[rid];
        uint32_t len = q.w;
        if (len > wpr * 32u) len = wpr * 32u;
        if (w == 0) lens[rid] = len;
        const uint32_t b0 = 32u * w;
        const uint32_t nb = len > b0 ? (len - b0 < 32u ? len - b0 : 32u) : 0u;
        const uint8_t* src = text + q.z + b0;
        for (uint32_t j = 0; j < nb; ++j) {
            const uint8_t c = src[j] & 0xDF;   // upper-case
            const uint64_t code = c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 0u;
            v |= code << (2 * j);
        }
    }
    tiles[gid] = v;
}

}  // namespace

uint32_t fq_chunks(uint64_t begin, uint64_t end) {
    const uint64_t base16 = begin & ~15ull;
    return (uint32_t)((end - base16 + FQ_CHUNK - 1) / FQ_CHUNK);
}

size_t fq_scan_tmp_bytes(uint32_t n_chunks) {
    size_t bytes = 0;
    (void)rocprim::exclusive_scan(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (size_t)n_chunks + 1, rocprim::plus<uint32_t>(), (hipStream_t) nullptr);
    return bytes;
}

// The records of the window text[begin, end): d_info (zeroed here) receives what the host needs to go on. d_chunk / d_first hold n_chunks + 1
// u32 each. `rescan`: the counts and prefixes of an earlier call on the same window are still valid (line_start / rec were too small).
int launch_fq_scan(const uint8_t* d_text, uint64_t begin, uint64_t end, uint32_t* d_chunk, uint32_t* d_first, void* d_tmp, size_t tmp_bytes, uint32_t* d_line_start,
                   uint64_t cap_lines, uint4* d_rec, uint64_t cap_recs, FqInfo* d_info, bool rescan, hipStream_t stream) {
    const uint64_t base16 = begin & ~15ull;
    const uint32_t n_chunks = fq_chunks(begin, end);
    hipError_t e = hipMemsetAsync(d_info, 0, sizeof(FqInfo), stream);
    if (e != hipSuccess) return (int)e;
    if (!rescan) {
        hipLaunchKernelGGL(pa_fq_count_kernel, dim3(n_chunks), dim3(256), 0, stream, d_text, base16, begin, end, d_chunk, n_chunks);
        if ((e = hipGetLastError()) != hipSuccess) return (int)e;
        e = rocprim::exclusive_scan(d_tmp, tmp_bytes, (const uint32_t*)d_chunk, d_first, 0u, (size_t)n_chunks + 1, rocprim::plus<uint32_t>(), stream);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(pa_fq_fill_kernel, dim3(n_chunks), dim3(256), 0, stream, d_text, base16, begin, end, (const uint32_t*)d_first, d_line_start, cap_lines, d_info);
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    const uint64_t max_recs = std::min<uint64_t>(cap_recs, cap_lines / 4) + 1;   // (lane 0 writes the summary even when the window holds no record)
    hipLaunchKernelGGL(pa_fq_records_kernel, dim3((uint32_t)((max_recs + 255) / 256)), dim3(256), 0, stream, d_text, begin, (const uint32_t*)d_first, n_chunks,
                       (const uint32_t*)d_line_start, cap_lines, d_rec, cap_recs, d_info);
    return (int)hipGetLastError();
}

int launch_encode_rec(const uint8_t* d_text, const uint4* d_rec, uint64_t n, uint32_t wpr, uint64_t* tiles, uint32_t* lens, hipStream_t stream) {
    const uint64_t threads = ((n + 63) >> 6) * wpr * 64;
    if (threads == 0) return 0;
    hipLaunchKernelGGL(pa_encode_rec_kernel, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, stream, d_text, d_rec, n, wpr, tiles, lens);
    return (int)hipGetLastError();
}

}  // namespace pa
