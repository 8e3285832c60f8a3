// pa_resolve_kernel — the content lookups of a mapping launch, after the launch.
//
// A read whose class is the intersection of several visited classes (nodes_to_eq_class, src/pseudoaligner.rs:323-356) ends in
// one of two ways: the intersection IS one of the classes seen (the map kernel knows which: returned by reference, counted by
// class id), or it is a strict subset of all of them. The second kind still may equal SOME class of the index — that decides
// its slot in the class-count table and whether the record can point at the index class — and finding out is a hash lookup by
// content: a rare (3 % of the config-3 reads), latency-bound operation. Round 2 ran it inside the pool scheduler as a state of
// its own: ~11 lanes per step, two dependent round trips, 9 % of the wave time. Here the map kernel only appends a 32-byte
// entry per such read (map_pool.hip, append_deferred) and this kernel resolves them at full width, one thread per entry:
//
//   window entry  {rid, coverage, mismatches, DEFER_WINDOW | count} {base1, mask1, base2, mask2}
//                 canonical windows -> window table (one line). An index class: record by reference. Else: the ids are written
//                 to the class arena (space taken with one atomic per wave) and the record points there.
//   list entry    {rid, coverage, mismatches, DEFER_LIST | count} {arena offset, ...}   record and ids are already written:
//                 only the class (count key, colour, novel list) is looked up in the class-list hash table.
//
// Every entry that resolves to an index class yields one count key (written right behind the map kernel's key stream; padding
// entries yield padding keys); the others are counted here, in the table's "novel" slot (one atomic per workgroup);
// optionally the class id in colour_out[rid], and — when an overflow table is attached — a novel-list entry for classes that
// are no index class.
#include <hip/hip_runtime.h>

#include "kernel_utils.hpp"
#include "kernels.hpp"
#include "lane_steps.hpp"

namespace pa {
namespace {

constexpr uint32_t NO_KEY = 0xFFFFFFFFu;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr uint32_t RV_PER = 4;   // entries per lane and iteration: their loads are independent and in flight together (the
                                 // lookup is three dependent round trips — entry, window-table line, record — and nothing else)

__global__ __launch_bounds__(256) void pa_resolve_kernel(const MapParams p, uint64_t defer_cap, uint64_t keys_cap) {
    __shared__ unsigned long long s_novel;   // reads of this workgroup whose class is no index class (one slot of the count table: counted here,
                                             // not as keys — a million identical keys at the end of the stream serialise the count kernel's LDS atomics)
    if (threadIdx.x == 0) s_novel = 0;
    __syncthreads();
    const unsigned long long top = *p.defer_top;
    const uint64_t n = top < defer_cap ? top : defer_cap;
    const uint32_t lane = lane_id();
    const bool counting = p.keys != nullptr;
    // this kernel's keys go right behind the map kernel's chunks: the count kernels read ONE contiguous stream
    uint32_t* keys_b = nullptr;
    if (counting) { const unsigned long long kt = *p.keys_top; keys_b = p.keys + (kt < keys_cap ? kt : keys_cap); }
    // arena space for the window results that are no index class: a private slice per wave, taken PA_ARENA_CHUNK ids at a time
    // (one atomic per wave and iteration on the one counter is 80 k dependent atomics on a hot word: 0.7 ms)
    unsigned long long a_cur = 0, a_end = 0;
    uint32_t my_novel = 0;
    const uint64_t wave0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) & ~63ull, nwave_lanes = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i0 = wave0 * RV_PER; i0 < n; i0 += nwave_lanes * RV_PER) {   // whole waves: the scans and ballots below need every lane
        u32x4 e0[RV_PER], e1[RV_PER];
        bool in[RV_PER], live[RV_PER];
        uint64_t idx[RV_PER];
#pragma unroll
        for (uint32_t j = 0; j < RV_PER; ++j) {   // entry i0 + 64 j + lane: coalesced
            idx[j] = i0 + 64ull * j + lane;
            in[j] = idx[j] < n;
            e0[j] = in[j] ? reinterpret_cast<const u32x4*>(p.defer)[2 * idx[j]] : u32x4{NO_KEY, 0u, 0u, 0u};
        }
#pragma unroll
        for (uint32_t j = 0; j < RV_PER; ++j) {
            live[j] = e0[j].x != NO_KEY;
            e1[j] = live[j] ? reinterpret_cast<const u32x4*>(p.defer)[2 * idx[j] + 1] : u32x4{0u, 0u, 0u, 0u};
        }
        uint32_t colour[RV_PER], want[RV_PER];
        uint32_t wsum = 0;
#pragma unroll
        for (uint32_t j = 0; j < RV_PER; ++j) {   // (the compiler hoists the table loads of the four lookups above their compares)
            colour[j] = NO_CLASS;
            const uint32_t count = e0[j].w & 0x3FFFFFFFu;
            if (live[j] && (e0[j].w & PA_DEFER_WINDOW)) {
                uint32_t b1 = e1[j].x, m1 = e1[j].y, b2 = e1[j].z, m2 = e1[j].w;
                window_canon(b1, m1, b2, m2);
                colour[j] = window_class(p.ix, b1, m1, b2, m2);
            } else if (live[j]) {
                colour[j] = class_of_list(p.arena + e1[j].x, count, p.ix, p.class_table, p.class_table_size);
            }
            want[j] = (live[j] && (e0[j].w & PA_DEFER_WINDOW) && colour[j] == NO_CLASS) ? count : 0u;   // window results that are no index class: ids to the arena
            wsum += want[j];
        }
        const uint32_t incl = wave_incl_scan(wsum);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (total && a_cur + total > a_end) {   // (what is left of the old slice stays unused, as in the map kernel)
            const unsigned long long take = total > PA_ARENA_CHUNK ? total : PA_ARENA_CHUNK;
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(p.arena_top, take);
            a_cur = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            a_end = a_cur + take;
        }
        unsigned long long my_off = a_cur + (incl - wsum);
        a_cur += total;
#pragma unroll
        for (uint32_t j = 0; j < RV_PER; ++j) {
            const uint32_t rid = e0[j].x, cov = e0[j].y, mm = e0[j].z, count = e0[j].w & 0x3FFFFFFFu;
            const bool window = live[j] && (e0[j].w & PA_DEFER_WINDOW);
            bool listed = live[j] && colour[j] == NO_CLASS;   // goes on the novel list (its ids are in the arena)
            uint32_t arena_off = e1[j].x;
            if (window) {
                uint32_t class_off = PA_CLASS_REF | colour[j];
                if (want[j]) {
                    class_off = (uint32_t)my_off;
                    arena_off = (uint32_t)my_off;
                    if (my_off + count > p.arena_cap) { atomicOr(p.status, PA_STATUS_ARENA_FULL); listed = false; }
                    else {
                        uint32_t* dst = p.arena + my_off;
                        uint32_t k = 0;
                        for (uint32_t t = e1[j].y; t; t &= t - 1) dst[k++] = e1[j].x + (uint32_t)(__ffs((int)t) - 1);
                        for (uint32_t t = e1[j].w; t; t &= t - 1) dst[k++] = e1[j].z + (uint32_t)(__ffs((int)t) - 1);
                    }
                    my_off += count;
                }
                reinterpret_cast<u32x4*>(p.results)[rid] = u32x4{cov, mm | PA_MAPPED_BIT, class_off, count};
            }
            if (live[j] && p.colour_out) p.colour_out[rid] = colour[j];
            if (live[j] && colour[j] == NO_CLASS) ++my_novel;
            if (counting && in[j]) keys_b[idx[j]] = (live[j] && colour[j] != NO_CLASS) ? colour[j] : NO_KEY;
            if (p.novel_list) {   // a class no index class equals: remember where its ids are (one atomic per wave)
                const uint64_t m = __ballot(listed);
                if (m) {
                    const uint32_t first = (uint32_t)(__ffsll((unsigned long long)m) - 1);
                    unsigned long long nb = 0;
                    if (lane == first) nb = atomicAdd(p.novel_ctr, (unsigned long long)__popcll(m));
                    nb = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(nb >> 32), (int)first) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)nb, (int)first);
                    if (listed) {
                        const unsigned long long at = nb + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                        if (at < p.novel_cap) {
                            p.novel_list[2 * at] = arena_off;
                            p.novel_list[2 * at + 1] = count;
                        } else atomicOr(p.novel_status, PA_NOVEL_LIST_FULL);
                    }
                }
            }
        }
    }
    if (counting) {   // the novel slot of the table: one atomic per workgroup
        if (my_novel) atomicAdd(&s_novel, (unsigned long long)my_novel);
        __syncthreads();
        if (threadIdx.x == 0 && s_novel) atomicAdd(p.counts + p.ix.num_classes, s_novel);
    }
}

}  // namespace

uint64_t defer_capacity(uint64_t n_reads, uint32_t nwaves) {   // entries: every read can be deferred; chunks are filled to the last entry, a wave leaves one partly used
    return (n_reads / PA_DEFER_CHUNK + nwaves + 2) * PA_DEFER_CHUNK;
}

int launch_resolve(const MapParams& p, uint64_t defer_cap, uint64_t keys_cap, int num_cus, hipStream_t stream) {
    const uint32_t cus = num_cus > 0 ? (uint32_t)num_cus : 256u;
    hipLaunchKernelGGL(pa_resolve_kernel, dim3(cus * 4), dim3(256), 0, stream, p, defer_cap, keys_cap);
    return (int)hipGetLastError();
}

}  // namespace pa
