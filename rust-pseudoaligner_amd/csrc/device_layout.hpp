// GPU-resident form of the index and the per-lane state of the mapping kernel. Shared by the host flattener
// (device_flatten.cpp), the HIP kernels (map_pool.hip) and the host lane emulator used by the CPU tests (tests/emu).
//
// HBM layout (all little-endian, all read-only after pa_index_create):
//
//   dictionary  (k <= 32) open addressing over 16-byte SLOTS, four to a 64-byte bucket line:
//                 slot = {u32 key_lo, u32 key_hi, u32 handle (0xFFFFFFFF = empty), u32 off (bits 0..23) | ~flags << 24}
//               bucket = mulhi32(fmix64(key) >> 32, nbuckets); every key also has a HOME slot in its bucket, j = fmix64(key) & 3.
//               A key sits in its home slot when that was free (78 % of the keys at load 0.5: the builders place all home
//               keys first), else in another slot t of the bucket — then flag bit ((t - j - 1) & 3) of the HOME slot says so —
//               else it overflows into the next bucket (flag bit 3 of the home slot) where the same rule applies. Flags are
//               stored inverted (a set flag is a cleared bit) so that an all-ones fill is "empty, no flags". A lookup is ONE
//               16-byte load — the home slot holds the whole key and the answer — and only when the home slot holds another
//               key AND names other slots, one more load from the same line (21 % of the hits, 9 % of the misses). Round 2's
//               line {fp[4], entries[4]} took two dependent loads for every hit, and on this chip the second load of a line that
//               lives in HBM costs almost what the first did (tools/microbench/gather_multi.hip: 47 G lines/s with one load per
//               lane, 33 G with a dependent second one). The dictionary stores whole keys, so unlike the reference's MPHF
//               (src/pseudoaligner.rs:96-107) no node sequence has to be fetched to confirm a hit.
//               k > 32 (two-word k-mers): a line holds two whole entries {key word 0..3, handle, off, -, -}, handle
//               0xFFFFFFFF = empty, load <= 1/3, linear probing over lines (a line with a free entry ends the probe sequence).
//   node blobs  one blob per unitig, starting on a 128-byte block, addressed by handle = byte offset / 64 — so bit 0 of a blob's
//               offset/64 is always clear, and every handle (dictionary slots, edges, lane state) carries there the WIDE flag
//               of its node: the third 16-byte vector of the header is needed (second class window, or no windows at all).
//               The blob address is (handle & ~1) * 64; nid_of_handle / ledge are indexed by the handle as it is.
//                 +0  u32 len (bits 0..23) | debruijn::Exts byte (bits 24..31)     +4  u32 class id
//                 +8  u32 cmin, cmask   the class as WINDOWS of 32 transcript ids: {cmin + i : bit i of cmask} ...
//                 +16 u32 redge[4]      handle of the node reached by right-extending with base b (Node::r_edges)
//                 +32 u32 cmin2, cmask2 ... U {cmin2 + i : bit i of cmask2}, cmin2 >= cmin + 32 (cmask2 = 0: one window);
//                                       cmask = 0 when the class does not fit (then only the id list describes it)
//                 +40 u32 class record ref   +44 u32 class length (ids)
//               A forward step loads +0 and +16 for every lane and +32 only for lanes whose handle has the WIDE flag or whose
//               read collects class lists (list mode): one access of the vector L1 less for ~90 % of the node visits.
//                 +48 u64 seq[ceil(len/32)]  2-bit packed, LSB-first
//               so a node visit is ONE dependent fetch (header and the first 64 bases share a line), the hop to the
//               next node needs no further lookup (the reference re-derives every edge by hashing: SURVEY.md §3.2),
//               the colour's id list is addressable without an offsets table, and for window classes (transcripts of
//               one gene are neighbours in the FASTA; a second window covers a paralog or an overlapping gene) the
//               intersection of nodes_to_eq_class is an AND of masks that never touches the id lists.
//   ledge       u32[8*granules] by blob handle: {handle, length}[4] of the nodes reached by left-extending with base b (Node::l_edges;
//               handle 0xFFFFFFFF = none). Only the left extension touches it; the length lets a lane that hops left fetch the END
//               of the neighbour's sequence together with its header (lane_steps.hpp, left_issue)
//   nid_of_handle  u32[granules] node id by blob handle (only the node-trace test surface reads it)
//   ec          class records, 16-byte aligned, at least 32 bytes, padded with 0xFFFFFFFF: record r = words [4r, ...) =
//               {class id, id0, id1, ...} — the sorted transcript-id lists of eq_classes: Vec<Vec<u32>>
//               (src/pseudoaligner.rs:29); a class of <= 7 ids is two 16-byte loads and needs no length checks.
//   class_ref/class_len  u32[num_classes] record ref and length by class id (only the count table's content lookup)
//   wtable      window classes by content: open addressing over 64-byte lines of three {cmin, cmask, cmin2, cmask2, class
//               id} entries (class id 0xFFFFFFFF = empty), line = mulhi32(hash(windows), wbuckets), linear probing —
//               tells in ONE fetch whether a window result that is a strict subset of every class seen is itself a class
#pragma once
#include <cstdint>

#include "../../include/pseudoaligner_amd.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PA_HD __host__ __device__ __forceinline__
#else
#define PA_HD inline
#endif

namespace pa {

constexpr uint32_t NO_HANDLE = 0xFFFFFFFFu;
constexpr uint32_t BLOB_GRANULE = 64;
constexpr uint32_t HANDLE_WIDE = 1u;   // bit 0 of a handle: the node's header vector at +32 is needed (device_layout.hpp, node blobs)
constexpr uint32_t BLOB_ALIGN = 128;   // the memory system moves 128-byte blocks: two adjacent 64-byte lines of ONE block cost what one line costs, lines of two blocks cost double (tools/microbench/gather_pair.hip); header + first 320 bases = one block
constexpr uint32_t BLOB_HDR_BYTES = 48;
constexpr uint32_t CLASS_WINDOW = 32;   // ids per class window (one mask word)
constexpr uint32_t SLOTS_PER_BUCKET = 4;
constexpr uint32_t BUCKET_WORDS = 16;
constexpr uint32_t SLOT_WORDS = 4;                 // k <= 32: {key_lo, key_hi, handle, off | ~flags << 24}
constexpr uint32_t SLOT_OFF_MASK = 0xFFFFFFu;      // node length < 2^24
constexpr uint32_t SLOT_FLAG_SHIFT = 24;           // flags 0..2: slot (home + 1 + i) & 3 holds a key of this home; flag 3: one overflowed to the next bucket
constexpr uint32_t SLOT_FLAG_OVERFLOW = 8u;
constexpr uint32_t DICT_MAX_PROBES = 15;           // buckets a key may overflow through (the builders keep every chain shorter)

struct alignas(16) U4 {
    uint32_t x, y, z, w;
};
struct alignas(8) Q2 {   // two sequence words, 8-byte aligned (global_load_dwordx4)
    uint64_t a, b;
};

constexpr uint32_t WT_ENTRIES = 3;   // window-table entries per 64-byte line, 5 words each

struct DevIndexView {
    const uint32_t* table;    // nbuckets * 16 words
    uint64_t nbuckets;
    const uint8_t* blobs;     // node blobs
    const uint32_t* ledge;    // [8 * granules]: {handle, length}[4] by handle
    const uint32_t* nid_of_handle;   // [granules]
    const uint32_t* ec;       // class records (16-byte aligned records of u32)
    const uint32_t* class_ref;   // [num_classes]
    const uint32_t* class_len;   // [num_classes]
    const uint32_t* wtable;      // wbuckets * 16 words
    uint32_t wbuckets;
    uint64_t kmask;           // k <= 32: mask of the k-mer word
    uint64_t kmask_hi;        // k > 32: mask of the second k-mer word (bases 32..k-1)
    uint32_t k;
    uint32_t num_nodes, num_classes;
};

}  // namespace pa
