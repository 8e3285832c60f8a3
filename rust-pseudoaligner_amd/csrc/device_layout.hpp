// GPU-resident form of the index and the per-lane state of the mapping kernel. Shared by the host flattener
// (device_flatten.cpp), the HIP kernels (map_pool.hip) and the host lane emulator used by the CPU tests (tests/emu).
//
// HBM layout (all little-endian, all read-only after pa_index_create):
//
//   dictionary  (k <= 32) open addressing over 16-byte SLOTS, four to a 64-byte bucket line:
//                 slot = {u32 key_lo, u32 key_hi, u32 handle (0xFFFFFFFF = empty), u32 off (bits 0..23) | ~flags << 24}
//               bucket = mulhi32(x, nbuckets) with x = a 32-bit multiplicative mix of the key's two words (lane_steps.hpp,
//               pa_bucket_home); every key also has a HOME slot in its bucket, j = x & 3.
//               A key sits in its home slot when that was free (88 % of the keys at the load of 0.25 the table is built with;
//               78 % at 0.5: the builders place all home keys first), else in another slot t of the bucket — then flag bit ((t - j - 1) & 3) of the HOME slot says so —
//               else it overflows into the next bucket (flag bit 3 of the home slot) where the same rule applies. Flags are
//               stored inverted (a set flag is a cleared bit) so that an all-ones fill is "empty, no flags". A lookup is ONE
//               16-byte load — the home slot holds the whole key and the answer — and only when the home slot holds another
//               key AND names other slots does the lane go on, ONE named slot per step (11 % of the hits at load 0.25): a probe
//               never waits for a second load of its own, and a wave never for the one lane that needs it. The
//               dictionary stores whole keys, so unlike the reference's MPHF (src/pseudoaligner.rs:96-107) no node sequence
//               has to be fetched to confirm a hit. The table is SPARSE on purpose (DICT_LOAD = 0.25: 64 bytes per k-mer, 6.6 GB
//               at config 3 of 288 GB): the second load and the overflow bucket are dependent round trips of a whole wave's
//               step, and halving the load from 0.5 takes 4 % (config 3) / 2 % (config 5) off the mapping kernel's time
//               (profiles/r05_dict_load.txt). One 16-byte load per probe is also what keeps a table of this size usable at all:
//               beyond ~4 GB of randomly accessed footprint every load instruction is an address-translation miss, and
//               several loads per random block then run at a quarter of the rate (profiles/r05_tlb_footprint.txt).
//               k > 32 (two-word k-mers): a line holds two whole entries {key word 0..3, handle, off, -, -}, handle
//               0xFFFFFFFF = empty, load <= 1/3, linear probing over lines (a line with a free entry ends the probe sequence).
//               What an entry says (both forms): handle = the chain BLOCK the k-mer starts in (below), off = where:
//                 bits 0..5  p      position of the k-mer's first base in that block's window (0..63)
//                 bit  6     the k-mer is the FIRST k-mer of its node (kmer_offset == 0: the quirk of :129 needs to know)
//                 bits 7..8  min(block index within the chain, 3): how many blocks a left extension may step back at once
//                 bits 9..10 which of the block's four slots holds the record of the k-mer's node: the forward step loads the
//                            slots ROTATED by it (slot 0 of what it loads = that record) instead of looking for it afterwards
//
//   chain blocks  Unitigs that the reference's graph cuts ONLY because the colour changes (A has one right extension, it
//               leads to B, B has one left extension: 64-69 % of the nodes of a transcriptome) are laid out as ONE sequence,
//               a CHAIN: node i+1 starts K-1 bases before node i ends, so the chain is A's sequence followed by the bases B
//               adds, and node i covers chain positions [s_i, e_i), s_{i+1} = e_i - (K-1). A read that walks A -> B in the
//               reference (has_ext + r_edges + get_node, :267-283) here just goes on comparing: position e_i is the base the
//               extension test looks at, the positions between two such bases are one node's compare loop (:236-255).
//               A chain is stored as OVERLAPPING 128-byte blocks: block j holds the 256 bases [64 j, 64 j + 256) and the
//               records of every node that overlaps them, so whatever a 150-base read needs of a chain — sequence, node
//               boundaries, classes, edges at the chain's end — arrives with ONE 128-byte request (the unit the memory
//               system moves; the kernel is bound by the rate of such requests: DESIGN.md §4) and is consumed by ONE step.
//               The price is replication (every base is stored four times: 2 bytes per base, 0.2 GB at config 3).
//               handle = byte offset / 128; the block of chain position c is handle(chain) + c / 64.
//                 +0   four 16-byte SLOTS
//                 +64  u64 seq[8]   bases [64 j, 64 j + 256) of the chain, 2-bit packed, LSB-first, zero beyond the chain's end
//               Slots, in order: one RECORD per node with e > 64 j and s < 64 j + 256 (ascending), each followed by its
//               extension slot if WIDE, the chain's last record also by the edge slot if it has one (and so a BRANCH record:
//               EDGES without LAST, see TAILS):
//                 record     {w0, class id, cmin, cmask}   the class as a WINDOW of 32 transcript ids {cmin + i : bit i of cmask}
//                            w0 bits 0..15  e - 64 j, the node's end relative to the block (0xFFFF: further than that)
//                               bit 16 WIDE     the next slot is this record's extension
//                               bit 17 LAST     last node of the chain
//                               bit 18 EDGES    (LAST) the slot after this record (and its extension) holds the right edges
//                               bit 19 LINK     (LAST) a copy cut short: that slot holds {block, position} of the next base in the node's own chain
//                               slot 0 only: bits 20..21 min(j, 3); bits 28..31 which of the four slots are records
//                 extension  {cmin2, cmask2, class record ref, class length}: a second window (cmin2 >= cmin + 32), or — cmask
//                            == 0 — a class that does not fit two windows and is only described by its id list
//                 edges      {handle of the chain reached by right-extending the chain's last node with base b}[4]
//                            (Node::r_edges; 0xFFFFFFFF = has_ext(Right, b) is false)
//               The flattener only merges B onto a chain when every block still fits its four slots; a node on its own
//               always does (record + extension + edges).
//               TAILS. Most nodes with ONE right extension lead to a node that others lead to as well (a join): that node
//               starts a chain of its own, and the walk would leave the block. So after a chain's last node Z the flattener
//               appends COPIES of what can only follow — Z's one successor, that node's one successor, ... — for up to 128
//               bases (what a 150-base read can still need), records and sequence like any other node of the chain. A copy
//               that does not fit whole ends in a LINK: at that position the walk goes on, mid-node, in the node's own chain.
//               A node with SEVERAL right extensions is followed by a copy of the FAVOURED one (the successor most of its
//               transcripts go on to): its record is a BRANCH record — EDGES set, LAST not — followed by its edge slot and then
//               the copy's record. A read whose next base is the copy's goes on in the block; any other base takes the edge
//               slot (has_ext / r_edges, :267-283, as at a chain's end).
//               The dictionary, the left edges and the right edges only ever point at a node's own place, never at a copy.
//   ledge       u32[8 * blocks] by CHAIN handle: {block handle, y + 1}[4] — where a left extension that leaves the chain's
//               first node with base b goes on (Node::l_edges): the last k-mer of the neighbour chain's last node, as position
//               y of a block that has (up to) 192 bases to the left of it. Handle 0xFFFFFFFF = has_ext(Left, b) is false.
//   seg_g / seg_nid  (node-trace test surface only) u64 64 * handle(chain) + s_i of every node, ascending, and its node id
//   ec          class records, 16-byte aligned, at least 32 bytes, padded with 0xFFFFFFFF: record r = words [4r, ...) =
//               {class id, id0, id1, ...} — the sorted transcript-id lists of eq_classes: Vec<Vec<u32>>
//               (src/pseudoaligner.rs:29); a class of <= 7 ids is two 16-byte loads and needs no length checks.
//               A class WITHOUT windows (its ids do not fit two windows of 32 consecutive ids) of at least bitmap_min ids is followed by its
//               membership BITMAP: bitmap_words u32, bit t = transcript t belongs to the class (class_bitmap below). Records are found
//               through class_ref / the chain blocks' extension slots only, never by walking from one record to the next.
//   class_ref/class_len  u32[num_classes] record ref and length by class id (list mode: the record of a one-window class)
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
constexpr uint32_t CH_BLOCK = 128;          // bytes per chain block = the unit the memory system moves (tools/microbench/gather_pair.hip)
constexpr uint32_t CH_STRIDE_LOG2 = 6;
constexpr uint32_t CH_STRIDE = 1u << CH_STRIDE_LOG2;   // chain positions between consecutive blocks
constexpr uint32_t CH_WINDOW = 256;         // bases a block holds
constexpr uint32_t CH_SLOTS = 4;            // 16-byte slots per block
constexpr uint32_t CH_SEQ_BYTES = 64;       // offset of the sequence words in a block
constexpr uint32_t CH_BACK_MAX = 3;         // blocks a left extension steps back at once (window / stride - 1)
constexpr uint32_t SEG_E_MASK = 0xFFFFu, SEG_E_FAR = 0xFFFFu;
constexpr uint32_t SEG_WIDE = 1u << 16, SEG_LAST = 1u << 17, SEG_EDGES = 1u << 18, SEG_LINK = 1u << 19;
constexpr uint32_t CH_TAIL = 128;           // bases of copied successor nodes after a chain's last node
constexpr uint32_t SEG_BACK_SHIFT = 20, SEG_RECMASK_SHIFT = 28;
constexpr uint32_t ENT_P_MASK = 63u, ENT_NODE_START = 64u, ENT_BACK_SHIFT = 7, ENT_CUR_SHIFT = 9;   // dictionary entry, word `off`
// the slot of block `blk` that holds the first record whose node ends beyond window position y (the node of the k-mer ending at y)
PA_HD uint32_t block_slot_of(const uint8_t* blk, uint32_t y) {
    const uint32_t* sl = reinterpret_cast<const uint32_t*>(blk);
    const uint32_t recmask = sl[0] >> SEG_RECMASK_SHIFT;
    for (uint32_t t = 0; t < CH_SLOTS; ++t)
        if (((recmask >> t) & 1u) && (sl[4 * t] & SEG_E_MASK) > y) return t;
    return 0;   // (no such record: a malformed block; the builders' self-checks report it)
}
// second word of the dictionary entry of the k-mer that starts at chain position c (node_start: it is its node's first k-mer;
// slot: block_slot_of(its block, (c & 63) + k - 1))
PA_HD uint32_t dict_entry_off(uint32_t c, bool node_start, uint32_t slot) {
    const uint32_t j = c >> CH_STRIDE_LOG2;
    return (c & ENT_P_MASK) | (node_start ? ENT_NODE_START : 0u) | ((j < CH_BACK_MAX ? j : CH_BACK_MAX) << ENT_BACK_SHIFT) | (slot << ENT_CUR_SHIFT);
}
constexpr uint32_t CLASS_WINDOW = 32;   // ids per class window (one mask word)
constexpr uint32_t SLOTS_PER_BUCKET = 4;
constexpr uint32_t BUCKET_WORDS = 16;
constexpr uint32_t SLOT_WORDS = 4;                 // k <= 32: {key_lo, key_hi, handle, off | ~flags << 24}
constexpr uint32_t SLOT_OFF_MASK = 0xFFFFFFu;
constexpr uint32_t SLOT_FLAG_SHIFT = 24;           // flags 0..2: slot (home + 1 + i) & 3 holds a key of this home; flag 3: one overflowed to the next bucket
constexpr uint32_t SLOT_FLAG_OVERFLOW = 8u;
constexpr double DICT_LOAD = 0.25;                 // k <= 32: keys per slot the builders aim for (see above)
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
    const uint8_t* blobs;     // chain blocks
    const uint32_t* ledge;    // [8 * blocks]: {block handle, y + 1}[4] by chain handle
    const uint64_t* seg_g;    // [num_nodes] 64 * chain handle + node start, ascending (node traces only)
    const uint32_t* seg_nid;  // [num_nodes] node id of seg_g[i]
    const uint32_t* ec;       // class records (16-byte aligned records of u32)
    const uint32_t* class_ref;   // [num_classes]
    const uint32_t* class_len;   // [num_classes]
    const uint32_t* wtable;      // wbuckets * 16 words
    uint32_t wbuckets;
    uint64_t kmask;           // k <= 32: mask of the k-mer word
    uint64_t kmask_hi;        // k > 32: mask of the second k-mer word (bases 32..k-1)
    uint32_t k;
    uint32_t num_nodes, num_classes;
    uint32_t num_segs;        // entries of seg_g / seg_nid (nodes and their copies in tails)
    uint32_t stream_nt;       // 1: the dictionary is larger than the caches — its lines and the read words are loaded non-temporal (lane_steps.hpp, ld_stream)
    uint32_t bitmap_min;      // != 0: every class WITHOUT windows of at least this many ids has a membership bitmap behind its record (class_bitmap)
    uint32_t bitmap_words;    // u32 words of such a bitmap (bit t = transcript t is in the class; two spare words: windows are read as 64 bits)
};

// 16-byte chunks of the record of a class of `len` ids ({class id, ids..., padding}: at least two)
PA_HD uint32_t class_record_chunks(uint32_t len) { return len < 4 ? 2u : (len + 4) >> 2; }
// Where the membership bitmap of the window-less class (ref, len) starts, as a word index into ec (only when ix.bitmap_min != 0 and
// len >= ix.bitmap_min). A class that does not fit two windows of 32 ids — a repeat shared by unrelated genes — can only REMOVE ids from
// a read's running window (lane_steps.hpp, mask_pending); with the bitmap "which ids of [b, b + 32) are in the class" is two loads of
// consecutive words instead of a binary search into the id list and a scan from there (tens of dependent round trips for a read that
// crosses a repeat element: 37 % of the mapping kernel's time on a transcriptome with repeat families).
PA_HD uint64_t class_bitmap(uint32_t ref, uint32_t len) { return 4ull * ((uint64_t)ref + class_record_chunks(len)); }

}  // namespace pa
