// GPU-resident form of the index and the per-lane state of the mapping kernel. Shared by the host flattener
// (device_index.cpp), the HIP kernels (kernels.hip) and the host lane emulator used by the CPU tests (tests/emu).
//
// HBM layout (all little-endian, all read-only after pa_index_create):
//
//   dictionary  bucketed open addressing, 64-byte buckets of four 16-byte slots {key:u64, handle:u32, off:u32};
//               bucket = mulhi64(fmix64(key), nbuckets), linear probing over buckets. A slot carries the k-mer itself,
//               so one 64-byte fetch both finds and VERIFIES a k-mer (the reference needs MPHF levels + a table read +
//               a node-sequence read for the same answer: src/pseudoaligner.rs:96-107). handle == NO_HANDLE = empty.
//   node blobs  one blob per unitig, 32-byte granules, addressed by handle = byte offset / 32:
//                 +0  u32 len (bases)   +4  u32 exts (debruijn::Exts byte)   +8  u32 colour   +12 u32 node id
//                 +16 u32 redge[4]      handle of the node reached by right-extending with base b (Node::r_edges)
//                 +32 u64 seq[ceil(len/32)]  2-bit packed, LSB-first
//               so a node visit is ONE dependent fetch (header + sequence share a line for len <= 128) and the hop to the
//               next node needs no further lookup (the reference re-derives every edge by hashing: SURVEY.md §3.2).
//   ledge       u32[4*num_nodes] handles by node id (Node::l_edges), only touched by the left extension
//   ec_off/ids  CSR of the sorted transcript-id lists (eq_classes: Vec<Vec<u32>>, src/pseudoaligner.rs:29)
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
constexpr uint32_t BLOB_GRANULE = 32;
constexpr uint32_t SLOTS_PER_BUCKET = 4;

struct alignas(16) U4 {
    uint32_t x, y, z, w;
};

struct DevIndexView {
    const U4* table;          // nbuckets * 4 slots
    uint64_t nbuckets;
    const uint8_t* blobs;     // node blobs
    const uint32_t* ledge;    // [4 * num_nodes]
    const uint32_t* ec_off;   // [num_classes + 1]
    const uint32_t* ec_ids;
    uint64_t kmask;
    uint32_t k;
    uint32_t num_nodes, num_classes;
};

}  // namespace pa
