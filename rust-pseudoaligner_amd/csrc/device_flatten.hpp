// Host-side flattening of a pa_flat_index into the GPU layout of device_layout.hpp (pure host code, no HIP).
#pragma once
#include <vector>

#include "device_layout.hpp"

namespace pa {

struct FlatDevice {
    std::vector<uint32_t> table;   // nbuckets * BUCKET_WORDS
    uint64_t nbuckets = 0;
    std::vector<uint8_t> blobs;     // chain blocks
    std::vector<uint32_t> handle;   // node id -> handle of its chain
    std::vector<uint32_t> node_s;   // node id -> position of its first base in the chain
    std::vector<uint32_t> ledge;    // by chain handle
    std::vector<uint64_t> seg_g;    // node traces: 64 * chain handle + node_s, ascending ...
    std::vector<uint32_t> seg_nid;  // ... and the node there
    uint32_t num_chains = 0;
    std::vector<uint32_t> ec;                      // class records (16-byte aligned)
    std::vector<uint32_t> class_ref, class_len;    // by class id
    std::vector<uint32_t> wtable;                  // window classes by content (wbuckets lines of 16 words)
    uint32_t wbuckets = 0;
    uint64_t num_kmers = 0;
    uint32_t k = 0, num_nodes = 0, num_classes = 0, max_class_len = 0;
    uint32_t bitmap_min = 0, bitmap_words = 0;     // membership bitmaps of the window-less classes (device_layout.hpp, class_bitmap)
    uint64_t num_bitmaps = 0;
    // device_dict mode only: k-mers before node i (num_nodes + 1 entries)
    std::vector<uint64_t> node_kcum;
    DevIndexView host_view() const;   // pointers into the vectors above
};

// Derives the edges (or takes them from the flat index), merges nodes into chains, lays the chain blocks out, builds the
// dictionary and validates (duplicate k-mers, dangling extensions). Returns PA_OK or a pa_status (message via pa_last_error).
// device_dict: the dictionary is left to the GPU (index_fill.hip, what pa_index_create does): `table` stays empty and
// `nbuckets` 0, node_kcum is filled.
int flatten_for_device(const pa_flat_index& f, int threads, FlatDevice& out, bool device_dict = false);

// index_fill.hip: dictionary fill (CAS into the bucket lines) and verification (duplicate k-mers, probe distance) on the
// device, from chain blocks already resident in HBM. Allocates *d_table (hipMalloc) and sets *nbuckets.
int device_fill_index(const FlatDevice& fd, void* d_blobs, void** d_table, uint64_t* nbuckets);

}  // namespace pa
