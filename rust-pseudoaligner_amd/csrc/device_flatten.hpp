// Host-side flattening of a pa_flat_index into the GPU layout of device_layout.hpp (pure host code, no HIP).
#pragma once
#include <vector>

#include "device_layout.hpp"

namespace pa {

struct FlatDevice {
    std::vector<uint32_t> table;   // nbuckets * BUCKET_WORDS
    uint64_t nbuckets = 0;
    std::vector<uint8_t> blobs;
    std::vector<uint32_t> handle;   // node id -> blob handle
    std::vector<uint32_t> ledge;           // by handle
    std::vector<uint32_t> nid_of_handle;
    std::vector<uint32_t> ec;                      // class records (16-byte aligned)
    std::vector<uint32_t> class_ref, class_len;    // by class id
    std::vector<uint32_t> wtable;                  // window classes by content (wbuckets lines of 16 words)
    uint32_t wbuckets = 0;
    uint64_t num_kmers = 0;
    uint32_t k = 0, num_nodes = 0, num_classes = 0, max_class_len = 0;
    // device_dict mode only: k-mers before node i (num_nodes + 1 entries) and which edge directions the flat index supplied
    std::vector<uint64_t> node_kcum;
    bool have_redge = false, have_ledge = false;
    DevIndexView host_view() const;   // pointers into the vectors above
};

// Builds the dictionary, derives the edges (or takes them from the flat index), lays the blobs out and validates
// (duplicate k-mers, dangling extensions). Returns PA_OK or a pa_status (message via pa_last_error).
// device_dict: the dictionary and the edges the flat index does not supply are left to the GPU (index_fill.hip, what
// pa_index_create does): `table` stays empty and `nbuckets` 0, edges not supplied stay NO_HANDLE, node_kcum is filled.
int flatten_for_device(const pa_flat_index& f, int threads, FlatDevice& out, bool device_dict = false);

// index_fill.hip: dictionary fill (CAS into the bucket lines), verification (duplicate k-mers, probe distance) and edge
// derivation on the device, from blobs already resident in HBM. Allocates *d_table (hipMalloc) and sets *nbuckets.
int device_fill_index(const FlatDevice& fd, void* d_blobs, void* d_ledge, void** d_table, uint64_t* nbuckets);

}  // namespace pa
