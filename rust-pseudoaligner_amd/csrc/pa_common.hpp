// Shared host-side definitions: error plumbing, 2-bit sequence helpers, the k-mer hash, the host index.
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pseudoaligner_amd.h"

namespace pa {

// ---- thread-local error message (pa_last_error) ----
std::string& last_error_ref();
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

// host copy of the class records of a device index (device_index.hip): class c = ec[4 * class_ref[c] + 1 ...]; used to
// resolve results returned by reference (PA_CLASS_REF) without a device round trip
void index_host_classes(const pa_index* idx, const uint32_t** ec, const uint32_t** class_ref, int* device);
// every index class rendered once as the reference prints its ids ("1, 5, 9", no brackets): class c = text[off[c] .. off[c + 1]). Built on first use.
void index_host_class_text(pa_index* idx, const uint64_t** off, const char** text);
// ... and its copy in HBM for the render kernels (render.hip); uploaded on first use. Returns a pa_status
int index_device_class_text(pa_index* idx, const uint64_t** d_off, const uint8_t** d_text);
// One opaque object the FASTQ driver parks on the index between calls (its pinned + device batch buffers: allocating them
// costs more than packing a batch). take() hands it to the caller and empties the slot, so concurrent calls never share
// it; put() stores it back (or frees it with `free_fn` when another call already parked one). pa_index_destroy frees it.
void* index_take_ingest_cache(pa_index* idx);
void index_put_ingest_cache(pa_index* idx, void* cache, void (*free_fn)(void*));
// the same for the streams and staging buffers of pa_map_tiles_host (host_batch.cpp)
void* index_take_host_pipe(pa_index* idx);
void index_put_host_pipe(pa_index* idx, void* pipe, void (*free_fn)(void*));

// A/B tuning knobs (PA_MAP_ABLATE, PA_MAP_STATS, PA_POOL_SLOTS, PA_MAP_BLOCKS_PER_CU, PA_DICT_LOAD, PA_SIM_TX_LIMIT) exist only
// in builds made with -DPA_DEBUG_KNOBS (tools/build_variant.sh). The shipped library never reads them: an exported variable
// cannot make it skip result stores or counts.
static inline const char* knob_str(const char* name) {
#ifdef PA_DEBUG_KNOBS
    const char* v = getenv(name);
    return v && *v ? v : nullptr;
#else
    (void)name;
    return nullptr;
#endif
}
static inline int knob_int(const char* name, int dflt) {
    const char* v = knob_str(name);
    return v ? atoi(v) : dflt;
}

// CPUs this process may use: hardware threads, capped by the cgroup CPU quota (a container that sees 256 CPUs may be
// limited to 16 CPUs' worth of time; 256 threads there only add scheduling noise). host_index.cpp
int usable_threads();

// ---- 2-bit packed sequences, LSB-first (base j -> bits 2*(j%32) of word j/32) ----
static inline uint64_t kmer_mask(uint32_t k) { return k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1); }

static inline uint32_t get_base(const uint64_t* w, uint64_t pos) { return (uint32_t)(w[pos >> 5] >> ((pos & 31) * 2)) & 3u; }

static inline void set_base(uint64_t* w, uint64_t pos, uint32_t b) {
    const uint32_t s = (uint32_t)(pos & 31) * 2;
    w[pos >> 5] = (w[pos >> 5] & ~(3ull << s)) | ((uint64_t)(b & 3) << s);
}

// 32 bases starting at base `pos` (caller guarantees word (pos>>5)+1 is readable)
static inline uint64_t window32(const uint64_t* w, uint64_t pos) {
    const uint64_t i = pos >> 5;
    const uint32_t s = (uint32_t)(pos & 31) * 2;
    return s ? (w[i] >> s) | (w[i + 1] << (64 - s)) : w[i];
}

static inline uint64_t get_kmer(const uint64_t* w, uint64_t pos, uint32_t k) { return window32(w, pos) & kmer_mask(k); }

// ASCII -> 2-bit code; 4 = not ACGT (either case)
static inline uint32_t base_code(uint8_t c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 4;
    }
}

// The dictionary hash (shared by the host table builder and the HIP kernel): murmur3 fmix64.
static inline uint64_t mix64(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

// ---- k-mers of up to 32 bases (one word) or up to 64 bases (two words: Kmer64 of the debruijn crate, the other k the
// reference's CLI accepts, src/bin/pseudoaligner.rs:88) ----
typedef unsigned __int128 u128;

template <class KT> struct KmerOps;
template <> struct KmerOps<uint64_t> {
    static uint64_t mask(uint32_t k) { return kmer_mask(k); }
    static uint64_t get(const uint64_t* w, uint64_t pos, uint32_t k) { return get_kmer(w, pos, k); }   // words (pos>>5)+1 readable
    static uint64_t hash(uint64_t km) { return mix64(km); }
};
template <> struct KmerOps<u128> {   // 32 < k <= 64
    static u128 mask(uint32_t k) { return k >= 64 ? ~(u128)0 : (((u128)1 << (2 * k)) - 1); }
    static u128 get(const uint64_t* w, uint64_t pos, uint32_t k) {   // words (pos>>5)+2 readable
        return ((u128)(window32(w, pos + 32) & kmer_mask(k - 32)) << 64) | window32(w, pos);
    }
    static uint64_t hash(u128 km) { return mix64((uint64_t)km ^ (mix64((uint64_t)(km >> 64)) * 0x9e3779b97f4a7c15ull)); }
};

static inline uint64_t splitmix64(uint64_t& s) {
    uint64_t z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

// ---- host index: the flat form of Pseudoaligner<K> (src/pseudoaligner.rs:26-33) ----
struct HostIndex {
    uint32_t k = 0;
    uint32_t num_transcripts = 0;
    // graph
    std::vector<uint64_t> node_seq;     // packed, +1 pad word
    std::vector<uint64_t> node_start;   // n+1
    std::vector<uint32_t> node_len;
    std::vector<uint8_t> node_exts;
    std::vector<uint32_t> node_colour;
    std::vector<uint32_t> node_redge, node_ledge;   // optional (empty = derive)
    // classes
    std::vector<uint64_t> ec_offset;    // c+1
    std::vector<uint32_t> ec_ids;
    // transcript metadata + the packed transcripts (tx_names :31, tx_gene_mapping :32)
    std::vector<std::string> tx_names, tx_genes;
    std::vector<uint64_t> tx_packed;    // +1 pad word
    std::vector<uint64_t> tx_start;     // num_transcripts+1
};

struct Txome {
    std::vector<uint64_t> packed;       // +1 pad word
    std::vector<uint64_t> tx_start;     // n+1
    std::vector<std::string> names, genes;
    uint32_t num_tx() const { return tx_start.empty() ? 0 : (uint32_t)(tx_start.size() - 1); }
};

// dbg_build.cpp
int build_graph(const uint64_t* packed, const uint64_t* tx_start, uint32_t num_tx, uint32_t k, int threads, HostIndex& out);
// index_build.hip: the same graph built on HIP device `device`
int build_graph_device(const uint64_t* packed, const uint64_t* tx_start, uint32_t num_tx, uint32_t k, int device, HostIndex& out);
// fasta.cpp
int read_fasta(const char* path, Txome& out);

}  // namespace pa

struct pa_host_index { pa::HostIndex h; };
struct pa_txome { pa::Txome t; };
