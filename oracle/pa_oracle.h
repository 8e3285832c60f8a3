/*
 * pa_oracle.h — CPU restatement of the reference hot path. TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (rust-pseudoaligner_amd/) never links, imports or calls it.
 *
 * Restates, line by line, 10XGenomics/rust-pseudoaligner (debruijn_mapping v0.6.0):
 *   map_read_to_nodes_with_mismatch   src/pseudoaligner.rs:64-319
 *   nodes_to_eq_class                 src/pseudoaligner.rs:323-356
 *   map_read_with_mismatch / map_read src/pseudoaligner.rs:361-384
 *   intersect                         src/pseudoaligner.rs:389-418
 *   output tuple + flag rule          src/pseudoaligner.rs:453-462
 * with constants from src/config.rs:16-18.
 *
 * Third-party pieces the reference calls that are NOT under /root/reference (restated from their published
 * behaviour): debruijn 0.3.4 @ git 8d9a5c52 (DnaString::get/get_kmer, Exts::has_ext/get,
 * Node::l_edges/r_edges = look the terminal k-mer extended by each set base up among node-terminal k-mers) and
 * boomphf 0.6.0 (NoKeyBoomHashMap::get = key-less slot, every hit verified by the caller at :99-107 — restated as a
 * key-less open-addressing table + the same verification, which is observationally an exact dictionary).
 *
 * Parity pinning (the reference is Rust; no rustc/cargo in this image, so it cannot be run here):
 * checked against every known-answer vector the reference's own tests hold for this path — intersect_test's 16
 * vectors (:544-559), the intersect property (:573-586), validate_dbg's properties on test/gencode_small.fa
 * (src/build_index.rs:262-368) and the literals of test_alignment (:429-441) — see tests/test_oracle_*.py.
 */
#ifndef PA_ORACLE_H
#define PA_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_index oracle_index;

typedef struct oracle_counters {   /* feeds the roofline byte model */
    uint64_t reads, mapped;
    uint64_t probes;          /* p: dictionary probes (kmer_lookups, :87,95) */
    uint64_t node_visits;     /* n: nodes.push calls */
    uint64_t bases_compared;  /* c: base comparisons in the two extension loops */
    uint64_t class_sizes;     /* E: sum of |eq_classes[colour(node)]| over visited nodes */
    uint64_t result_sizes;    /* r: sum of |result| */
    uint64_t left_extensions, reseeks;
} oracle_counters;

typedef struct oracle_result {
    uint32_t mapped;       /* 1 = Some, 0 = None */
    uint32_t coverage;
    uint32_t mismatches;
    uint32_t class_len;
} oracle_result;

/* Sequences: 2-bit codes A0 C1 G2 T3, packed LSB-first in u64 words (base j -> bits 2*(j%32) of word j/32). */
oracle_index* oracle_index_new(uint32_t k, uint32_t num_nodes, const uint64_t* node_seq, const uint64_t* node_start,
                               const uint32_t* node_len, const uint8_t* node_exts, const uint32_t* node_colour,
                               uint32_t num_classes, const uint64_t* ec_offset, const uint32_t* ec_ids);
void oracle_index_free(oracle_index* idx);
const char* oracle_last_error(void);

/* fn intersect (src/pseudoaligner.rs:389-418): in place on v1, returns the new length. */
size_t oracle_intersect(uint32_t* v1, size_t n1, const uint32_t* v2, size_t n2);

/* map_read_with_mismatch (:361-376). read = packed words. Returns 1 = Some, 0 = None, <0 = buffer too small.
 * nodes_out (optional) receives the node list in visit order as map_read_to_nodes would fill it (:54-61). */
int oracle_map_read(const oracle_index* idx, const uint64_t* read, uint32_t len, uint32_t allowed_mismatches,
                    uint32_t* class_out, uint32_t class_cap, uint32_t* class_len, uint32_t* coverage,
                    uint32_t* mismatches, uint32_t* nodes_out, uint32_t nodes_cap, uint32_t* num_nodes,
                    oracle_counters* ctr);

/* Batch over `nthreads` pthreads, static contiguous chunks (the body of the worker loop of process_reads, :449-462,
 * without the reader mutex and the println). reads: read i occupies words [i*words_per_read, ...). class ids are
 * returned as a malloc'd CSR in read order (free with oracle_free). */
int oracle_map_batch(const oracle_index* idx, const uint64_t* reads, uint32_t words_per_read, const uint32_t* lens,
                     uint64_t n, uint32_t allowed_mismatches, int nthreads, oracle_result* results,
                     uint64_t* class_offsets, uint32_t** class_ids, oracle_counters* ctr);
/* same, reads in the product's tile layout: tiles[(t*words_per_read + w)*64 + r] */
int oracle_map_batch_tiles(const oracle_index* idx, const uint64_t* tiles, uint32_t words_per_read, const uint32_t* lens,
                           uint64_t n, uint32_t allowed_mismatches, int nthreads, oracle_result* results,
                           uint64_t* class_offsets, uint32_t** class_ids, oracle_counters* ctr);
void oracle_free(void* p);
/* wall time of the mapping threads (create -> join) of the last oracle_map_batch* call: the cpu_baseline figure, which
 * excludes the serial assembly of the CSR output */
double oracle_last_batch_seconds(void);

/* dbg_index.get(kmer) + verification (:96-108): 1 = found. */
int oracle_lookup_kmer(const oracle_index* idx, uint64_t kmer_lo, uint64_t kmer_hi, uint32_t* node, uint32_t* offset);

#ifdef __cplusplus
}
#endif
#endif
