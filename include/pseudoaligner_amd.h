/*
 * pseudoaligner_amd.h — C ABI of the MI355X-native pseudoalignment hot path.
 *
 * This is the drop-in boundary for ONE path of 10XGenomics/rust-pseudoaligner
 * (crate `debruijn_mapping` v0.6.0): per-read k-mer lookup + De Bruijn graph
 * extension + equivalence-class intersection, i.e.
 *
 *     Pseudoaligner::map_read                 src/pseudoaligner.rs:381-384
 *       -> map_read_with_mismatch             src/pseudoaligner.rs:361-376
 *          -> map_read_to_nodes_with_mismatch src/pseudoaligner.rs:64-319
 *          -> nodes_to_eq_class / intersect   src/pseudoaligner.rs:323-356, 389-418
 *     process_reads (driver)                  src/pseudoaligner.rs:420-514
 *
 * The reference has no FFI of its own (it is a pure-Rust crate); the symbols
 * below are what a Rust `extern "C"` block would bind to replace the body of
 * `process_reads` / `map_read` (see INTEGRATION.md for the binding).
 *
 * Conventions
 *   - plain pointers + sizes, no C++/torch types; every function returns
 *     0 (PA_OK) or a negative pa_status; the message of the last failure on the
 *     calling thread is available from pa_last_error().
 *   - nothing aborts or throws across the ABI (the reference panics instead:
 *     src/pseudoaligner.rs:307,446,464).
 *   - bases are 2-bit codes A=0 C=1 G=2 T=3, packed LSB-first: base j of a
 *     sequence lives in bits [2*(j%32), 2*(j%32)+1] of 64-bit word j/32.
 *   - "device" pointers are HIP device pointers on the GPU the index lives on;
 *     `stream` is a hipStream_t passed as void* (NULL = the null stream).
 */
#ifndef PSEUDOALIGNER_AMD_H
#define PSEUDOALIGNER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PA_ABI_VERSION 1u

/* config.rs:16-18 — the three constants that parameterise the hot path. */
#define PA_READ_COVERAGE_THRESHOLD 32u   /* src/config.rs:16 */
#define PA_LEFT_EXTEND_NUM 1u            /* LEFT_EXTEND_FRACTION = 0.2 = 1/5, src/config.rs:17 */
#define PA_LEFT_EXTEND_DEN 5u
#define PA_DEFAULT_ALLOWED_MISMATCHES 2u /* src/config.rs:18 */
#define PA_SEEK_STRIDE 3u                /* `*kmer_pos += 3`, src/pseudoaligner.rs:110 */

#define PA_NO_EDGE 0xFFFFFFFFu
#define PA_MIN_K 8u
#define PA_MAX_K 64u                     /* one or two 64-bit words per k-mer (Kmer20..Kmer32, Kmer48, Kmer64 of the debruijn crate) */
#define PA_MAX_READ_LEN 1048575u         /* 2^20 - 1 bases. The reference has no limit: its validate_dbg maps whole transcripts (src/build_index.rs:309; the
                                            longest of GENCODE are a few hundred kb). Reads of more than 512 bases stay in their HBM tile while they are
                                            mapped and take the wide lane state (28-bit positions: csrc/lane_steps.hpp); per slot of the pool such a launch
                                            keeps 8 x length words of class-list scratch, so the grid shrinks with the length (a batch of Mb reads is
                                            mapped by a handful of waves: completeness, not throughput) */
#define PA_MAX_SIM_READ_LEN 2048u        /* pa_simulate_reads_*: longest synthetic read */

typedef enum pa_status {
    PA_OK = 0,
    PA_ERR_INVALID_ARG = -1,
    PA_ERR_IO = -2,
    PA_ERR_FORMAT = -3,        /* malformed FASTA/FASTQ or inconsistent flat index */
    PA_ERR_NO_DEVICE = -4,     /* HIP runtime / GPU not usable: the product never falls back to a CPU path */
    PA_ERR_HIP = -5,
    PA_ERR_OOM = -6,
    PA_ERR_ARENA_FULL = -7,    /* caller-provided class arena too small; required size reported */
    PA_ERR_UNSUPPORTED = -8,
    PA_ERR_INTERNAL = -9,
    PA_ERR_BUFFER_TOO_SMALL = -10   /* pa_records_pull: the caller's buffer cannot hold even the next tuple; *n_bytes = the bytes it needs */
} pa_status;

/* ------------------------------------------------------------------------------------------
 * Flat index: the interchange form of `pub struct Pseudoaligner<K>` (src/pseudoaligner.rs:26-33).
 * A Rust exporter fills it from the struct's pub fields (dbg, eq_classes); `dbg_index` (the boomphf
 * MPHF, :30) is NOT part of the interchange because every hit is verified against the node sequence
 * (:99-107), which makes it equivalent to an exact k-mer dictionary that the library rebuilds.
 * All arrays are borrowed for the duration of the call that receives the struct.
 * ------------------------------------------------------------------------------------------ */
typedef struct pa_flat_index {
    uint32_t k;                  /* K::k() */
    uint32_t num_nodes;          /* dbg.len() */
    uint32_t num_classes;        /* eq_classes.len() */
    uint32_t num_transcripts;    /* tx_names.len() */
    uint64_t seq_bases;          /* sum of node_len */
    const uint64_t* node_seq;    /* packed bases of all nodes back to back (node i starts at base node_start[i]) */
    const uint64_t* node_start;  /* [num_nodes + 1] base offsets into node_seq */
    const uint32_t* node_len;    /* [num_nodes] node.len() in bases (>= k) */
    const uint8_t*  node_exts;   /* [num_nodes] debruijn::Exts byte: bit b = right ext b, bit 4+b = left ext b */
    const uint32_t* node_colour; /* [num_nodes] *node.data() = equivalence-class id */
    const uint64_t* ec_offset;   /* [num_classes + 1] CSR offsets into ec_ids */
    const uint32_t* ec_ids;      /* eq_classes[c] = ec_ids[ec_offset[c] .. ec_offset[c+1]) sorted, dedup'd */
    /* optional (may be NULL): edges as node.r_edges()/l_edges() would resolve them, indexed by BASE
     * (not by rank): node_redge[4*i + b] = r_edges()[rank of b among set right exts].0, PA_NO_EDGE when
     * the ext bit is clear. When NULL the library derives them the way the debruijn crate does (look up
     * the terminal k-mer extended by b). */
    const uint32_t* node_redge;  /* [4 * num_nodes] */
    const uint32_t* node_ledge;  /* [4 * num_nodes] */
} pa_flat_index;

/* ---------------- host-side index (CPU; "index construction stays on the CPU") ---------------- */
typedef struct pa_host_index pa_host_index;

/* build_index (src/build_index.rs:27-91) + utils::read_transcripts (src/utils.rs:61-97):
 * FASTA -> stranded coloured compacted De Bruijn graph + equivalence classes. */
int pa_host_index_build_fasta(const char* fasta_path, uint32_t k, int num_threads, pa_host_index** out);
/* same from already packed transcripts: tx_start[num_tx+1] base offsets into `packed`. */
int pa_host_index_build_packed(const uint64_t* packed, const uint64_t* tx_start, uint32_t num_tx,
                               uint32_t k, int num_threads, pa_host_index** out);
/* The same two builders with the graph construction on HIP device `device` (k-mer enumeration, radix sort,
 * colour interning, unitig compaction by pointer jumping — csrc/index_build.hip). The result is the SAME index, array for
 * array, as the CPU builders give (src/build_index.rs:27-91 semantics; numbering as in csrc/dbg_build.cpp).
 * PA_ERR_NO_DEVICE without a GPU; PA_ERR_UNSUPPORTED beyond 2^24 transcripts or 2^32 k-mer occurrences. */
int pa_host_index_build_fasta_device(const char* fasta_path, uint32_t k, int device, pa_host_index** out);
int pa_host_index_build_packed_device(const uint64_t* packed, const uint64_t* tx_start, uint32_t num_tx,
                                      uint32_t k, int device, pa_host_index** out);
/* wrap caller arrays (deep copy) — the import path for an index exported from the Rust side. */
int pa_host_index_from_flat(const pa_flat_index* flat, pa_host_index** out);
int pa_host_index_view(const pa_host_index* h, pa_flat_index* view);   /* pointers valid until destroy */
/* The diff tool of the interchange (an index exported from the Rust side vs the one built here from the same FASTA):
 * 0 = equivalent, 1 = different, 2 = undecided (node sets differ and the k-mer level check would exceed max_kmers),
 * < 0 error. Node / class numbering and node order are free; compared are the nodes as a set of (sequence, extension bits,
 * class id LIST) and, when only unitig break points differ, the k-mer -> id-list map. `report` gets one line of text. */
int pa_host_index_compare(const pa_host_index* a, const pa_host_index* b, uint64_t max_kmers, char* report, size_t report_cap);
int pa_host_index_save(const pa_host_index* h, const char* path);      /* own little-endian container, not bincode */
int pa_host_index_load(const char* path, pa_host_index** out);
/* transcript metadata: tx_names (:31) and the gene of each transcript (:32) */
uint32_t pa_host_index_num_transcripts(const pa_host_index* h);
const char* pa_host_index_tx_name(const pa_host_index* h, uint32_t tx);
const char* pa_host_index_tx_gene(const pa_host_index* h, uint32_t tx);
/* Gene-level collapse of the class-count table through tx_gene_mapping (src/pseudoaligner.rs:32; the reference stores the
 * mapping and leaves the collapse to its callers). Genes are numbered by first appearance in transcript order.
 *   pa_host_index_genes            tx_gene[num_transcripts] (may be NULL) and the number of genes
 *   pa_host_index_gene_name        name of gene g
 *   pa_counts_collapse_genes       gene_counts[g] += class_counts[c] for every class c whose transcripts all belong to gene
 *                                  g; classes that span several genes go to gene_counts[num_genes]; the three tail slots
 *                                  of the class table (novel / empty / unmapped, pa_counts_len) are not gene-resolvable
 *                                  and are skipped. gene_counts has num_genes + 1 entries and is NOT cleared. */
int pa_host_index_genes(const pa_host_index* h, uint32_t* tx_gene, uint32_t* num_genes);
const char* pa_host_index_gene_name(const pa_host_index* h, uint32_t gene);
int pa_counts_collapse_genes(const pa_host_index* h, const uint64_t* class_counts, uint64_t counts_len, uint64_t* gene_counts);
/* Mappability of every transcript (analyze_graph, src/mappability.rs:120-156; `pseudoaligner mappability`,
 * src/bin/pseudoaligner.rs:152-172): a node of L bases holds L - K + 1 k-mers shared by the transcripts of its class.
 *   tx_mult[t * PA_MAPPABILITY_COUNTS_LEN + min(j, LEN) - 1]   k-mers of transcript t whose class has j transcripts
 *   gene_mult[t * PA_MAPPABILITY_COUNTS_LEN + min(g, LEN) - 1] k-mers of transcript t whose class spans g distinct genes
 * Both arrays have num_transcripts * PA_MAPPABILITY_COUNTS_LEN entries and are overwritten; either may be NULL.
 * pa_write_mappability_tsv writes the reference's tx_mappability.tsv (write_mappability_tsv, src/mappability.rs:91-104:
 * header + "tx_name gene_name tx_kmer_count frac_kmer_unique_tx frac_kmer_unique_gene", tab separated, fractions printed
 * like Rust's `{}` of an f64: shortest round-trip digits, fixed notation, "NaN" for a transcript without k-mers). */
#define PA_MAPPABILITY_COUNTS_LEN 11   /* src/config.rs:23 */
int pa_host_index_mappability(const pa_host_index* h, uint64_t* tx_mult, uint64_t* gene_mult);
int pa_write_mappability_tsv(const pa_host_index* h, const char* path);
/* packed transcripts the index was built from (kept for read simulation / validation) */
int pa_host_index_transcripts(const pa_host_index* h, const uint64_t** packed, const uint64_t** tx_start,
                              uint32_t* num_tx);
void pa_host_index_destroy(pa_host_index* h);

/* ---------------- device index ---------------- */
typedef struct pa_index pa_index;

typedef struct pa_index_stats {
    uint64_t num_kmers;        /* distinct k-mers = dictionary entries */
    uint64_t table_slots;      /* dictionary capacity (16-byte slots) */
    uint64_t bytes_table, bytes_graph, bytes_classes, bytes_total;
    uint32_t num_nodes, num_classes, k, max_class_len;
} pa_index_stats;

/* Flatten for the GPU and upload to HIP device `device`. Fails with PA_ERR_NO_DEVICE when no GPU. */
int pa_index_create(const pa_flat_index* flat, int device, pa_index** out);
/* The same for several GPUs of one process (`devices, ndev`): out[i] receives the handle of devices[i]; the
 * index is replicated (reads shard over the handles, §8e). All or nothing: on failure every handle created so far is
 * destroyed and out[] is NULL throughout. A host that runs one process per GPU calls pa_index_create instead. */
int pa_index_create_multi(const pa_flat_index* flat, const int* devices, int ndev, pa_index** out);
int pa_index_get_stats(const pa_index* idx, pa_index_stats* stats);
void pa_index_destroy(pa_index* idx);

/* ---------------- read batches ---------------- */
/* Device tile layout ("coalesced HBM tiles"): reads are grouped 64 to a tile; a tile holds
 * `words_per_read` 64-bit words per read, word-major: tiles[(t*words_per_read + w)*64 + r] is word w of
 * read 64*t + r. lens[i] is the length in bases of read i. n_reads need not be a multiple of 64 but the
 * tile buffer must be sized for ceil(n/64) whole tiles. */
typedef struct pa_read_result {   /* one per read, same order as the input */
    uint32_t coverage;            /* map_read .1 (bases aligned); 0 when unmapped */
    uint32_t mismatches;          /* map_read_with_mismatch .2; bit 31 = mapped (Some vs None) */
    uint32_t class_off;           /* where the class (map_read .0) is: bit 31 set = it IS index class (class_off & 0x7FFFFFFF),
                                     i.e. eq_classes[id] of the flat index, returned by reference; bit 31 clear = offset of
                                     the ids in the arena (u32 units): an intersection that is no single visited class */
    uint32_t class_len;           /* number of transcript ids */
} pa_read_result;
#define PA_MAPPED_BIT 0x80000000u
#define PA_CLASS_REF 0x80000000u
/* Arena offsets must leave bit 31 of class_off free: a launch uses at most this many arena entries, however large the
 * caller's buffer is (a batch that needs more fails with PA_ERR_ARENA_FULL: split it). */
#define PA_MAX_ARENA_ENTRIES 0x7FFFFFFFull

size_t pa_tiles_words(uint64_t n_reads, uint32_t words_per_read);   /* u64 words in the tile buffer */
uint32_t pa_words_per_read(uint32_t max_read_len);

/* DnaString::from_dna_string (src/pseudoaligner.rs:449-450) for a batch, on the GPU:
 * ASCII reads (concatenated, offsets[n+1]) already on the device -> tiles + lens. */
int pa_encode_reads_device(const pa_index* idx, const uint8_t* d_ascii, const uint64_t* d_offsets, uint64_t n_reads,
                           uint32_t words_per_read, uint64_t* d_tiles, uint32_t* d_lens, void* stream);
/* host reference of the same packing (used by the host driver for the single-read path and by tests) */
int pa_encode_reads_host(const uint8_t* ascii, const uint64_t* offsets, uint64_t n_reads, uint32_t words_per_read,
                         uint64_t* tiles, uint32_t* lens);

/* The hot path. map_read_with_mismatch for every read of a device-resident batch.
 *   d_results  [n_reads] pa_read_result
 *   d_arena    [arena_cap] u32: ids of the classes that are not index classes, referenced by (class_off, class_len)
 *   d_colour   optional [n_reads] u32: equivalence-class id of the result when it equals an index class
 *              reached by the read, 0xFFFFFFFF otherwise (input of pa_counts_accumulate_device); may be NULL
 * Asynchronous on `stream`; completion status is fetched with pa_map_finish(idx, stream, ...) (which synchronises that
 * stream). The index is immutable and shareable: launches on DIFFERENT streams — from one host thread or several — run
 * concurrently (every stream gets its own control block, list-mode rows and result streams inside the handle; two launches may
 * accumulate into one d_counts). Launches on ONE stream are ordered by the stream and share a control block: call
 * pa_map_finish between them if you need each launch's own status / arena use. */
int pa_map_batch_device(pa_index* idx, const uint64_t* d_tiles, const uint32_t* d_lens, uint64_t n_reads,
                        uint32_t words_per_read, uint32_t allowed_mismatches, pa_read_result* d_results,
                        uint32_t* d_arena, uint64_t arena_cap, uint32_t* d_colour, void* stream);
/* Same launch with the class-count table fused in: d_counts[pa_counts_len(idx)] (u64, caller-owned so that it can be
 * all-reduced with RCCL) is incremented once per read as pa_counts_accumulate_device would. On PA_ERR_ARENA_FULL every
 * read of that launch is still counted once, but a list-mode result whose ids did not fit cannot be looked up by content
 * and lands in the "novel" slot; the class ids are incomplete. Re-run the batch with the arena pa_map_finish asks for —
 * into a table restored to its value before the failed launch (snapshot it, or count into a scratch table and add it on
 * success): the failed launch has already added its reads. */
int pa_map_count_batch_device(pa_index* idx, const uint64_t* d_tiles, const uint32_t* d_lens, uint64_t n_reads,
                              uint32_t words_per_read, uint32_t allowed_mismatches, pa_read_result* d_results,
                              uint32_t* d_arena, uint64_t arena_cap, uint64_t* d_counts, void* stream);
/* The same launch for a batch whose reads all have `read_len` bases (what a sequencer run delivers): no per-read length array — 4 of
 * the 44 bytes per 150-base read that cross PCIe in the host-to-host pipeline, and one request stream less inside the kernel. */
int pa_map_count_batch_uniform_device(pa_index* idx, const uint64_t* d_tiles, uint32_t read_len, uint64_t n_reads,
                                      uint32_t words_per_read, uint32_t allowed_mismatches, pa_read_result* d_results,
                                      uint32_t* d_arena, uint64_t arena_cap, uint64_t* d_counts, void* stream);
/* Synchronise and report: PA_OK, or PA_ERR_ARENA_FULL with *arena_needed set (re-run with a larger arena).
 * *arena_used = entries of d_arena that may hold ids (never more than the arena_cap of the launch). */
int pa_map_finish(pa_index* idx, void* stream, uint64_t* arena_used, uint64_t* arena_needed);
/* Per-stream scratch: the first launch on a stream creates that stream's launch context inside the handle and keeps it for
 * later launches on the same stream: a control block; list-mode rows of CUs x 3 x 4 waves x 128 slots x (256 x words_per_read
 * + 24) x 4 bytes (about 2 GB at 150 bp on a 256-CU part); the stream of reads whose class is looked up by content after the
 * mapping kernel, sized for the worst case (32 bytes per read of the largest launch); and for class-count launches the key
 * streams (about 13 bytes per read) and, with an overflow table attached, the novel list (8 bytes per read) — some 4.5 GB for
 * 100 M-read launches. pa_index_release_stream frees it (after synchronising the stream): call it before destroying a stream
 * you launched on; pa_index_destroy frees what is left. A host that launches from a pool of N streams holds N contexts. */
int pa_index_release_stream(pa_index* idx, void* stream);
/* Measurement (bench.py's roofline leg): with timing on, every launch records HIP events around its mapping kernel on the launch
 * stream; pa_map_kernel_ms returns the duration of the last launch's mapping kernel on `stream` (it waits for that kernel).
 * The class-count kernels of pa_map_count_batch_device run after the second event. */
int pa_index_set_timing(pa_index* idx, int on);
int pa_map_kernel_ms(pa_index* idx, void* stream, float* ms);
/* The three stages of the last timed launch on `stream`, in ms: ms[0] the mapping kernel, ms[1] the kernel that resolves the
 * deferred content lookups (it writes the records of the reads whose class is looked up by content: part of mapping the
 * batch), ms[2] the class-count kernels (zero-length for launches without a count table). Waits for the launch. */
int pa_map_stage_ms(pa_index* idx, void* stream, float ms[3]);
/* arena capacity (u32 entries) that suffices for typical batches of n_reads; the exact need is data dependent */
uint64_t pa_map_arena_hint(const pa_index* idx, uint64_t n_reads);

/* Compact records for the way back to the host (SURVEY.md §8d measures host to host; the 16-byte records are 40 % of what a 150-base read
 * costs the link in the other direction): what map_read_with_mismatch returns (src/pseudoaligner.rs:361-376) in 8 bytes per read,
 *   bits  0..13  coverage        bits 14..27  mismatches        bit 28  mapped (Some / None)
 *   bit  29      PA_COMPACT_BY_REF: the class IS index class (record >> 32), i.e. eq_classes[id] of the flat index
 *   bit  30      PA_COMPACT_PACKED: the class is no index class: its ids are the next entry of the PACKED stream — entries {length, id0,
 *                id1, ...} back to back in READ order (record >> 32 = the entry's word offset in the packed stream of its launch, modulo 2^32;
 *                a reader that walks the records in order needs none of it)
 *   neither bit: the class is empty (or the read unmapped); both bits: the ids did not fit the launch's arena (pa_map_finish said so)
 * made from a launch's records and arena on the device: d_compact[n_reads] u64, d_packed[packed_cap] u32, *d_packed_words (device u64) =
 * words the packed stream needs (entries that would end beyond packed_cap are not written). d_scratch: pa_compact_scratch_bytes(n_reads)
 * bytes. Asynchronous on `stream`, behind the launch that wrote d_results. */
#define PA_COMPACT_MAPPED 0x10000000u
#define PA_COMPACT_BY_REF 0x20000000u
#define PA_COMPACT_PACKED 0x40000000u
size_t pa_compact_scratch_bytes(uint64_t n_reads);
int pa_results_compact_device(pa_index* idx, const pa_read_result* d_results, const uint32_t* d_arena, uint64_t arena_cap, uint64_t n_reads,
                              uint64_t* d_compact, uint32_t* d_packed, uint64_t packed_cap, uint64_t* d_packed_words, void* d_scratch,
                              size_t scratch_bytes, void* stream);

/* The hot path HOST TO HOST (SURVEY.md §8d's literal metric): a batch that lies in host memory in the tile layout — pinned memory
 * (pa_host_alloc_pinned, hipHostMalloc, hipHostRegister) for the copies to run at the link's rate and beside the kernels — mapped in chunks
 * of chunk_reads reads (0: 1 M) that rotate over n_streams streams of the handle (0: 4; at most 8): the copy of chunk i + 1 to the GPU, the
 * kernels of chunk i and the copy of chunk i - 1's outputs back overlap. h_lens NULL: every read has uniform_len bases (no length array
 * crosses the link). Outputs in read order: h_compact[n_reads] (the 8-byte records above), h_packed[*packed_words] (their packed classes;
 * PA_ERR_ARENA_FULL when packed_cap is too small), h_counts[pa_counts_len(idx)] (the class-count table of the batch, overwritten; may
 * be NULL). Synchronous: everything has arrived when the call returns. Streams and staging buffers stay parked on the handle. */
int pa_map_tiles_host(pa_index* idx, const uint64_t* h_tiles, const uint32_t* h_lens, uint32_t uniform_len, uint64_t n_reads,
                      uint32_t words_per_read, uint32_t allowed_mismatches, uint64_t* h_compact, uint32_t* h_packed, uint64_t packed_cap,
                      uint64_t* packed_words, uint64_t* h_counts, uint64_t chunk_reads, int n_streams);
int pa_host_alloc_pinned(size_t bytes, void** out);
int pa_host_free_pinned(void* p);

/* Host-buffer convenience (H2D, map, D2H; grows its own arena). results[n], class ids returned as a
 * CSR in read order: class_offsets[n+1], class_ids (library-owned, valid until the next call on idx
 * from this thread or pa_index_destroy). Reads are ASCII, concatenated, offsets[n+1]. */
int pa_map_batch(pa_index* idx, const uint8_t* ascii, const uint64_t* offsets, uint64_t n_reads,
                 uint32_t allowed_mismatches, pa_read_result* results, uint64_t* class_offsets,
                 const uint32_t** class_ids);
/* The same for reads the caller already holds 2-bit packed — what `DnaString::from_dna_string` made of the record (:450) and
 * what `map_read(&self, read_seq: &DnaString)` (:381) receives: no ASCII round trip. Read i = lens[i] bases in the words
 * words[word_offsets[i] .. word_offsets[i+1]) (every read starts on a word boundary; bases beyond the length are ignored).
 * layout 0: this library's words (base j in bits 2 (j % 32) of word j / 32); layout 1: MSB-first words (base j in bits
 * 62 - 2 (j % 32)), the storage order of the debruijn crate's DnaString as far as it is known here. */
#define PA_PACKED_LSB_FIRST 0
#define PA_PACKED_MSB_FIRST 1
int pa_map_batch_packed(pa_index* idx, const uint64_t* words, const uint64_t* word_offsets, const uint32_t* lens, uint64_t n_reads,
                        int layout, uint32_t allowed_mismatches, pa_read_result* results, uint64_t* class_offsets,
                        const uint32_t** class_ids);
/* map_read_with_mismatch (:361) / map_read (:381) of one packed read: returns 1 = Some, 0 = None, <0 error. */
int pa_map_read_packed(pa_index* idx, const uint64_t* words, uint32_t len, int layout, uint32_t allowed_mismatches,
                       uint32_t* class_buf, uint32_t class_cap, uint32_t* class_len, uint32_t* coverage, uint32_t* mismatches);
/* map_read (src/pseudoaligner.rs:381): returns 1 = Some, 0 = None, <0 error. */
int pa_map_read(pa_index* idx, const uint8_t* ascii, uint32_t len, uint32_t* class_buf, uint32_t class_cap,
                uint32_t* class_len, uint32_t* coverage);
int pa_map_read_with_mismatch(pa_index* idx, const uint8_t* ascii, uint32_t len, uint32_t allowed_mismatches,
                              uint32_t* class_buf, uint32_t class_cap, uint32_t* class_len, uint32_t* coverage,
                              uint32_t* mismatches);
/* map_read_to_nodes (src/pseudoaligner.rs:54-61, test surface): node ids in visit order. */
int pa_map_read_to_nodes(pa_index* idx, const uint8_t* ascii, uint32_t len, uint32_t allowed_mismatches,
                         uint32_t* node_buf, uint32_t node_cap, uint32_t* num_nodes, uint32_t* coverage,
                         uint32_t* mismatches);

/* node lists of a whole batch (test surface): nodes_flat[i*nodes_stride ..] holds the first min(nodes_len[i],
 * nodes_stride) node ids of read i */
int pa_map_batch_nodes(pa_index* idx, const uint8_t* ascii, const uint64_t* offsets, uint64_t n_reads,
                       uint32_t allowed_mismatches, pa_read_result* results, uint32_t* nodes_flat, uint32_t nodes_stride,
                       uint32_t* nodes_len);

/* process_reads (src/pseudoaligner.rs:420-514): FASTQ in, one Debug-formatted tuple per read on `out_path`
 * ("-" = stdout) in INPUT order (the reference's order is completion order, :490). num_threads sizes the
 * host parse/format pool. n_reads_out/n_flagged_out may be NULL.
 * Input: plain or gzip'ed (multi-member) FASTQ (a gzip'ed file is inflated into host memory first — one zlib stream, some 0.4 GB/s, and the whole
 * text resident: the reference's CLI takes plain files only, src/bin/pseudoaligner.rs:139; inflate large files upstream) in the four-line form every sequencer writes — "@id ...", sequence, "+...",
 * qualities; LF or CRLF; trailing blank lines tolerated (a last record with an empty sequence is still that record). A file whose records are not four lines each (sequence or qualities
 * wrapped over several lines, which bio's reader accepts) is first rewritten into that form by a sequential pass (as many
 * quality lines as sequence lines, as bio 1.5 reads them), then scanned in parallel like any other. PA_ERR_FORMAT with
 * the record number for text that is no FASTQ or ends inside a record. Read ids are cut at the first space, as record.id() does.
 * How it runs (round 6): the host does not look at the text. Worker threads copy WINDOWS of the file into pinned memory (out of the file's
 * mapping with streaming stores; 64 MiB each, PA_INGEST_WINDOW overrides), a window goes to HBM as it is on a copy stream, the GPU finds its records (line breaks, '@' / '+'
 * markers, record.id(), record.seq(): csrc/fastq_scan.hip) and the encode / map / render kernels read sequences and ids where they lie;
 * a window ends where the file offset says, the unfinished record is read again as the head of the next window. The last piece of the
 * text and any text that is not in four-line shape go through the host's tolerant scan (and the same kernels).
 * The pinned host and device buffers of the four windows in flight (about 0.15 GB of each per window for 150-base reads) stay parked
 * on `idx` after a successful call, so that the next file starts with warm buffers; concurrent calls on one index each use their own
 * set; pa_index_destroy frees them. */
int pa_process_reads(pa_index* idx, const char* fastq_path, const char* out_path, int num_threads,
                     uint64_t* n_reads_out, uint64_t* n_flagged_out);
/* process_reads on every GPU it is given — the reference's driver uses every worker it is given (src/pseudoaligner.rs:434-474).
 * idx[0 .. n_idx) are replicas of ONE index (pa_index_create_multi: one handle per GPU of the node); the windows of the text are dealt
 * round-robin to the handles, each with its own streams and buffers, and the tuples are written in INPUT order: the output is byte for
 * byte that of pa_process_reads on one handle. A handle may be listed more than once (it then serves several windows at a time on
 * streams of its own: two lanes on one GPU). PA_ERR_INVALID_ARG when the handles are not replicas (k, nodes, classes, k-mers). */
int pa_process_reads_multi(pa_index* const* idx, int n_idx, const char* fastq_path, const char* out_path, int num_threads,
                           uint64_t* n_reads_out, uint64_t* n_flagged_out);

/* process_reads for a caller that HOLDS the reader. The reference's signature consumes an open fastq::Reader
 * (src/pseudoaligner.rs:420-425), so its drop-in replacement cannot ask for a path: the caller pushes the records it reads —
 * ids as record.id() returns them (:456), sequences as record.seq() (:449), both concatenated with offsets[n+1] — and pulls the
 * reference's Debug tuples (:490), one line per record, in PUSH order. Behind the two calls runs the batch pipeline of
 * pa_process_reads: a full batch (batch_reads, 0 = 2 Mi reads) is packed by `num_threads` workers (0 = all usable CPUs) and
 * launched on the stream's own HIP stream while the caller goes on reading; the batch before it is rendered meanwhile.
 *   push   copies the records (the caller's buffers are free afterwards); may pack + launch a batch and render the previous one
 *   pull   copies rendered text into buf, whole lines only, never waits for the GPU; *n_bytes = 0: nothing ready yet. A buffer that
 *          cannot hold even the next tuple gets PA_ERR_BUFFER_TOO_SMALL with *n_bytes = the bytes that tuple needs (nothing is lost:
 *          pull again with a larger buffer; this failure is not sticky)
 *   flush  launches what is left, waits and renders: afterwards pull drains every record pushed so far
 * One thread at a time per stream object; several objects may share an index. A failure is sticky: every later call on the
 * object returns it. */
typedef struct pa_record_stream pa_record_stream;
int pa_record_stream_create(pa_index* idx, int num_threads, uint64_t batch_reads, pa_record_stream** out);
/* The same over several handles of one index (pa_index_create_multi: the GPUs of a node; a handle may be listed twice): full batches go round-robin to the
 * handles, each on its own stream with its own buffers, and the tuples come back in PUSH order — byte for byte what one handle gives. (The reader-fed form is
 * bound by its reader long before one GPU is: INTEGRATION.md §1; this entry is for callers that parse FASTQ on many threads of their own.) */
int pa_record_stream_create_multi(pa_index* const* idx, int n_idx, int num_threads, uint64_t batch_reads, pa_record_stream** out);
int pa_records_push(pa_record_stream* s, const uint8_t* ids, const uint64_t* id_offsets, const uint8_t* seqs,
                    const uint64_t* seq_offsets, uint64_t n_records);
int pa_records_pull(pa_record_stream* s, char* buf, size_t cap, size_t* n_bytes);
int pa_records_flush(pa_record_stream* s);
/* records rendered so far and how many of them carry the flag of :455 (coverage >= 32 and an empty class) */
int pa_record_stream_stats(const pa_record_stream* s, uint64_t* n_reads, uint64_t* n_flagged);
void pa_record_stream_destroy(pa_record_stream* s);

/* Measurement (bench.py's ingest leg): wall seconds the HOST stages of the last pa_process_reads[_multi] call of this thread /
 * of a record stream since its creation took — the stages run one after the other on the caller's thread, each spread over
 * the worker pool, while the GPU works on the windows before: out[0] scan (record boundaries found by the HOST: the end of the text,
 * text that is not in four-line shape; 0 for a record stream), out[1] pack (pa_process_reads: reading the windows' text into pinned
 * memory; a record stream: gathering ids and sequences), out[2] waiting for the GPU (a window's scan, its kernels), out[3] launch,
 * out[4] waiting for the rendered tuples, out[5] waiting for the writer (0 for a record stream), out[6] the whole call (streams: the
 * sum of the others), out[7] reads. */
#define PA_INGEST_STAGES 8
int pa_process_reads_stage_seconds(double out[PA_INGEST_STAGES]);
int pa_record_stream_stage_seconds(const pa_record_stream* s, double out[PA_INGEST_STAGES]);

/* The scan stage of pa_process_reads by itself, without a GPU: the number of records of a FASTQ file (plain, gzip'ed or with
 * wrapped lines: same acceptance rules and errors as above) and, for the first `capacity` of them, where the record starts,
 * how many bytes its header line has before the line feed ('@' included, and the CR of a CRLF file) and how many bases its
 * sequence has. The sequence begins at start + header_len + 1. Offsets refer to the text as scanned: *text_kind = 0 the file itself, 1 the inflated gzip stream, 2 the text
 * rewritten into four-line records (a file that is in four-line shape at first and wrapped further on is rewritten from the first scan
 * window that is not in shape: offsets of the records from there on refer to the rewritten rest). Every output pointer but n_records may be NULL. */
int pa_fastq_scan_host(const char* fastq_path, int num_threads, uint64_t* n_records, uint64_t* starts, uint32_t* header_len,
                       uint32_t* seq_len, uint64_t capacity, int* text_kind);

/* ---------------- equivalence-class count table (multi-GPU reduction unit) ---------------- */
/* counts[c] += number of reads whose class equals index class c; reads with a novel (non-index)
 * non-empty class are counted in counts[num_classes] ("novel"), empty-class mapped reads in
 * counts[num_classes+1], unmapped reads in counts[num_classes+2]. d_counts is a caller-owned device
 * array of pa_counts_len(idx) u64 (so that the caller can all-reduce it with RCCL). */
uint64_t pa_counts_len(const pa_index* idx);
int pa_counts_accumulate_device(pa_index* idx, const pa_read_result* d_results, const uint32_t* d_arena,
                                const uint32_t* d_colour, uint64_t n_reads, uint64_t* d_counts, void* stream);

/* Per-barcode (single-cell) counts, SURVEY.md §8f.3 (the reference's stated purpose, README.md:3): d_barcode[i] = index of the
 * cell barcode of read i (assigned by the host from the barcode read / whitelist). Output = the sparse matrix
 * (barcode, column) -> reads as sorted unique keys (barcode << 32 | column) with their counts, columns as in the dense
 * table (class id, pa_counts_len-3.. = novel / empty / unmapped). d_keys / d_vals are caller-owned device arrays of n_reads
 * entries (the worst case); *n_entries (host) receives the number of non-zero cells. barcode_bits: bits of the barcode
 * index that can be set (0 = all 32; fewer bits = fewer sort passes). Synchronous on `stream`. */
int pa_counts_by_barcode_device(pa_index* idx, const pa_read_result* d_results, const uint32_t* d_arena,
                                const uint32_t* d_barcode, uint64_t n_reads, uint32_t barcode_bits, uint64_t* d_keys,
                                uint32_t* d_vals, uint64_t* n_entries, void* stream);

/* ---------------- novel classes + the reduction over GPUs ---------------- */
/* The dense table counts every result that is no index class in ONE slot (counts[num_classes]). A pa_overflow keeps WHICH
 * id sets those were, keyed by content, on the GPU: attach one to an index and every pa_map_count_batch_device launch files
 * its novel results there (a small follow-up kernel on the same stream). The reduction unit over GPUs is then
 *   dense table      -> pa_counts_allreduce  (RCCL all-reduce, sum of u64[pa_counts_len])
 *   overflow tables  -> pa_overflow_allgather (RCCL all-gather of the serialised tables, merged by content on every rank)
 * and sum(counts of the merged overflow records) == counts[num_classes] of the reduced dense table.
 * Serialised form (u32 words): [0] = records, [1] = words used (this header included), then per record
 * {len, count low, count high, ids[len]}; merged tables are ordered lexicographically by id list (canonical).
 * max_classes / max_ids size the table (distinct novel classes, sum of their lengths); running out is reported by
 * pa_overflow_fetch / pa_overflow_allgather as PA_ERR_ARENA_FULL, never silently. */
typedef struct pa_overflow pa_overflow;
int pa_overflow_create(int device, uint64_t max_classes, uint64_t max_ids, pa_overflow** out);
void pa_overflow_destroy(pa_overflow* ovf);
int pa_overflow_reset(pa_overflow* ovf, void* stream);
int pa_index_set_overflow(pa_index* idx, pa_overflow* ovf);   /* NULL detaches; the table must outlive its attachment */
/* this GPU's table, serialised and canonical, on the host (library-owned until the next call on ovf) */
int pa_overflow_fetch(pa_overflow* ovf, void* stream, const uint32_t** words, uint64_t* n_words);
/* host-side merge of serialised tables (what every rank does with the gathered buffers); out may be NULL to size */
int pa_overflow_merge(const uint32_t* const* bufs, const uint64_t* n_words, int nbufs, uint32_t* out, uint64_t out_cap,
                      uint64_t* out_words);

/* RCCL communicator, one rank per GPU (xGMI). The 128-byte id comes from ONE rank (pa_comm_unique_id) and reaches the
 * others through whatever the host already has (MPI, a file, torch.distributed ...). RCCL is bound at run time: without
 * librccl these calls fail with PA_ERR_UNSUPPORTED, everything else works. comm == NULL means "one GPU": the reduce is a
 * no-op and the gather is pa_overflow_fetch. */
typedef struct pa_comm pa_comm;
int pa_comm_unique_id(uint8_t id[128]);
int pa_comm_create(int device, int nranks, int rank, const uint8_t id[128], pa_comm** out);
void pa_comm_destroy(pa_comm* comm);
int pa_comm_rank(const pa_comm* comm);
int pa_comm_size(const pa_comm* comm);
int pa_counts_allreduce(pa_index* idx, uint64_t* d_counts, pa_comm* comm, void* stream);   /* in place, asynchronous on stream */
int pa_overflow_allgather(pa_overflow* ovf, pa_comm* comm, void* stream, const uint32_t** words, uint64_t* n_words);

/* ---------------- synthetic workloads (BASELINE.json configs; deterministic, counter-based) ------------- */
/* GENCODE-like transcriptome: returns a host index-less transcript set. */
typedef struct pa_txome pa_txome;
int pa_txome_synthesize(uint32_t num_genes, uint32_t target_transcripts, uint64_t seed, pa_txome** out);
/* The same transcriptome (same genes, exons and isoforms for the same seed) with REAL-GRAPH structure added (bench.py's workload
 * "config3r"): interspersed repeats — `families` + `young_families` consensus elements of `element_len` random bases; a gene is hit with
 * probability gene_fraction_ppm / 1e6 and then carries ONE copy of a random family in its last exon ("3' UTR": every isoform that keeps
 * that exon has it), every base of the copy substituted with a probability drawn per copy from [div_lo_ppm, div_hi_ppm] (old families,
 * Alu-like) or [young_div_lo_ppm, young_div_hi_ppm] — and `low_complexity_genes` genes whose last exon ends in a poly-A, (CA)n or (CAG)n
 * tract of 30..89 units. Result: k-mers shared by tens to hundreds of transcripts of unrelated genes (classes far beyond two 32-id
 * windows), branch-dense unitigs inside the elements, self-loops in the tracts. */
typedef struct pa_synth_repeats {
    uint32_t families, element_len, div_lo_ppm, div_hi_ppm;
    uint32_t young_families, young_div_lo_ppm, young_div_hi_ppm;
    uint32_t gene_fraction_ppm, low_complexity_genes;
} pa_synth_repeats;
int pa_txome_synthesize_repeats(uint32_t num_genes, uint32_t target_transcripts, uint64_t seed, const pa_synth_repeats* repeats, pa_txome** out);
int pa_txome_from_host_index(const pa_host_index* h, pa_txome** out);
int pa_txome_from_fasta(const char* fasta_path, pa_txome** out);
int pa_txome_view(const pa_txome* t, const uint64_t** packed, const uint64_t** tx_start, uint32_t* num_tx);
void pa_txome_destroy(pa_txome* t);
/* reads: read i = transcript drawn with probability proportional to (len - read_len + 1), uniform start,
 * forward strand, per-base substitution with probability sub_rate_ppm/1e6; function of (seed, first_read+i) only. */
int pa_simulate_reads_host(const pa_txome* t, uint32_t read_len, uint64_t seed, uint32_t sub_rate_ppm,
                           uint64_t first_read, uint64_t n_reads, uint32_t words_per_read, uint64_t* tiles,
                           uint32_t* lens);
typedef struct pa_txome_device pa_txome_device;
int pa_txome_upload(const pa_txome* t, uint32_t read_len, int device, pa_txome_device** out);
void pa_txome_device_destroy(pa_txome_device* t);
int pa_simulate_reads_device(const pa_txome_device* t, uint64_t seed, uint32_t sub_rate_ppm, uint64_t first_read,
                             uint64_t n_reads, uint32_t words_per_read, uint64_t* d_tiles, uint32_t* d_lens,
                             void* stream);

/* ---------------- misc ---------------- */
uint32_t pa_abi_version(void);
int pa_device_count(void);
const char* pa_last_error(void);
/* HIP-event timing on the stream kernels are launched on (bench.py's roofline leg). */
int pa_event_create(void** ev);
int pa_event_record(void* ev, void* stream);
int pa_event_elapsed_ms(void* start, void* stop, float* ms);   /* synchronises `stop` */
int pa_event_destroy(void* ev);
/* raw device memory for hosts without their own allocator */
int pa_device_malloc(int device, size_t bytes, void** out);
int pa_device_free(void* p);
int pa_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream);
int pa_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream);
int pa_memset_device(void* dst, int value, size_t bytes, void* stream);
int pa_stream_synchronize(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PSEUDOALIGNER_AMD_H */
