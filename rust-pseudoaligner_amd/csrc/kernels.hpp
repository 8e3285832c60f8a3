// Kernel parameter blocks and launch entry points shared by map_pool.hip, kernels.hip and device_index.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "device_layout.hpp"

namespace pa {

constexpr uint32_t PA_MAP_BLOCK = 256;          // 4 independent waves per workgroup, no barriers
constexpr uint32_t PA_ARENA_CHUNK = 1024;       // u32 entries a wave reserves per global atomic
constexpr uint32_t PA_LDS_READ_WORDS = 16;        // reads of up to 512 bases live in LDS while they are mapped, longer ones stay in their HBM tile
constexpr uint32_t PA_KEY_CHUNK = 1024;         // count keys a wave reserves per global atomic (map_pool.hip, count_sort.hip)
constexpr uint32_t PA_DEFER_CHUNK = 128;        // deferred reads (32-byte entries) a wave reserves per global atomic (map_pool.hip, resolve.hip)
constexpr uint32_t PA_DEFER_WINDOW = 0x80000000u, PA_DEFER_LIST = 0x40000000u;   // kind of a deferred entry (bits 31 / 30 of word 3, the id count below)
constexpr uint32_t PA_KEY_BIN_SHIFT = 15;       // count_sort.hip: a bin = 32768 consecutive count slots = 128 KiB of LDS counters
constexpr uint32_t PA_STATUS_ARENA_FULL = 1u;
constexpr uint32_t PA_STATUS_SPILL_OVERFLOW = 2u;
constexpr unsigned long long PA_NOVEL_LIST_FULL = 4ull;   // OVF_STATUS_LIST_FULL of collective.hip

struct MapParams {
    DevIndexView ix;
    const uint64_t* tiles;
    const uint32_t* lens;      // [n_reads], or nullptr: every read has uniform_len bases (pa_map_count_batch_uniform_device)
    uint32_t uniform_len;
    uint64_t n_reads;
    uint32_t wpr;
    uint32_t allowed;
    pa_read_result* results;
    uint32_t* arena;
    uint64_t arena_cap;
    uint32_t* colour_out;
    unsigned long long* arena_top;
    uint32_t* status;
    uint32_t* tile_ctr;        // tiles handed out beyond every wave's first chunk (zero at launch)
    uint32_t* spill;
    uint32_t spill_cap;
    uint32_t pool_slots;       // read slots per wave (<= 256)
    // optional class-count table (pa_map_count_batch_device): every finished read appends its KEY — the slot of the table it
    // counts in — to its wave's stream in `keys` (chunks of PA_KEY_CHUNK entries handed out through *keys_top, unused tails
    // padded with 0xFFFFFFFF); count_sort.hip turns the streams into the table after the launch. class_table: the class-list
    // hash table the keys of novel subsets are looked up in.
    uint32_t* keys;
    unsigned long long* keys_top;      // entries handed out so far (zero at launch)
    unsigned long long* counts;        // the caller's table (resolve.hip adds the reads of the "novel" slot itself)
    // finished reads whose class still has to be looked up by content (map_pool.hip append_deferred, resolve.hip): 32-byte entries
    // in chunks of PA_DEFER_CHUNK handed out through *defer_top, unused tails padded with rid 0xFFFFFFFF; sized for every read
    uint32_t* defer;
    unsigned long long* defer_top;
    uint64_t keys_cap, defer_cap;      // entries `keys` / `defer` hold: a chunk that would end beyond them is not written (PA_STATUS_SPILL_OVERFLOW)
    const uint32_t* class_table;
    uint64_t class_table_size;
    // optional (with the fused count table): {arena offset, length} of every result that is NO index class goes on this list;
    // pa_overflow_insert_kernel files them by content in the per-GPU overflow table afterwards (collective.hip)
    uint32_t* novel_list;
    unsigned long long* novel_ctr;     // results listed so far (may run past novel_cap: then novel_status gets PA_NOVEL_LIST_FULL)
    unsigned long long* novel_status;
    uint64_t novel_cap;                // pairs
    // optional scheduler statistics (PA_MAP_STATS): [ST_COUNT] iterations, [ST_COUNT] slots served, [ST_COUNT] clock ticks per state
    unsigned long long* dbg;
    uint32_t ablate;           // A/B knob (PA_MAP_ABLATE; 0 in production): 1 = window-mode results are not stored, 2 = no class counts,
                               // 4 = no dual (forward + probe) iterations, 8 = output steps do not refill the slots they free
    // trace launches only (pa_map_read_to_nodes): per-lane scratch, per-read node lists (stride spill_cap) and lengths
    uint32_t* trace;
    uint32_t* nodes_out;
    uint32_t* nodes_len;
};

// the map kernel (map_pool.hip)
size_t pool_slot_bytes(uint32_t wpr);   // LDS bytes per read slot
size_t pool_fixed_bytes();              // LDS bytes per wave besides the slots
uint32_t pool_max_slots();              // slots a wave can schedule
size_t pool_lds_bytes(uint32_t wpr, uint32_t slots);
int launch_map_pool(const MapParams& p, uint32_t grid, size_t lds_bytes, hipStream_t stream);
int pool_kernel_occupancy(size_t lds_bytes, int* blocks_per_cu);
// count_sort.hip: the key streams of a launch -> counts[counts_len] += (keys partitioned by range, counted in LDS). `sorted` holds
// n_reads u32 and `ctl` count_keys_ctl_bytes() of scratch; both may be reused once the stream has passed these kernels.
uint64_t key_stream_capacity(uint64_t n_reads, uint32_t nwaves);
size_t count_keys_ctl_bytes(uint64_t counts_len);                  // bytes of `ctl`   // u32 entries `keys` must hold for a launch of nwaves waves
int launch_count_keys(const uint32_t* keys, const unsigned long long* keys_top, uint64_t keys_cap, const unsigned long long* extra_top, uint64_t extra_cap,
                      uint32_t* sorted, uint32_t* ctl, unsigned long long* counts, uint64_t counts_len, int num_cus, hipStream_t stream, uint64_t n_reads);
// resolve.hip: the deferred content lookups of a launch (records by reference / in the arena, count keys into keys_b, colours, novel list)
uint64_t defer_capacity(uint64_t n_reads, uint32_t nwaves);   // 32-byte entries `defer` must hold
int launch_resolve(const MapParams& p, uint64_t defer_cap, uint64_t keys_cap, int num_cus, hipStream_t stream);
int launch_encode(const uint8_t* ascii, const uint64_t* offsets, uint64_t n, uint32_t wpr, uint64_t* tiles, uint32_t* lens,
                  hipStream_t stream);
int launch_simulate(const uint64_t* packed, const uint64_t* tx_start, const uint64_t* cum, uint32_t num_tx, uint64_t total,
                    uint32_t read_len, uint64_t seed, uint32_t ppm, uint64_t first_read, uint64_t n, uint32_t wpr, uint64_t* tiles,
                    uint32_t* lens, hipStream_t stream);
int launch_count(const pa_read_result* results, const uint32_t* arena, const uint32_t* colour, uint64_t n, const DevIndexView& ix,
                 const uint32_t* class_table, uint64_t class_table_size, unsigned long long* counts, hipStream_t stream);

// the reference's output tuples rendered on the GPU (render.hip). The ids are either back to back with offsets (d_id_off[n + 1], d_rec == nullptr)
// or where they lie in a window's text: d_rec[i] = {id offset, id length, sequence offset, sequence length} into d_ids (fastq_scan.hip).
// d_flagged[PA_RENDER_FLAG_BUCKETS]: reads flagged by the rule of :455 — [0] += those among the first `flag_mark` reads of the batch, [j] += those
// of the j-th million behind them (the progress line of :497-503 is printed with the counts of exactly the first 10^6 m reads).
constexpr uint32_t PA_RENDER_FLAG_BUCKETS = 64;
size_t render_scan_bytes(uint64_t n);
int launch_render_len(const pa_read_result* d_results, const uint32_t* d_arena, const uint8_t* d_ids, const uint64_t* d_id_off, const uint4* d_rec, const uint64_t* d_cls_off,
                      const uint8_t* d_cls_txt, uint64_t n, uint64_t arena_cap, uint64_t flag_mark, uint32_t* d_len, uint64_t* d_off, unsigned long long* d_flagged, void* d_tmp,
                      size_t tmp_bytes, hipStream_t stream);
int launch_render_write(const pa_read_result* d_results, const uint32_t* d_arena, const uint8_t* d_ids, const uint64_t* d_id_off, const uint4* d_rec, const uint64_t* d_cls_off,
                        const uint8_t* d_cls_txt, uint64_t n, uint64_t arena_cap, const uint64_t* d_off, uint8_t* d_text, uint64_t text_cap, hipStream_t stream);

// record finding on the GPU (fastq_scan.hip): what the host learns about a window of FASTQ text
struct FqInfo {
    uint64_t lines;      // line breaks in the window
    uint64_t n;          // whole records (lines / 4)
    uint64_t consumed;   // bytes they take: the next window's first record starts here
    uint32_t max_seq;    // longest record.seq()
    uint32_t odd;        // a first line without '@' or a third without '+': not four-line text (the host scans it with its tolerant rules)
    uint32_t overflow;   // line_start / rec were too small for `lines` / `n`: grow them and scan again
    uint32_t pad;
};
uint32_t fq_chunks(uint64_t begin, uint64_t end);
size_t fq_scan_tmp_bytes(uint32_t n_chunks);
int launch_fq_scan(const uint8_t* d_text, uint64_t begin, uint64_t end, uint32_t* d_chunk, uint32_t* d_first, void* d_tmp, size_t tmp_bytes, uint32_t* d_line_start,
                   uint64_t cap_lines, uint4* d_rec, uint64_t cap_recs, FqInfo* d_info, bool rescan, hipStream_t stream);
int launch_encode_rec(const uint8_t* d_text, const uint4* d_rec, uint64_t n, uint32_t wpr, uint64_t* tiles, uint32_t* lens, hipStream_t stream);

// per-barcode counts (barcode_counts.hip)
int barcode_counts(const DevIndexView& ix, const uint32_t* class_table, uint64_t class_table_size, const pa_read_result* d_results,
                   const uint32_t* d_arena, const uint32_t* d_barcode, uint64_t n, uint32_t barcode_bits, uint64_t* d_keys, uint32_t* d_vals,
                   uint64_t* n_entries, hipStream_t stream);

// overflow table hooks (collective.hip)
void overflow_launch_params(pa_overflow* o, MapParams& p);   // where a full novel list is reported
int overflow_after_map(pa_overflow* o, const uint32_t* novel_list, const unsigned long long* novel_ctr, uint64_t novel_cap, const uint32_t* d_arena,
                       hipStream_t stream);
int overflow_device(const pa_overflow* o);

}  // namespace pa
