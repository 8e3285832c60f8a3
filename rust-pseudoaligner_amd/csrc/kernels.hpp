// Kernel parameter blocks and launch entry points shared by kernels.hip and device_index.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "device_layout.hpp"

namespace pa {

constexpr uint32_t PA_MAP_BLOCK = 256;          // 4 independent waves per workgroup, no barriers
constexpr uint32_t PA_ARENA_CHUNK = 1024;       // u32 entries a wave reserves per global atomic
constexpr int PA_DEFAULT_MAP_WAVES = 6;
constexpr uint32_t PA_DEFAULT_FAST_STEPS = 4;   // forward steps of the lock-step fast phase         // launch-bounds variant of the map kernel (waves per SIMD)
constexpr uint32_t PA_COUNT_REPLICAS = 8;        // XCDs of an MI355X
constexpr uint32_t PA_STATUS_ARENA_FULL = 1u;
constexpr uint32_t PA_STATUS_SPILL_OVERFLOW = 2u;

struct MapParams {
    DevIndexView ix;
    const uint64_t* tiles;
    const uint32_t* lens;
    uint64_t n_reads;
    uint32_t wpr;
    uint32_t allowed;
    pa_read_result* results;
    uint32_t* arena;
    uint64_t arena_cap;
    uint32_t* colour_out;
    unsigned long long* arena_top;
    uint32_t* status;
    uint32_t* spill;
    uint32_t spill_cap;
    uint32_t pool_slots;       // pooled kernel: read slots per wave (<= 256)
    uint32_t* slow;            // [n_reads] ids of the reads the lock-step fast phase hands to the general state machine
    uint32_t thr_scan, thr_coop, thr_novel, thr_idle;   // scheduler thresholds of the rare finishing states
    uint32_t fast_steps;       // forward steps the fast phase gives a read before handing it over (0 = no fast phase)
    // optional fused class-count table (pa_counts_len entries) and the class-list hash table it needs for novel subsets
    unsigned long long* counts;
    // pooled kernel: the fused counts go to one u32 replica of the table per XCD (PA_COUNT_REPLICAS x xcd_stride entries),
    // which pa_counts_fold_kernel adds into `counts` afterwards: an atomic on a line that only one XCD touches stays in
    // that XCD's L2, while one table shared by the eight L2s moves its lines between them at ~3 G atomics/s
    uint32_t* xcd_counts;
    uint32_t xcd_stride;
    const uint32_t* class_table;
    uint64_t class_table_size;
    // timing experiments only (PA_MAP_ABLATE; results are WRONG when non-zero): 1 = skip the intersection, 2 = skip the
    // forward walk, 4 = skip the dictionary probe
    uint32_t ablate;
    // optional scheduler statistics: [0..4] iterations of refill/seek/fwd/finish/left, [5..9] lanes served by them
    unsigned long long* dbg;
    // trace launches only (pa_map_read_to_nodes): per-lane scratch, per-read node lists (stride spill_cap) and lengths
    uint32_t* trace;
    uint32_t* nodes_out;
    uint32_t* nodes_len;
};

int launch_map(const MapParams& p, uint32_t grid, size_t lds_bytes, int waves, hipStream_t stream);
int map_kernel_occupancy(size_t lds_bytes, int waves, int* blocks_per_cu);
// pooled form of the map kernel (map_pool.hip)
size_t pool_lds_bytes(uint32_t wpr, uint32_t slots);
int launch_map_pool(const MapParams& p, uint32_t grid, size_t lds_bytes, hipStream_t stream);
int pool_kernel_occupancy(size_t lds_bytes, int* blocks_per_cu);
int launch_counts_fold(uint32_t* xcd_counts, uint32_t xcd_stride, unsigned long long* counts, uint64_t len, hipStream_t stream);
int launch_encode(const uint8_t* ascii, const uint64_t* offsets, uint64_t n, uint32_t wpr, uint64_t* tiles, uint32_t* lens,
                  hipStream_t stream);
int launch_simulate(const uint64_t* packed, const uint64_t* tx_start, const uint64_t* cum, uint32_t num_tx, uint64_t total,
                    uint32_t read_len, uint64_t seed, uint32_t ppm, uint64_t first_read, uint64_t n, uint32_t wpr, uint64_t* tiles,
                    uint32_t* lens, hipStream_t stream);
int launch_count(const pa_read_result* results, const uint32_t* arena, const uint32_t* colour, uint64_t n, const DevIndexView& ix,
                 const uint32_t* class_table, uint64_t class_table_size, unsigned long long* counts, hipStream_t stream);

}  // namespace pa
