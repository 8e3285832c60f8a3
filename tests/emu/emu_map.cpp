// Host emulation of ONE lane of the mapping kernel. TEST INFRASTRUCTURE ONLY.
//
// Compiles the product's per-lane state machine (rust-pseudoaligner_amd/csrc/lane_steps.hpp) and its GPU index
// flattener (device_flatten.cpp) for the host, and runs every read to completion by calling the same step
// functions the HIP kernel calls. It lets the CPU-only test tier (`-m "not gpu"`) check the device data layout and
// the step logic against the oracle bit for bit; the pooled scheduling, LDS staging, arena allocation, the count cache and
// the group (one-list-per-lane / whole-wave) forms of the list intersection exist only in map_pool.hip and are covered
// by the `-m gpu` tier — here list mode runs the per-lane forms (isect_light / isect_scan / binary search). Nothing in the product links or loads this file.
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#include "../../rust-pseudoaligner_amd/csrc/device_flatten.hpp"
#include "../../rust-pseudoaligner_amd/csrc/lane_steps.hpp"
#include "../../rust-pseudoaligner_amd/csrc/pa_common.hpp"

using namespace pa;

struct emu_index {
    FlatDevice fd;
    std::map<std::vector<uint32_t>, uint32_t> by_list;   // id list -> class id: brute-force check of the window table
};

extern "C" {

int emu_index_new(const pa_flat_index* f, int threads, emu_index** out) {
    emu_index* e = new emu_index();
    int rc = flatten_for_device(*f, threads, e->fd);
    if (rc != PA_OK) { delete e; return rc; }
    for (uint32_t c = 0; c < f->num_classes; ++c)
        e->by_list[std::vector<uint32_t>(f->ec_ids + f->ec_offset[c], f->ec_ids + f->ec_offset[c + 1])] = c;
    *out = e;
    return PA_OK;
}
void emu_index_free(emu_index* e) { delete e; }
const char* emu_last_error(void) { return last_error_ref().c_str(); }

uint64_t emu_index_info(const emu_index* e, int what) {
    switch (what) {
        case 0: return e->fd.num_kmers;
        case 1: return e->fd.nbuckets;
        case 2: return e->fd.blobs.size();
        case 4: return e->fd.num_chains;
        case 8: return e->fd.num_bitmaps;   // window-less classes with a membership bitmap behind their record
        case 9: return e->fd.bitmap_min;
        case 5: return e->fd.seg_g.size();
        case 6: {   // chain blocks that break a rule of device_layout.hpp: slots in order record [extension] [edges | link], ends ascending,
                    // the record mask of slot 0 naming exactly the records, edges / link only behind a chain's last record
            uint64_t bad = 0;
            for (size_t b = 0; b + pa::CH_BLOCK <= e->fd.blobs.size(); b += pa::CH_BLOCK) {
                const uint32_t* sl = reinterpret_cast<const uint32_t*>(e->fd.blobs.data() + b);
                const uint32_t recmask = sl[0] >> pa::SEG_RECMASK_SHIFT;
                uint32_t t = 0, seen = 0, prev_e = 0;
                bool ok = (recmask & 1u) != 0, ended = false;
                while (ok && t < pa::CH_SLOTS && ((recmask >> t) & 1u)) {
                    const uint32_t w0 = sl[4 * t], e_rel = w0 & pa::SEG_E_MASK;
                    ok = !ended && e_rel > prev_e;
                    prev_e = e_rel;
                    seen |= 1u << t;
                    t += 1 + ((w0 & pa::SEG_WIDE) ? 1 : 0);
                    // an edge slot follows the chain's last record or a BRANCH record (EDGES without LAST: a copy of the favoured successor follows it); a link only the last
                    if (w0 & (pa::SEG_EDGES | pa::SEG_LINK)) { ok = ok && ((w0 & pa::SEG_LAST) || !(w0 & pa::SEG_LINK)) && !((w0 & pa::SEG_EDGES) && (w0 & pa::SEG_LINK)); ++t; }
                    if (w0 & pa::SEG_LAST) ended = true;
                    ok = ok && t <= pa::CH_SLOTS;
                }
                if (!ok || seen != recmask) ++bad;
            }
            return bad;
        }
        case 7: {   // BRANCH records (layout statistics for the tests)
            uint64_t n = 0;
            for (size_t b = 0; b + pa::CH_BLOCK <= e->fd.blobs.size(); b += pa::CH_BLOCK) {
                const uint32_t* sl = reinterpret_cast<const uint32_t*>(e->fd.blobs.data() + b);
                const uint32_t recmask = sl[0] >> pa::SEG_RECMASK_SHIFT;
                for (uint32_t t = 0; t < pa::CH_SLOTS; ++t)
                    if (((recmask >> t) & 1u) && (sl[4 * t] & (pa::SEG_EDGES | pa::SEG_LAST)) == pa::SEG_EDGES) ++n;
            }
            return n;
        }
        case 3: return e->fd.max_class_len;
        default: return 0;
    }
}

}  // extern "C"

struct emu_index;
namespace pa {
namespace narrow {
#include "emu_loop.inc"
}
namespace wide {
#include "emu_loop.inc"
}
}  // namespace pa

extern "C" {

// results as pa_read_result; class ids as malloc'd CSR in read order; optional step counters [5] = seek, fwd, left steps, spills, reads whose pending
// classes were masked. The lane packing follows the kernel's choice: reads of more than PA_LDS_READ_WORDS (16) words take the WIDE lane state
// (lane_steps.hpp), PA_EMU_WIDE=1 forces it for every batch (the wide text on short reads too)
int emu_map_batch(const emu_index* e, const uint64_t* tiles, uint32_t wpr, const uint32_t* lens, uint64_t n, uint32_t allowed,
                  uint32_t col_cap, pa_read_result* results, uint64_t* class_offsets, uint32_t** class_ids, uint32_t* colour_out,
                  uint64_t* steps, uint32_t* nodes_out, uint32_t nodes_stride, uint32_t* nodes_len) {
    const char* force = getenv("PA_EMU_WIDE");
    if (wpr > 16 || (force && *force == '1'))
        return pa::wide::emu_map_batch_impl(e, tiles, wpr, lens, n, allowed, col_cap, results, class_offsets, class_ids, colour_out, steps, nodes_out, nodes_stride, nodes_len);
    return pa::narrow::emu_map_batch_impl(e, tiles, wpr, lens, n, allowed, col_cap, results, class_offsets, class_ids, colour_out, steps, nodes_out, nodes_stride, nodes_len);
}

void emu_free(void* p) { free(p); }

}  // extern "C"
