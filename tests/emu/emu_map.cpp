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

// results as pa_read_result; class ids as malloc'd CSR in read order; optional step counters [5] = seek, fwd, left steps, spills, reads whose pending classes were masked
int emu_map_batch(const emu_index* e, const uint64_t* tiles, uint32_t wpr, const uint32_t* lens, uint64_t n, uint32_t allowed,
                  uint32_t col_cap, pa_read_result* results, uint64_t* class_offsets, uint32_t** class_ids, uint32_t* colour_out,
                  uint64_t* steps, uint32_t* nodes_out, uint32_t nodes_stride, uint32_t* nodes_len) {
    const DevIndexView ix = e->fd.host_view();
    std::vector<uint32_t> all;
    std::vector<uint64_t> rd(wpr + 2);
    alignas(16) uint32_t refs[4], lens4[4], cids4[4], win4[4];
    uint32_t wcand[2];
    std::vector<uint32_t> spill, trace, pend;
    (void)col_cap;
    uint64_t st_seek = 0, st_fwd = 0, st_left = 0, st_spill = 0, st_mask = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t t = i >> 6, r = i & 63;
        for (uint32_t w = 0; w < wpr; ++w) rd[w] = tiles[(t * wpr + w) * 64 + r];
        rd[wpr] = rd[wpr + 1] = 0;
        const uint32_t L = lens[i];
        spill.assign(8 * (size_t)L + 8, 0);
        trace.assign(8 * (size_t)L + 8, 0);
        pend.assign(8 * (size_t)L + 8, 0);
        Lane s;
        lane_start(s, (uint32_t)i, L, ix.k);
        const ReadRef rr{rd.data(), 1, wpr};
        const ColRef cr{win4, wcand, refs, lens4, cids4, spill.data(), (uint32_t)spill.size(), pend.data(), trace.data()};
        for (;;) {
            while (l_st(s) == ST_SEEK || l_st(s) == ST_FWD || l_st(s) == ST_LEFT) {
                if (l_st(s) == ST_SEEK) { seek_step(s, ix, rr); ++st_seek; }
                else if (l_st(s) == ST_FWD) {   // a traced batch takes the general form of the step, any other the kernel's common text
                    if (nodes_out) fwd_step<true>(s, ix, rr, cr, allowed); else fwd_step<false>(s, ix, rr, cr, allowed);
                    ++st_fwd;
                } else {
                    if (nodes_out) left_step<true>(s, ix, rr, cr, allowed); else left_step<false>(s, ix, rr, cr, allowed);
                    ++st_left;
                }
            }
            if (l_st(s) != ST_ISECT || (l_flags(s) & F_LISTS)) break;
            const uint32_t todo = window_todo(s);   // window mode: classes without windows still to apply?
            if (todo == 2) { restart_lists(s, ix.k); continue; }
            if (todo == 1) { mask_pending(s, ix, cr); ++st_mask; }
            break;
        }
        if (l_flags(s) & F_SPILL_OVERFLOW) return PA_ERR_INTERNAL;
        if ((l_flags(s) & F_LISTS) && l_ncol(s) > LDS_CLASSES) ++st_spill;
        if (nodes_out) {
            nodes_len[i] = l_ntrace(s);
            for (uint32_t j = 0; j < l_ntrace(s) && j < nodes_stride; ++j) nodes_out[i * nodes_stride + j] = trace[j];
        }
        class_offsets[i] = all.size();
        pa_read_result res{0, 0, 0, 0};
        uint32_t colour = 0xFFFFFFFFu;
        if (l_st(s) == ST_ISECT && !(l_flags(s) & F_LISTS)) {   // window mode: {base1, mask1, base2, mask2} + class id or NO_CLASS
            const uint32_t cand = wcand[0];
            const uint32_t count = (uint32_t)(__builtin_popcount(win4[1]) + __builtin_popcount(win4[3]));
            const size_t o = all.size();
            all.resize(o + count);
            uint32_t j = 0;
            for (uint32_t t = win4[1]; t; t &= t - 1) all[o + j++] = win4[0] + (uint32_t)__builtin_ctz(t);
            for (uint32_t t = win4[3]; t; t &= t - 1) all[o + j++] = win4[2] + (uint32_t)__builtin_ctz(t);
            res.coverage = l_cov(s);
            res.mismatches = l_mism(s) | PA_MAPPED_BIT;
            res.class_len = count;
            res.class_off = (uint32_t)o;
            const std::vector<uint32_t> ids(all.begin() + o, all.end());
            const auto it = e->by_list.find(ids);
            const uint32_t truth = it == e->by_list.end() ? NO_CLASS : it->second;
            if (cand != NO_CLASS) {   // returned by reference: must be exactly that index class
                colour = cand;
                res.class_off = PA_CLASS_REF | colour;
                if (colour != truth) return PA_ERR_INTERNAL;
            } else if (count) {       // strict subset of every class seen: the window table says whether it is a class anyway
                uint32_t b1 = win4[0], m1 = win4[1], b2 = win4[2], m2 = win4[3];
                window_canon(b1, m1, b2, m2);
                colour = window_class(ix, b1, m1, b2, m2);
                if (colour != truth) return PA_ERR_INTERNAL;
            }
        } else if (l_st(s) == ST_ISECT) {
            const Isect is = isect_count(s, ix, cr);
            const size_t o = all.size();
            all.resize(o + is.count);
            res.coverage = l_cov(s);
            res.mismatches = l_mism(s) | PA_MAPPED_BIT;
            res.class_len = is.count;
            if (is.count == is.base_len) {   // the class is index class base_colour: returned by reference; the CSR the
                colour = is.base_colour;     // tests compare is resolved from the class table like a host would
                res.class_off = PA_CLASS_REF | colour;
                const uint32_t* cls = pa::class_ids(ix, ix.class_ref[colour]);
                for (uint32_t j = 0; j < is.count; ++j) all[o + j] = cls[j];
                if (ix.class_len[colour] != is.count || ix.class_ref[colour] != is.base_ref) return PA_ERR_INTERNAL;
            } else {
                isect_write(s, ix, cr, is, all.data() + o);
                res.class_off = (uint32_t)o;
            }
        }
        results[i] = res;
        if (colour_out) colour_out[i] = colour;
    }
    class_offsets[n] = all.size();
    uint32_t* p = (uint32_t*)malloc((all.size() ? all.size() : 1) * 4);
    if (!p) return PA_ERR_OOM;
    memcpy(p, all.data(), all.size() * 4);
    *class_ids = p;
    if (steps) { steps[0] = st_seek; steps[1] = st_fwd; steps[2] = st_left; steps[3] = st_spill; steps[4] = st_mask; }
    return PA_OK;
}

void emu_free(void* p) { free(p); }

}  // extern "C"
