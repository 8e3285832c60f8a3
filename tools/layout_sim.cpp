// Tuning aid (CPU only): how many 128-byte blocks of the node blobs does a read touch under different blob placements?
// Runs the product's lane state machine on the host (as tests/emu does) over simulated reads of a workload, records for
// every forward / left step which bytes of which node it loads (the address arithmetic of fwd_issue / left_step), and
// prices candidate placements (node -> start byte) by the distinct 128-byte blocks per step — the unit the memory
// system moves (tools/microbench/gather_pair.hip).
//
//   g++ -O2 -std=c++17 -pthread tools/layout_sim.cpp rust-pseudoaligner_amd/csrc/{host_index,dbg_build,device_flatten,synth}.cpp -lz -o /tmp/layout_sim
//   /tmp/layout_sim <k> <read_len> <ppm> <n_reads> [fasta]
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <unordered_map>
#include <vector>

#include "../rust-pseudoaligner_amd/csrc/device_flatten.hpp"
#include "../rust-pseudoaligner_amd/csrc/dict_slots.hpp"
#include "../rust-pseudoaligner_amd/csrc/lane_steps.hpp"
#include "../rust-pseudoaligner_amd/csrc/pa_common.hpp"

using namespace pa;

struct Touch {   // one step: node + byte ranges [a, b) relative to the blob start
    uint32_t read, node;
    uint32_t r[4][2];
    uint32_t nr;
};

static std::vector<uint64_t> place_blobs(const pa_flat_index& f, const std::vector<uint32_t>& order, uint32_t gran, uint32_t keep_together, uint64_t* total) {
    // blobs in `order`, start rounded up to `gran`; the first `keep_together` bytes (clipped to the blob) never straddle a 128-byte block
    std::vector<uint64_t> at(f.num_nodes);
    uint64_t cur = 0;
    for (uint32_t i : order) {
        const uint64_t size = BLOB_HDR_BYTES + 8ull * ((f.node_len[i] + 31) / 32);
        cur = (cur + gran - 1) / gran * gran;
        const uint64_t kt = std::min<uint64_t>(keep_together, size);
        if (cur / 128 != (cur + kt - 1) / 128) cur = (cur + 127) / 128 * 128;
        at[i] = cur;
        cur += size;
    }
    *total = cur;
    return at;
}

int main(int argc, char** argv) {
    const uint32_t k = argc > 1 ? atoi(argv[1]) : 24, read_len = argc > 2 ? atoi(argv[2]) : 150, ppm = argc > 3 ? atoi(argv[3]) : 0;
    const uint64_t n = argc > 4 ? atoll(argv[4]) : 100000;
    pa_txome* tx = nullptr;
    if (argc > 5) { if (pa_txome_from_fasta(argv[5], &tx)) { fprintf(stderr, "%s\n", pa_last_error()); return 1; } }
    else if (pa_txome_synthesize(58000, 203000, 7, &tx)) { fprintf(stderr, "%s\n", pa_last_error()); return 1; }
    const uint64_t *packed, *tx_start;
    uint32_t num_tx;
    pa_txome_view(tx, &packed, &tx_start, &num_tx);
    char cache[256];
    snprintf(cache, sizeof cache, "/tmp/layout_sim_%s_k%u.idx", argc > 5 ? "fasta" : "synth", k);
    pa_host_index* h = nullptr;
    if (pa_host_index_load(cache, &h) != PA_OK) {
        if (pa_host_index_build_packed(packed, tx_start, num_tx, k, 8, &h)) { fprintf(stderr, "%s\n", pa_last_error()); return 1; }
        pa_host_index_save(h, cache);
    }
    pa_flat_index f;
    pa_host_index_view(h, &f);
    fprintf(stderr, "index: %u nodes, %u classes\n", f.num_nodes, f.num_classes);
    FlatDevice fd;
    if (flatten_for_device(f, 8, fd, false)) { fprintf(stderr, "%s\n", pa_last_error()); return 1; }
    const DevIndexView ix = fd.host_view();
    const uint32_t wpr = (read_len + 31) / 32;
    std::vector<uint64_t> tiles(((n + 63) / 64) * wpr * 64);
    std::vector<uint32_t> lens(n);
    pa_simulate_reads_host(tx, read_len, k == 31 ? 4 : 2, ppm, 0, n, wpr, tiles.data(), lens.data());

    std::vector<Touch> touches;
    std::vector<uint64_t> rd(wpr + 2);
    alignas(16) uint32_t refs[4], lens4[4], cids4[4], win4[4];
    uint32_t wcand[2];
    std::vector<uint32_t> spill(8 * read_len + 64), trace(8 * read_len + 64), pend(8 * read_len + 64);
    uint64_t n_fwd = 0, n_left = 0, n_seek = 0, hops_next = 0, hops = 0, hop_small = 0, hop_small_fits = 0, hop_mid = 0;
    std::vector<uint32_t> succ_count(f.num_nodes, 0);
    std::map<std::pair<uint32_t, uint32_t>, uint32_t> edge_use;
    std::map<uint32_t, uint64_t> ncol_hist, maxlen_hist, base_hist, bigmax_hist;
    std::map<std::pair<uint32_t, uint32_t>, uint64_t> big_hist;
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t t = i >> 6, r = i & 63;
        for (uint32_t w = 0; w < wpr; ++w) rd[w] = tiles[(t * wpr + w) * 64 + r];
        rd[wpr] = rd[wpr + 1] = 0;
        Lane s;
        lane_start(s, (uint32_t)i, lens[i], ix.k);
        const ReadRef rr{rd.data(), 1, wpr};
        const ColRef cr{win4, wcand, refs, lens4, cids4, spill.data(), (uint32_t)spill.size(), pend.data(), trace.data()};
        uint32_t prev_node = 0xFFFFFFFFu;
        while (l_st(s) == ST_SEEK || l_st(s) == ST_FWD || l_st(s) == ST_LEFT) {
            if (l_st(s) == ST_SEEK) { seek_step(s, ix, rr); ++n_seek; prev_node = 0xFFFFFFFFu; }
            else if (l_st(s) == ST_FWD) {
                // the loads of fwd_issue
                const uint32_t K = ix.k, L = l_L(s);
                const bool fresh = l_flags(s) & F_FRESH;
                const uint32_t ro0 = fresh ? l_off(s) + K : (s.rr & 0xFFFFFFu);
                const uint32_t kp0 = fresh ? l_kp(s) + K : l_kp(s);
                const uint32_t most = pa_min(fresh ? L - kp0 : (s.rm & 0xFFFFu), 128u), nwords = ((ro0 & 31) + most + 31) >> 5;
                Touch tc;
                tc.read = (uint32_t)i;
                tc.node = fd.nid_of_handle[s.h];
                tc.nr = 0;
                tc.r[tc.nr][0] = 0; tc.r[tc.nr][1] = 48; ++tc.nr;
                const uint32_t sb = BLOB_HDR_BYTES + 8 * (ro0 >> 5);
                tc.r[tc.nr][0] = sb; tc.r[tc.nr][1] = sb + 16; ++tc.nr;
                if (nwords > 2) { tc.r[tc.nr][0] = sb + 16; tc.r[tc.nr][1] = sb + 32; ++tc.nr; }
                if (nwords > 4) { tc.r[tc.nr][0] = sb + 32; tc.r[tc.nr][1] = sb + 48; ++tc.nr; }
                touches.push_back(tc);
                if (fresh && l_off(s) == 0 && prev_node != 0xFFFFFFFFu) {
                    ++hops; edge_use[{prev_node, tc.node}]++;
                    // tuning question: how often is the node hopped into a small blob (<= 64 B), and does the predecessor's last block have room for it?
                    const uint32_t bsz = BLOB_HDR_BYTES + 8 * ((f.node_len[tc.node] + 31) / 32), psz = BLOB_HDR_BYTES + 8 * ((f.node_len[prev_node] + 31) / 32);
                    hop_small += bsz <= 64;
                    hop_small_fits += bsz <= 64 && ((psz + 127) / 128 * 128 - psz) >= 64;
                    hop_mid += bsz > 64 && bsz <= 128;
                }
                prev_node = tc.node;
                fwd_step<true>(s, ix, rr, cr, 2);
                ++n_fwd;
            } else {
                Touch tc;
                tc.read = (uint32_t)i;
                tc.node = fd.nid_of_handle[s.ph];
                tc.nr = 0;
                tc.r[tc.nr][0] = 0; tc.r[tc.nr][1] = 48; ++tc.nr;
                const uint32_t na = (l_flags(s) & F_FRESH) && !(l_flags(s) & F_LEFT_SEED) ? f.node_len[tc.node] - ix.k + 1 : (s.rr & 0xFFFFFFu);
                if (na) {
                    const uint32_t po = na - 1, st = po >= 31 ? po - 31 : 0;
                    tc.r[tc.nr][0] = BLOB_HDR_BYTES + 8 * (st >> 5); tc.r[tc.nr][1] = tc.r[tc.nr][0] + 16; ++tc.nr;
                }
                touches.push_back(tc);
                left_step<true>(s, ix, rr, cr, 2);
                ++n_left;
                prev_node = 0xFFFFFFFFu;
            }
        }
        if ((l_flags(s) & F_LISTS) && l_st(s) == ST_ISECT) {
            ncol_hist[l_ncol(s)]++;
            uint32_t mx = 0, mn = 0xFFFFFFFFu;
            for (uint32_t c = 0; c < l_ncol(s); ++c) { uint32_t ref, len; get_class(cr, c, ref, len); mx = std::max(mx, len); mn = std::min(mn, len); }
            maxlen_hist[mx <= 8 ? 8 : mx <= 16 ? 16 : mx <= 32 ? 32 : mx <= 64 ? 64 : mx <= 128 ? 128 : mx <= 256 ? 256 : mx <= 1024 ? 1024 : 1u << 20]++;
            base_hist[mn <= 8 ? 8 : mn <= 64 ? 64 : 1u << 20]++;
            // which of the classes do not fit two 32-id windows (the only reason the read is in list mode)?
            uint32_t nbig = 0, bigmax = 0;
            for (uint32_t c = 0; c < l_ncol(s); ++c) {
                uint32_t ref, len; get_class(cr, c, ref, len);
                const uint32_t* ids = fd.ec.data() + 4ull * ref + 1;
                uint32_t j = 0;
                while (j < len && ids[j] - ids[0] < 32) ++j;
                bool fits = true;
                if (j < len) { const uint32_t b2 = ids[j]; for (; j < len; ++j) if (ids[j] - b2 >= 32) fits = false; }
                if (!fits) { ++nbig; bigmax = std::max(bigmax, len); }
            }
            const uint32_t nwin = l_ncol(s) - nbig;
            big_hist[{nwin ? 1u : 0u, std::min(nbig, 5u)}]++;
            bigmax_hist[bigmax <= 8 ? 8 : bigmax <= 16 ? 16 : bigmax <= 32 ? 32 : bigmax <= 64 ? 64 : bigmax <= 128 ? 128 : bigmax <= 256 ? 256 : bigmax <= 1024 ? 1024 : 1u << 20]++;
        }
    }
    {
        uint64_t tot = 0;
        for (auto& kv : ncol_hist) tot += kv.second;
        fprintf(stderr, "list-mode reads: %llu of %llu; distinct classes per such read:", (unsigned long long)tot, (unsigned long long)n);
        uint64_t acc = 0;
        for (auto& kv : ncol_hist) { acc += kv.second; if (kv.first <= 12 || kv.first % 8 == 0) fprintf(stderr, " <=%u:%.1f%%", kv.first, 100.0 * acc / (tot ? tot : 1)); }
        fprintf(stderr, "\n   longest class list of such a read:");
        for (auto& kv : maxlen_hist) fprintf(stderr, " <=%u:%.1f%%", kv.first, 100.0 * kv.second / (tot ? tot : 1));
        fprintf(stderr, "\n   (has window classes, classes without windows [5 = more]):");
        for (auto& kv : big_hist) fprintf(stderr, " (%u,%u):%.1f%%", kv.first.first, kv.first.second, 100.0 * kv.second / (tot ? tot : 1));
        fprintf(stderr, "\n   longest class without windows:");
        for (auto& kv : bigmax_hist) fprintf(stderr, " <=%u:%.1f%%", kv.first, 100.0 * kv.second / (tot ? tot : 1));
        fprintf(stderr, "\n   shortest (the base):");
        for (auto& kv : base_hist) fprintf(stderr, " <=%u:%.1f%%", kv.first, 100.0 * kv.second / (tot ? tot : 1));
        fprintf(stderr, "\n");
    }
    fprintf(stderr, "hops into blobs <= 64 B: %.3f/read (%.3f/read where the predecessor's last block has 64 B free), into blobs of 65..128 B: %.3f/read\n", (double)hop_small / n,
            (double)hop_small_fits / n, (double)hop_mid / n);
    fprintf(stderr, "reads %llu: seek %.3f fwd %.3f left %.3f steps/read, hops %.3f/read\n", (unsigned long long)n, (double)n_seek / n, (double)n_fwd / n,
            (double)n_left / n, (double)hops / n);

    // ---- candidate orders ----
    const uint32_t N = f.num_nodes;
    std::vector<uint32_t> index_order(N);
    for (uint32_t i = 0; i < N; ++i) index_order[i] = i;
    // chain order: follow right edges greedily from nodes without a left extension first
    const uint32_t topshift = 2 * (k - 1);
    std::unordered_map<uint64_t, uint32_t> first_of;   // first k-mer -> node (k <= 32 only in this tool)
    first_of.reserve(N * 2);
    std::vector<uint64_t> seq_pad(f.node_seq, f.node_seq + (f.seq_bases + 31) / 32);
    seq_pad.resize(seq_pad.size() + 3, 0);
    for (uint32_t i = 0; i < N; ++i) first_of[get_kmer(seq_pad.data(), f.node_start[i], k)] = i;
    auto succ = [&](uint32_t i, uint32_t base) -> uint32_t {
        if (!(f.node_exts[i] & (1u << base))) return 0xFFFFFFFFu;
        const uint64_t last = get_kmer(seq_pad.data(), f.node_start[i] + f.node_len[i] - k, k);
        auto it = first_of.find((last >> 2) | ((uint64_t)base << topshift));
        return it == first_of.end() ? 0xFFFFFFFFu : it->second;
    };
    std::vector<uint32_t> chain_order;
    {
        std::vector<uint8_t> placed(N, 0);
        auto run = [&](uint32_t start) {
            uint32_t cur = start;
            while (cur != 0xFFFFFFFFu && !placed[cur]) {
                placed[cur] = 1;
                chain_order.push_back(cur);
                uint32_t nxt = 0xFFFFFFFFu;
                for (uint32_t b = 0; b < 4; ++b) {
                    const uint32_t t = succ(cur, b);
                    if (t != 0xFFFFFFFFu && !placed[t]) { nxt = t; break; }
                }
                cur = nxt;
            }
        };
        for (uint32_t i = 0; i < N; ++i) if (!(f.node_exts[i] >> 4)) run(i);
        for (uint32_t i = 0; i < N; ++i) run(i);
    }
    // transcript order: nodes in order of first appearance along the transcripts (what a builder that knows the transcripts can do)
    std::vector<uint32_t> tx_order;
    {
        std::vector<uint8_t> placed(N, 0);
        // k-mer -> node via the flat dictionary
        for (uint32_t t = 0; t < num_tx; ++t) {
            const uint64_t a = tx_start[t], b = tx_start[t + 1];
            if (b - a < k) continue;
            uint64_t p = a;
            while (p + k <= b) {
                const uint64_t km = get_kmer(packed, p, k);
                // find the node through the host dictionary of the flattened index
                uint32_t hh = NO_HANDLE, off = 0, probes = 0;
                if (!dict_find64(ix.table, (uint32_t)ix.nbuckets, km, hh, off, probes)) hh = NO_HANDLE;
                if (hh == NO_HANDLE) { ++p; continue; }
                const uint32_t node = fd.nid_of_handle[hh];
                if (!placed[node]) { placed[node] = 1; tx_order.push_back(node); }
                p += f.node_len[node] - k + 1 - off;   // jump to the k-mer after this node's last
            }
        }
        for (uint32_t i = 0; i < N; ++i) if (!placed[i]) tx_order.push_back(i);
    }

    {   // a header copy in EVERY 128-byte block of a blob (48 B header + 80 B = 10 sequence words): a step touches the blocks of its words only
        uint64_t blocks = 0;
        for (const Touch& tc : touches) {
            if (tc.nr < 2) { ++blocks; continue; }
            const uint32_t w0 = (tc.r[1][0] - BLOB_HDR_BYTES) / 8, w1 = (tc.r[tc.nr - 1][1] - BLOB_HDR_BYTES) / 8 - 1;
            blocks += w1 / 10 - w0 / 10 + 1;
        }
        printf("%-48s blocks/step %.3f  blocks/read %.3f\n", "header copy in every block (10 words per block)", (double)blocks / touches.size(), (double)blocks / n);
    }
    struct Cand { const char* name; const std::vector<uint32_t>* order; uint32_t gran, keep; };
    const Cand cands[] = {
        {"index order, 128-aligned (now)", &index_order, 128, 0},
        {"index order, 64 B granule", &index_order, 64, 64},
        {"index order, 16 B granule, hdr+16 together", &index_order, 16, 64},
        {"chain order, 128-aligned", &chain_order, 128, 0},
        {"chain order, 64 B granule", &chain_order, 64, 64},
        {"chain order, 16 B granule, hdr together", &chain_order, 16, 48},
        {"chain order, 16 B granule, hdr+16 together", &chain_order, 16, 64},
        {"chain order, 16 B granule, hdr+32 together", &chain_order, 16, 80},
        {"chain order, 16 B granule, free", &chain_order, 16, 0},
        {"tx order, 16 B granule, hdr+16 together", &tx_order, 16, 64},
        {"tx order, 64 B granule", &tx_order, 64, 64},
    };
    for (const Cand& c : cands) {
        uint64_t total = 0;
        const std::vector<uint64_t> at = place_blobs(f, *c.order, c.gran, c.keep, &total);
        uint64_t per_step = 0, per_read = 0, lines64 = 0, clipped = 0;
        std::set<uint64_t> seen_read, seen_step, seen64;
        uint32_t cur_read = 0xFFFFFFFFu;
        for (const Touch& tc : touches) {
            if (tc.read != cur_read) { per_read += seen_read.size(); seen_read.clear(); cur_read = tc.read; }
            seen_step.clear();
            seen64.clear();
            for (uint32_t j = 0; j < tc.nr; ++j)
                for (uint64_t b = (at[tc.node] + tc.r[j][0]); b < at[tc.node] + tc.r[j][1]; b += 16) { seen_step.insert(b / 128); seen_read.insert(b / 128); seen64.insert(b / 64); }
            per_step += seen_step.size();
            lines64 += seen64.size();
            {   // the same step if loads past the node's last sequence word were not issued
                std::set<uint64_t> sc;
                const uint64_t bsz = BLOB_HDR_BYTES + 8ull * ((f.node_len[tc.node] + 31) / 32);
                for (uint32_t j = 0; j < tc.nr; ++j)
                    for (uint64_t b = (at[tc.node] + tc.r[j][0]); b < at[tc.node] + std::min<uint64_t>(tc.r[j][1], bsz); b += 8) sc.insert(b / 128);
                clipped += sc.size();
            }
        }
        per_read += seen_read.size();
        printf("%-48s blobs %7.1f MB  blocks/step %.3f  blocks/read (steps summed) %.3f  distinct blocks/read %.3f  64B lines/read %.3f  clipped-to-node blocks/read %.3f\n", c.name, total / 1e6,
               (double)per_step / touches.size(), (double)per_step / n, (double)per_read / n, (double)lines64 / n, (double)clipped / n);
    }
    return 0;
}
