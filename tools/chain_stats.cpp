// Tuning aid (CPU only): what does the chain-block layout (device_layout.hpp) do to a workload? Builds the index of a workload,
// flattens it, and runs the product's lane state machine on the host (as tests/emu does) over simulated reads: chains per node,
// blocks, and per read the dictionary probes, forward / left steps and DISTINCT 128-byte chain blocks fetched — the unit the
// memory system moves and the mapping kernel is bound by.
//
//   g++ -O2 -std=c++17 -pthread tools/chain_stats.cpp rust-pseudoaligner_amd/csrc/{host_index,dbg_build,device_flatten,synth}.cpp -lz -o /tmp/chain_stats
//   /tmp/chain_stats <k> <read_len> <ppm> <n_reads> [fasta]
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#include "../rust-pseudoaligner_amd/csrc/device_flatten.hpp"
#include "../rust-pseudoaligner_amd/csrc/dict_slots.hpp"
#include "../rust-pseudoaligner_amd/csrc/lane_steps.hpp"
#include "../rust-pseudoaligner_amd/csrc/pa_common.hpp"

using namespace pa;

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
    snprintf(cache, sizeof cache, "/tmp/chain_stats_%s_k%u.idx", argc > 5 ? "fasta" : "synth", k);
    pa_host_index* h = nullptr;
    if (pa_host_index_load(cache, &h) != PA_OK) {
        if (pa_host_index_build_packed(packed, tx_start, num_tx, k, 8, &h)) { fprintf(stderr, "%s\n", pa_last_error()); return 1; }
        pa_host_index_save(h, cache);
    }
    pa_flat_index f;
    pa_host_index_view(h, &f);
    FlatDevice fd;
    if (flatten_for_device(f, 8, fd, false)) { fprintf(stderr, "%s\n", pa_last_error()); return 1; }
    const DevIndexView ix = fd.host_view();
    uint64_t mergeable = 0;
    {
        // nodes whose only right extension leads to a node whose only left extension leads back (what a chain may merge)
        std::vector<uint32_t> cnt(4, 0);
        for (uint32_t i = 0; i < f.num_nodes; ++i) {
            const uint32_t re = f.node_exts[i] & 15u;
            if (re && !(re & (re - 1))) ++mergeable;   // upper bound (the successor's left side is not checked here)
        }
    }
    printf("index: %u nodes (%llu with one right extension), %u chains, %zu blocks (%.1f MB), %u classes\n", f.num_nodes, (unsigned long long)mergeable,
           fd.num_chains, fd.blobs.size() / CH_BLOCK, fd.blobs.size() / 1e6, f.num_classes);
    // slot use per block
    {
        uint64_t hist[5] = {0, 0, 0, 0, 0};
        for (size_t b = 0; b + CH_BLOCK <= fd.blobs.size(); b += CH_BLOCK) {
            const uint32_t* sl = reinterpret_cast<const uint32_t*>(fd.blobs.data() + b);
            const uint32_t rm = sl[0] >> SEG_RECMASK_SHIFT;
            uint32_t used = 0;
            for (uint32_t t = 0; t < 4; ++t)
                if ((rm >> t) & 1u) { const uint32_t w0 = sl[4 * t]; used += 1 + ((w0 & SEG_WIDE) ? 1 : 0) + ((w0 & SEG_EDGES) ? 1 : 0); }
            hist[std::min(used, 4u)]++;
        }
        printf("slots used per block: 1:%llu 2:%llu 3:%llu 4:%llu\n", (unsigned long long)hist[1], (unsigned long long)hist[2], (unsigned long long)hist[3], (unsigned long long)hist[4]);
    }
    const uint32_t wpr = (read_len + 31) / 32;
    std::vector<uint64_t> tiles(((n + 63) / 64) * wpr * 64);
    std::vector<uint32_t> lens(n);
    pa_simulate_reads_host(tx, read_len, k == 31 ? 4 : 2, ppm, 0, n, wpr, tiles.data(), lens.data());
    std::vector<uint64_t> rd(wpr + 2);
    alignas(16) uint32_t refs[4], lens4[4], cids4[4], win4[4];
    uint32_t wcand[2];
    std::vector<uint32_t> spill(8 * read_len + 64), trace(8 * read_len + 64), pend(8 * read_len + 64);
    uint64_t n_unknown = 0, n_seek = 0, n_fwd = 0, n_left = 0, n_blocks = 0, n_nodes = 0, n_lists = 0, hop_edges = 0, hop_unique = 0, hop_unique_rem = 0;
    std::map<uint32_t, uint64_t> rem_hist;
    std::map<uint32_t, uint64_t> fwd_hist;
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t t = i >> 6, r = i & 63;
        for (uint32_t w = 0; w < wpr; ++w) rd[w] = tiles[(t * wpr + w) * 64 + r];
        rd[wpr] = rd[wpr + 1] = 0;
        Lane s;
        lane_start(s, (uint32_t)i, lens[i], ix.k);
        const ReadRef rr{rd.data(), 1, wpr};
        const ColRef cr{win4, wcand, refs, lens4, cids4, spill.data(), (uint32_t)spill.size(), pend.data(), trace.data()};
        std::vector<uint32_t> blocks;
        uint32_t nf = 0;
        for (;;) {
            while (l_st(s) == ST_SEEK || l_st(s) == ST_FWD || l_st(s) == ST_LEFT) {
                if (l_st(s) == ST_SEEK) { seek_step(s, ix, rr); ++n_seek; }
                else if (l_st(s) == ST_FWD) {
                    blocks.push_back(s.h);
                    if (!(s.of & OF_CUR_KNOWN)) ++n_unknown;
                    fwd_step<true>(s, ix, rr, cr, 2);
                    ++n_fwd; ++nf;
                    if (l_st(s) == ST_FWD && (l_flags(s) & F_FRESH) && l_ntrace(s)) {   // left its chain over an edge
                        ++hop_edges;
                        const uint32_t from = trace[l_ntrace(s) - 1], re = f.node_exts[from] & 15u;
                        if (!(re & (re - 1))) { ++hop_unique; hop_unique_rem += l_L(s) - (l_kp(s) + ix.k); rem_hist[(l_L(s) - (l_kp(s) + ix.k)) / 16]++; }
                    }
                }
                else { blocks.push_back(s.ph); left_step<true>(s, ix, rr, cr, 2); ++n_left; }
            }
            if (l_st(s) != ST_ISECT || (l_flags(s) & F_LISTS)) break;
            if (window_todo(s) == 2) { restart_lists(s, ix.k); ++n_lists; continue; }
            break;
        }
        fwd_hist[nf]++;
        n_nodes += l_ntrace(s);
        std::sort(blocks.begin(), blocks.end());
        n_blocks += std::unique(blocks.begin(), blocks.end()) - blocks.begin();
    }
    printf("per read: %.3f probes, %.3f forward steps, %.3f left steps, %.3f distinct chain blocks, %.3f nodes pushed, %.4f list-mode restarts\n", (double)n_seek / n,
           (double)n_fwd / n, (double)n_left / n, (double)n_blocks / n, (double)n_nodes / n, (double)n_lists / n);
    printf("per read: %.3f hops over an edge, %.3f of them from a node with ONE right extension (mean read bases left then: %.1f)\n", (double)hop_edges / n,
           (double)hop_unique / n, hop_unique ? (double)hop_unique_rem / hop_unique : 0.0);
    printf("forward steps that had to look for their node's slot: %.4f per read\n", (double)n_unknown / n);
    printf("forward steps per read:");
    for (auto& kv : fwd_hist) printf(" %u:%.3f", kv.first, (double)kv.second / n);
    printf("\n");
    return 0;
}
