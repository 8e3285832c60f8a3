// Synthetic workloads for BASELINE.json's configs: a GENCODE-like transcriptome and the read
// simulator (host version; the device version in kernels.hip computes the same function of (seed, read index)).
#include <algorithm>
#include <cmath>

#include "pa_common.hpp"
#include "synth_common.hpp"

namespace pa {
namespace {

struct Xoshiro {   // xoshiro256**, splitmix64-seeded
    uint64_t s[4];
    explicit Xoshiro(uint64_t seed) { for (auto& x : s) x = splitmix64(seed); }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    uint64_t below(uint64_t n) { return (uint64_t)(((unsigned __int128)next() * n) >> 64); }
    double normal() {
        double u1 = uniform(), u2 = uniform();
        if (u1 < 1e-300) u1 = 1e-300;
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
};

}  // namespace
}  // namespace pa

using namespace pa;

extern "C" {

// GENCODE-like: genes of 4..16 exons, exon length log-normal (median 130, sigma 0.9, min 30), isoforms = random exon
// subsets (each exon kept with p = 0.75, at least 2), 1 + Exp(mean target/genes - 1) isoforms per gene, 5 % of the genes
// are paralogs: a copy of an earlier gene's exons with 3 % substitutions.
static int synthesize_impl(uint32_t num_genes, uint32_t target_transcripts, uint64_t seed, const pa_synth_repeats* rep, pa_txome** out) {
    if (!out || num_genes == 0 || target_transcripts < num_genes) return fail(PA_ERR_INVALID_ARG, "bad synth arguments");
    // interspersed repeats and low-complexity tracts (config3r): decided by a generator of their OWN, so that the genes, exons and isoforms
    // are those of the plain transcriptome of the same seed — only the last exons of some genes grow
    Xoshiro rrng(seed ^ 0x5265706561747321ull);
    std::vector<std::vector<uint8_t>> families;
    if (rep) {
        if (rep->element_len == 0 || rep->element_len > 5000) return fail(PA_ERR_INVALID_ARG, "bad repeat element length");
        families.resize((size_t)rep->families + rep->young_families);
        for (auto& f : families) { f.resize(rep->element_len); for (auto& b : f) b = (uint8_t)(rrng.next() >> 62); }
    }
    uint32_t low_left = rep ? rep->low_complexity_genes : 0;
    pa_txome* t = new pa_txome();
    Txome& x = t->t;
    x.tx_start.push_back(0);
    Xoshiro rng(seed);
    std::vector<std::vector<std::vector<uint8_t>>> gene_exons;   // kept for paralog copies
    gene_exons.reserve(num_genes);
    // isoforms per gene = 1 + floor(Exp(m)); E[floor(Exp(m))] = 1/(e^(1/m) - 1) = x  <=>  m = 1/ln(1 + 1/x)
    const double xmean = (double)target_transcripts / num_genes - 1.0;
    const double extra = xmean > 0 ? 1.0 / std::log(1.0 + 1.0 / xmean) : 0.0;
    uint64_t pos = 0;
    auto push = [&](uint32_t b) {
        if ((pos & 31) == 0) x.packed.push_back(0);
        x.packed.back() |= (uint64_t)b << ((pos & 31) * 2);
        ++pos;
    };
    char name[64];
    for (uint32_t g = 0; g < num_genes; ++g) {
        std::vector<std::vector<uint8_t>> exons;
        if (g >= 20 && rng.uniform() < 0.05) {
            exons = gene_exons[rng.below(g)];
            for (auto& e : exons)
                for (auto& b : e)
                    if (rng.uniform() < 0.03) b = (uint8_t)((b + 1 + rng.below(3)) & 3);
        } else {
            const uint32_t ne = 4 + (uint32_t)rng.below(13);
            exons.resize(ne);
            for (auto& e : exons) {
                double l = 130.0 * std::exp(0.9 * rng.normal());
                uint32_t len = (uint32_t)l;
                if (len < 30) len = 30;
                if (len > 6000) len = 6000;
                e.resize(len);
                for (auto& b : e) b = (uint8_t)(rng.next() >> 62);
            }
        }
        std::vector<uint8_t> grown;   // the last exon with its repeat / tract (the exons themselves stay as they are: paralogs copy them, and the base generator's stream must not move)
        if (rep && !exons.empty()) {   // the gene's last exon ("3' UTR") takes a copy of a repeat family / a low-complexity tract: every isoform that keeps the exon carries it
            grown = exons.back();
            std::vector<uint8_t>& last = grown;
            if (!families.empty() && rrng.below(1000000) < rep->gene_fraction_ppm) {
                const size_t f = (size_t)rrng.below(families.size());
                const bool young = f >= rep->families;
                const uint32_t lo = young ? rep->young_div_lo_ppm : rep->div_lo_ppm, hi = young ? rep->young_div_hi_ppm : rep->div_hi_ppm;
                const uint32_t div = lo + (uint32_t)rrng.below((uint64_t)(hi > lo ? hi - lo : 0) + 1);   // this copy's distance from its family's consensus
                const size_t at = last.size() - (size_t)rrng.below(std::min<uint64_t>(last.size(), 40) + 1);
                std::vector<uint8_t> copy = families[f];
                for (auto& b : copy)
                    if (rrng.below(1000000) < div) b = (uint8_t)((b + 1 + rrng.below(3)) & 3);
                last.insert(last.begin() + (long)at, copy.begin(), copy.end());
            }
            if (low_left && rrng.below(num_genes - g) < low_left) {   // (exactly low_complexity_genes genes, spread over the whole set)
                --low_left;
                const uint32_t kind = (uint32_t)rrng.below(3), units = 30 + (uint32_t)rrng.below(60);
                static const uint8_t unit[3][3] = {{0, 0, 0}, {1, 0, 1}, {1, 0, 2}};   // poly-A, (CA)n, (CAG)n
                const uint32_t ulen = kind == 0 ? 1u : kind == 1 ? 2u : 3u;
                for (uint32_t u = 0; u < units; ++u)
                    for (uint32_t j = 0; j < ulen; ++j) last.push_back(unit[kind][j]);
            }
        }
        uint32_t niso = 1;
        if (extra > 0) niso += (uint32_t)(-std::log(1.0 - rng.uniform()) * extra);
        if (niso > 60) niso = 60;
        for (uint32_t i = 0; i < niso; ++i) {
            std::vector<uint32_t> keep;
            do {
                keep.clear();
                for (uint32_t e = 0; e < exons.size(); ++e)
                    if (rng.uniform() < 0.75) keep.push_back(e);
            } while (keep.size() < 2);
            for (uint32_t e : keep)
                for (uint8_t b : (rep && e + 1 == exons.size() ? grown : exons[e])) push(b);
            x.tx_start.push_back(pos);
            snprintf(name, sizeof name, "SYNT%08u.%u", g, i);
            x.names.push_back(name);
            snprintf(name, sizeof name, "SYNG%08u", g);
            x.genes.push_back(name);
        }
        gene_exons.push_back(std::move(exons));
    }
    x.packed.push_back(0);
    x.packed.push_back(0);
    *out = t;
    return PA_OK;
}

int pa_txome_synthesize(uint32_t num_genes, uint32_t target_transcripts, uint64_t seed, pa_txome** out) {
    return synthesize_impl(num_genes, target_transcripts, seed, nullptr, out);
}

int pa_txome_synthesize_repeats(uint32_t num_genes, uint32_t target_transcripts, uint64_t seed, const pa_synth_repeats* repeats, pa_txome** out) {
    if (!repeats) return fail(PA_ERR_INVALID_ARG, "null argument");
    return synthesize_impl(num_genes, target_transcripts, seed, repeats, out);
}

int pa_txome_from_host_index(const pa_host_index* h, pa_txome** out) {
    if (!h || !out) return fail(PA_ERR_INVALID_ARG, "null argument");
    if (h->h.tx_start.size() < 2) return fail(PA_ERR_INVALID_ARG, "host index carries no transcripts");
    pa_txome* t = new pa_txome();
    t->t.packed = h->h.tx_packed;
    t->t.tx_start = h->h.tx_start;
    t->t.names = h->h.tx_names;
    t->t.genes = h->h.tx_genes;
    *out = t;
    return PA_OK;
}

int pa_txome_from_fasta(const char* path, pa_txome** out) {
    if (!path || !out) return fail(PA_ERR_INVALID_ARG, "null argument");
    pa_txome* t = new pa_txome();
    int rc = read_fasta(path, t->t);
    if (rc != PA_OK) { delete t; return rc; }
    *out = t;
    return PA_OK;
}

int pa_txome_view(const pa_txome* t, const uint64_t** packed, const uint64_t** tx_start, uint32_t* num_tx) {
    if (!t) return fail(PA_ERR_INVALID_ARG, "null argument");
    if (packed) *packed = t->t.packed.data();
    if (tx_start) *tx_start = t->t.tx_start.data();
    if (num_tx) *num_tx = t->t.num_tx();
    return PA_OK;
}

void pa_txome_destroy(pa_txome* t) { delete t; }

int pa_simulate_reads_host(const pa_txome* t, uint32_t read_len, uint64_t seed, uint32_t sub_rate_ppm, uint64_t first_read,
                           uint64_t n_reads, uint32_t words_per_read, uint64_t* tiles, uint32_t* lens) {
    if (!t || !tiles || !lens) return fail(PA_ERR_INVALID_ARG, "null argument");
    if (read_len == 0 || words_per_read < (read_len + 31) / 32) return fail(PA_ERR_INVALID_ARG, "words_per_read too small");
    if (read_len > PA_MAX_SIM_READ_LEN) return fail(PA_ERR_INVALID_ARG, "synthetic reads are at most %u bases", PA_MAX_SIM_READ_LEN);
    std::vector<uint64_t> cum;
    synth::build_cum(t->t.tx_start.data(), t->t.num_tx(), read_len, cum);
    const uint64_t total = cum.back();
    if (total == 0) return fail(PA_ERR_INVALID_ARG, "no transcript is at least %u bases long", read_len);
    const uint64_t ntiles = (n_reads + 63) / 64;
    std::memset(tiles, 0, ntiles * words_per_read * 64 * sizeof(uint64_t));
    for (uint64_t i = 0; i < n_reads; ++i) {
        uint64_t words[PA_MAX_SIM_READ_LEN / 32 + 1];
        synth::simulate_read(t->t.packed.data(), t->t.tx_start.data(), cum.data(), t->t.num_tx(), total, read_len, seed,
                             sub_rate_ppm, first_read + i, words);
        const uint64_t tile = i >> 6, r = i & 63;
        for (uint32_t w = 0; w < (read_len + 31) / 32; ++w) tiles[(tile * words_per_read + w) * 64 + r] = words[w];
        lens[i] = read_len;
    }
    return PA_OK;
}

}  // extern "C"
