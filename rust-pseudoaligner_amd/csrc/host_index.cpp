// Host-side index object: FASTA ingest, flat view, own on-disk container, C ABI.
#include <algorithm>
#include <charconv>
#include <cstdio>
#include <cstring>
#include <cerrno>
#include <fstream>
#include <thread>

#include <unordered_map>

#include "pa_common.hpp"

namespace pa {

int usable_threads() {
    int n = (int)std::thread::hardware_concurrency();
    if (n < 1) n = 1;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                 // cgroup v2: "<quota|max> <period>"
        char q[32];
        long long period = 0;
        if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
            const long long c = atoll(q) / period;
            if (c >= 1 && c < n) n = (int)c;
        }
        fclose(f);
    } else {
        long long quota = -1, period = 0;                                // cgroup v1
        if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g); }
        if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &period) != 1) period = 0; fclose(g); }
        if (quota > 0 && period > 0 && quota / period >= 1 && quota / period < n) n = (int)(quota / period);
    }
    return n;
}

}  // namespace pa

namespace pa {

std::string& last_error_ref() {
    static thread_local std::string s;
    return s;
}

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error_ref() = buf;
    return code;
}

namespace {

// utils::detect_fasta_format / extract_tx_gene_id (src/utils.rs:99-150). The reference bails out on an unknown
// header format (:118); index-side metadata is outside the hot path, so unknown headers fall back to gene = tx id.
void tx_gene_from_header(const std::string& id, const std::string& desc, std::string& tx, std::string& gene) {
    size_t bars = 0;
    for (char c : id) bars += (c == '|');
    if (bars == 8) {   // Gencode: 9 '|'-separated tokens
        const size_t a = id.find('|');
        const size_t b = id.find('|', a + 1);
        tx = id.substr(0, a);
        gene = id.substr(a + 1, b - a - 1);
        return;
    }
    tx = id;
    if (desc.compare(0, 5, "gene=") == 0) {   // Gffread
        const size_t sp = desc.find(' ');
        gene = desc.substr(5, sp == std::string::npos ? std::string::npos : sp - 5);
        return;
    }
    const size_t g = desc.find("gene:");   // Ensembl: "... gene:ENSG... ..."
    if (g != std::string::npos) {
        const size_t sp = desc.find(' ', g);
        gene = desc.substr(g + 5, sp == std::string::npos ? std::string::npos : sp - g - 5);
        return;
    }
    gene = id;
}

}  // namespace

// utils::read_transcripts (src/utils.rs:61-97). Non-ACGT bytes become a base derived from a hash of the record id
// and the position, in the spirit of DnaString::from_acgt_bytes_hashn (:76; the crate's exact hash is not on disk —
// unpinned; gencode_small.fa has no such bytes).
int read_fasta(const char* path, Txome& out) {
    std::ifstream in(path, std::ios::binary);
    if (!in) return fail(PA_ERR_IO, "cannot open %s: %s", path, strerror(errno));
    out = Txome();
    out.tx_start.push_back(0);
    std::string line, id, desc;
    uint64_t pos = 0, name_hash = 0;
    bool have = false;
    auto push = [&](uint32_t b) {
        if ((pos & 31) == 0) out.packed.push_back(0);
        out.packed.back() |= (uint64_t)b << ((pos & 31) * 2);
        ++pos;
    };
    auto finish = [&]() {
        if (!have) return;
        out.tx_start.push_back(pos);
        std::string tx, gene;
        tx_gene_from_header(id, desc, tx, gene);
        out.names.push_back(tx);
        out.genes.push_back(gene);
    };
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;
        if (line[0] == '>') {
            finish();
            have = true;
            // record.id() / desc() of bio 1.5's FASTA reader: header[1..].trim_end().splitn(2, char::is_whitespace) — the id ends at
            // the first white-space character of ANY kind (its FASTQ reader splits on ' ' only: fastq.cpp keeps that rule). The id
            // names the transcript AND seeds from_acgt_bytes_hashn (src/utils.rs:76), so a tab-separated header must cut the same
            // way or the bases substituted for N differ. ASCII white space only (Unicode spaces in a FASTA header: unpinned).
            auto is_ws = [](char c) { return c == ' ' || c == '\t' || c == '\x0b' || c == '\x0c' || c == '\r' || c == '\n'; };
            while (!line.empty() && is_ws(line.back())) line.pop_back();
            size_t sp = 1;
            while (sp < line.size() && !is_ws(line[sp])) ++sp;
            if (sp >= line.size()) sp = std::string::npos;
            id = line.substr(1, sp == std::string::npos ? std::string::npos : sp - 1);
            desc = sp == std::string::npos ? std::string() : line.substr(sp + 1);
            name_hash = 0xcbf29ce484222325ull;
            for (char c : id) name_hash = (name_hash ^ (uint8_t)c) * 0x100000001b3ull;
        } else {
            if (!have) return fail(PA_ERR_FORMAT, "%s: sequence before first header", path);
            const uint64_t s = out.tx_start.back();
            for (char c : line) {
                uint32_t b = base_code((uint8_t)c);
                if (b > 3) b = (uint32_t)(mix64(name_hash + (pos - s)) & 3u);
                push(b);
            }
        }
    }
    finish();
    out.packed.push_back(0);
    out.packed.push_back(0);
    return PA_OK;
}

static int check_flat(const pa_flat_index* f) {
    if (!f) return fail(PA_ERR_INVALID_ARG, "null flat index");
    if (f->k < PA_MIN_K || f->k > PA_MAX_K) return fail(PA_ERR_UNSUPPORTED, "k=%u outside [%u,%u]", f->k, PA_MIN_K, PA_MAX_K);
    if (f->num_nodes && (!f->node_seq || !f->node_start || !f->node_len || !f->node_exts || !f->node_colour))
        return fail(PA_ERR_INVALID_ARG, "flat index: null node array");
    if (!f->ec_offset || (f->num_classes && f->ec_offset[f->num_classes] && !f->ec_ids))
        return fail(PA_ERR_INVALID_ARG, "flat index: null class array");
    for (uint32_t i = 0; i < f->num_nodes; ++i) {
        if (f->node_len[i] < f->k) return fail(PA_ERR_FORMAT, "node %u shorter than k", i);
        if (f->node_start[i + 1] - f->node_start[i] != f->node_len[i]) return fail(PA_ERR_FORMAT, "node %u: start/len mismatch", i);
        if (f->node_colour[i] >= f->num_classes) return fail(PA_ERR_FORMAT, "node %u: colour out of range", i);
    }
    for (uint32_t c = 0; c < f->num_classes; ++c) {
        if (f->ec_offset[c + 1] < f->ec_offset[c]) return fail(PA_ERR_FORMAT, "class %u: offsets not monotone", c);
        for (uint64_t j = f->ec_offset[c]; j < f->ec_offset[c + 1]; ++j) {
            if (f->ec_ids[j] >= f->num_transcripts) return fail(PA_ERR_FORMAT, "class %u: transcript id out of range", c);
            if (j > f->ec_offset[c] && f->ec_ids[j] <= f->ec_ids[j - 1]) return fail(PA_ERR_FORMAT, "class %u not sorted/dedup'd", c);
        }
    }
    return PA_OK;
}

}  // namespace pa

using namespace pa;

extern "C" {

uint32_t pa_abi_version(void) { return PA_ABI_VERSION; }
const char* pa_last_error(void) { return last_error_ref().c_str(); }

int pa_host_index_build_packed(const uint64_t* packed, const uint64_t* tx_start, uint32_t num_tx, uint32_t k,
                               int num_threads, pa_host_index** out) {
    if (!packed || !tx_start || !out) return fail(PA_ERR_INVALID_ARG, "null argument");
    if (num_threads <= 0) num_threads = usable_threads();
    pa_host_index* h = new (std::nothrow) pa_host_index();
    if (!h) return fail(PA_ERR_OOM, "out of memory");
    try {
        int rc = build_graph(packed, tx_start, num_tx, k, num_threads, h->h);
        if (rc != PA_OK) { delete h; return rc; }
        const uint64_t nb = tx_start[num_tx];
        h->h.tx_packed.assign(packed, packed + (nb + 31) / 32);
        h->h.tx_packed.push_back(0);
        h->h.tx_packed.push_back(0);
        h->h.tx_start.assign(tx_start, tx_start + num_tx + 1);
    } catch (const std::bad_alloc&) {
        delete h;
        return fail(PA_ERR_OOM, "out of memory while building the index");
    }
    *out = h;
    return PA_OK;
}

int pa_host_index_build_fasta(const char* fasta_path, uint32_t k, int num_threads, pa_host_index** out) {
    if (!fasta_path || !out) return fail(PA_ERR_INVALID_ARG, "null argument");
    Txome t;
    int rc = read_fasta(fasta_path, t);
    if (rc != PA_OK) return rc;
    rc = pa_host_index_build_packed(t.packed.data(), t.tx_start.data(), t.num_tx(), k, num_threads, out);
    if (rc != PA_OK) return rc;
    (*out)->h.tx_names = std::move(t.names);
    (*out)->h.tx_genes = std::move(t.genes);
    return PA_OK;
}

int pa_host_index_from_flat(const pa_flat_index* f, pa_host_index** out) {
    if (!out) return fail(PA_ERR_INVALID_ARG, "null argument");
    int rc = check_flat(f);
    if (rc != PA_OK) return rc;
    pa_host_index* h = new (std::nothrow) pa_host_index();
    if (!h) return fail(PA_ERR_OOM, "out of memory");
    HostIndex& x = h->h;
    x.k = f->k;
    x.num_transcripts = f->num_transcripts;
    const uint64_t nb = f->num_nodes ? f->node_start[f->num_nodes] : 0;
    x.node_seq.assign(f->node_seq, f->node_seq + (nb + 31) / 32);
    x.node_seq.push_back(0);
    x.node_seq.push_back(0);
    x.node_start.assign(f->node_start, f->node_start + f->num_nodes + 1);
    x.node_len.assign(f->node_len, f->node_len + f->num_nodes);
    x.node_exts.assign(f->node_exts, f->node_exts + f->num_nodes);
    x.node_colour.assign(f->node_colour, f->node_colour + f->num_nodes);
    if (f->node_redge) x.node_redge.assign(f->node_redge, f->node_redge + 4ull * f->num_nodes);
    if (f->node_ledge) x.node_ledge.assign(f->node_ledge, f->node_ledge + 4ull * f->num_nodes);
    x.ec_offset.assign(f->ec_offset, f->ec_offset + f->num_classes + 1);
    x.ec_ids.assign(f->ec_ids, f->ec_ids + f->ec_offset[f->num_classes]);
    x.tx_start.push_back(0);
    *out = h;
    return PA_OK;
}

// ---- index comparison (the diff tool of the interchange: an index exported from the Rust side vs the one built here) ----
// Node ids, class ids and node order are arbitrary in both builders; what a read can observe is, per k-mer, the id LIST of
// its class, and per node its sequence and extension bits. Level 1 compares the nodes as a set of (sequence, exts, id
// list); when the node sets differ (unitig break points of a k-mer cycle are not pinned by any reference test) level 2
// compares the k-mer -> id-list map, which is break-point independent.
namespace {

struct IndexDigest {
    std::unordered_map<std::string, uint64_t> nodes;   // packed sequence + exts byte -> hash of the class's id list
    std::unordered_map<std::string, uint64_t> kmers;   // packed k-mer -> hash of the class's id list (level 2 only)
};

uint64_t ids_hash(const HostIndex& x, uint32_t colour) {
    uint64_t h = 0x243f6a8885a308d3ull;
    for (uint64_t i = x.ec_offset[colour]; i < x.ec_offset[colour + 1]; ++i) h = mix64(h ^ x.ec_ids[i]) + 0x9e3779b97f4a7c15ull;
    return h ^ (x.ec_offset[colour + 1] - x.ec_offset[colour]);
}

std::string bases_key(const uint64_t* seq, uint64_t start, uint32_t len) {
    std::string k((len + 3) / 4, '\0');
    for (uint32_t i = 0; i < len; ++i) k[i >> 2] = (char)(k[i >> 2] | (get_base(seq, start + i) << (2 * (i & 3))));
    k.push_back((char)(len & 3));
    return k;
}

void digest_nodes(const HostIndex& x, IndexDigest& d) {
    for (size_t n = 0; n < x.node_len.size(); ++n) {
        std::string key = bases_key(x.node_seq.data(), x.node_start[n], x.node_len[n]);
        key.push_back((char)x.node_exts[n]);
        d.nodes[key] = ids_hash(x, x.node_colour[n]);
    }
}

void digest_kmers(const HostIndex& x, IndexDigest& d) {
    for (size_t n = 0; n < x.node_len.size(); ++n) {
        const uint64_t h = ids_hash(x, x.node_colour[n]);
        for (uint32_t o = 0; o + x.k <= x.node_len[n]; ++o) d.kmers[bases_key(x.node_seq.data(), x.node_start[n] + o, x.k)] = h;
    }
}

}  // namespace

int pa_host_index_compare(const pa_host_index* a, const pa_host_index* b, uint64_t max_kmers, char* report, size_t report_cap) {
    if (!a || !b) return fail(PA_ERR_INVALID_ARG, "null argument");
    auto say = [&](const char* fmt, auto... args) {
        if (report && report_cap) snprintf(report, report_cap, fmt, args...);
    };
    const HostIndex &x = a->h, &y = b->h;
    if (x.k != y.k) { say("k differs: %u vs %u", x.k, y.k); return 1; }
    IndexDigest dx, dy;
    digest_nodes(x, dx);
    digest_nodes(y, dy);
    if (dx.nodes.size() != x.node_len.size() || dy.nodes.size() != y.node_len.size()) { say("an index holds the same node twice"); return 1; }
    uint64_t missing = 0, other_class = 0;
    for (const auto& kv : dx.nodes) {
        const auto it = dy.nodes.find(kv.first);
        if (it == dy.nodes.end()) ++missing;
        else if (it->second != kv.second) ++other_class;
    }
    if (missing == 0 && other_class == 0 && dx.nodes.size() == dy.nodes.size()) {
        say("identical: %zu nodes (sequence, extensions, class id lists), %zu / %zu classes", dx.nodes.size(), x.ec_offset.size() - 1, y.ec_offset.size() - 1);
        return 0;
    }
    if (other_class) { say("%llu nodes carry a different class (id list)", (unsigned long long)other_class); return 1; }
    // node sets differ: are the k-mer -> class maps equal (different unitig break points only)?
    uint64_t kx = 0, ky = 0;
    for (size_t n = 0; n < x.node_len.size(); ++n) kx += x.node_len[n] - x.k + 1;
    for (size_t n = 0; n < y.node_len.size(); ++n) ky += y.node_len[n] - y.k + 1;
    if (kx != ky) { say("%llu vs %llu k-mers (%llu nodes only in the first index)", (unsigned long long)kx, (unsigned long long)ky, (unsigned long long)missing); return 1; }
    if (kx > max_kmers) { say("node sets differ (%llu nodes only in the first index) and the k-mer level check needs %llu k-mers (limit %llu)",
                              (unsigned long long)missing, (unsigned long long)kx, (unsigned long long)max_kmers); return 2; }
    digest_kmers(x, dx);
    digest_kmers(y, dy);
    uint64_t kdiff = 0;
    for (const auto& kv : dx.kmers) {
        const auto it = dy.kmers.find(kv.first);
        if (it == dy.kmers.end() || it->second != kv.second) ++kdiff;
    }
    if (kdiff || dx.kmers.size() != dy.kmers.size()) { say("%llu k-mers are missing or carry a different class", (unsigned long long)kdiff); return 1; }
    say("equivalent: same k-mer -> class map over %zu k-mers; unitig break points differ (%zu vs %zu nodes)", dx.kmers.size(), dx.nodes.size(), dy.nodes.size());
    return 0;
}

int pa_host_index_view(const pa_host_index* h, pa_flat_index* v) {
    if (!h || !v) return fail(PA_ERR_INVALID_ARG, "null argument");
    const HostIndex& x = h->h;
    v->k = x.k;
    v->num_nodes = (uint32_t)x.node_len.size();
    v->num_classes = (uint32_t)(x.ec_offset.size() - 1);
    v->num_transcripts = x.num_transcripts;
    v->seq_bases = x.node_start.back();
    v->node_seq = x.node_seq.data();
    v->node_start = x.node_start.data();
    v->node_len = x.node_len.data();
    v->node_exts = x.node_exts.data();
    v->node_colour = x.node_colour.data();
    v->ec_offset = x.ec_offset.data();
    v->ec_ids = x.ec_ids.data();
    v->node_redge = x.node_redge.empty() ? nullptr : x.node_redge.data();
    v->node_ledge = x.node_ledge.empty() ? nullptr : x.node_ledge.data();
    return PA_OK;
}

uint32_t pa_host_index_num_transcripts(const pa_host_index* h) { return h ? h->h.num_transcripts : 0; }
const char* pa_host_index_tx_name(const pa_host_index* h, uint32_t tx) {
    return (h && tx < h->h.tx_names.size()) ? h->h.tx_names[tx].c_str() : "";
}
const char* pa_host_index_tx_gene(const pa_host_index* h, uint32_t tx) {
    return (h && tx < h->h.tx_genes.size()) ? h->h.tx_genes[tx].c_str() : "";
}
// ---- gene-level collapse (tx_gene_mapping, src/pseudoaligner.rs:32) ----
static void gene_table(const pa_host_index* h, std::vector<uint32_t>& tx_gene, std::vector<const std::string*>& names) {
    std::unordered_map<std::string, uint32_t> id;
    tx_gene.resize(h->h.tx_genes.size());
    for (size_t t = 0; t < h->h.tx_genes.size(); ++t) {
        auto it = id.find(h->h.tx_genes[t]);
        if (it == id.end()) { it = id.emplace(h->h.tx_genes[t], (uint32_t)names.size()).first; names.push_back(&h->h.tx_genes[t]); }
        tx_gene[t] = it->second;
    }
}

int pa_host_index_genes(const pa_host_index* h, uint32_t* tx_gene, uint32_t* num_genes) {
    if (!h) return fail(PA_ERR_INVALID_ARG, "null argument");
    std::vector<uint32_t> tg;
    std::vector<const std::string*> names;
    gene_table(h, tg, names);
    if (tx_gene && !tg.empty()) memcpy(tx_gene, tg.data(), tg.size() * 4);
    if (num_genes) *num_genes = (uint32_t)names.size();
    return PA_OK;
}

const char* pa_host_index_gene_name(const pa_host_index* h, uint32_t gene) {
    if (!h) return "";
    std::vector<uint32_t> tg;
    std::vector<const std::string*> names;
    gene_table(h, tg, names);
    return gene < names.size() ? names[gene]->c_str() : "";
}

int pa_counts_collapse_genes(const pa_host_index* h, const uint64_t* class_counts, uint64_t counts_len, uint64_t* gene_counts) {
    if (!h || !class_counts || !gene_counts) return fail(PA_ERR_INVALID_ARG, "null argument");
    const uint64_t nc = h->h.ec_offset.size() - 1;
    if (counts_len != nc + 3) return fail(PA_ERR_INVALID_ARG, "class table has %llu entries, index has %llu classes (+3)",
                                          (unsigned long long)counts_len, (unsigned long long)nc);
    if (h->h.tx_genes.size() != h->h.num_transcripts) return fail(PA_ERR_FORMAT, "index carries no gene of each transcript");
    std::vector<uint32_t> tg;
    std::vector<const std::string*> names;
    gene_table(h, tg, names);
    const uint32_t multi = (uint32_t)names.size();
    for (uint64_t c = 0; c < nc; ++c) {
        if (!class_counts[c]) continue;
        uint32_t g = multi;
        for (uint64_t j = h->h.ec_offset[c]; j < h->h.ec_offset[c + 1]; ++j) {
            const uint32_t gj = tg[h->h.ec_ids[j]];
            if (j == h->h.ec_offset[c]) g = gj;
            else if (gj != g) { g = multi; break; }
        }
        gene_counts[g] += class_counts[c];
    }
    return PA_OK;
}

// ---- mappability (src/mappability.rs) ----
int pa_host_index_mappability(const pa_host_index* h, uint64_t* tx_mult, uint64_t* gene_mult) {
    if (!h) return fail(PA_ERR_INVALID_ARG, "null argument");
    const pa::HostIndex& x = h->h;
    const size_t ntx = x.num_transcripts, W = PA_MAPPABILITY_COUNTS_LEN;
    if (gene_mult && x.tx_genes.size() != ntx) return fail(PA_ERR_FORMAT, "index carries no gene of each transcript");
    if (tx_mult) memset(tx_mult, 0, ntx * W * sizeof(uint64_t));
    if (gene_mult) memset(gene_mult, 0, ntx * W * sizeof(uint64_t));
    std::vector<uint32_t> tg;
    std::vector<const std::string*> names;
    if (gene_mult) gene_table(h, tg, names);
    // genes per class, once per class (the reference recounts them at every node, :137-144)
    const size_t nc = x.ec_offset.size() - 1;
    std::vector<uint32_t> class_genes(gene_mult ? nc : 0);
    std::vector<uint32_t> scratch;
    for (size_t c = 0; c < class_genes.size(); ++c) {
        scratch.clear();
        for (uint64_t j = x.ec_offset[c]; j < x.ec_offset[c + 1]; ++j) scratch.push_back(tg[x.ec_ids[j]]);
        std::sort(scratch.begin(), scratch.end());
        class_genes[c] = (uint32_t)(std::unique(scratch.begin(), scratch.end()) - scratch.begin());
    }
    for (size_t n = 0; n < x.node_len.size(); ++n) {
        const uint64_t num_kmer = x.node_len[n] - x.k + 1;                        // :131
        const uint32_t c = x.node_colour[n];                                      // :133-134
        const uint64_t num_tx = x.ec_offset[c + 1] - x.ec_offset[c];              // :136
        const size_t tslot = (num_tx > W ? W : num_tx) - 1;                       // add_tx_count (:58-64)
        const size_t gslot = gene_mult ? (class_genes[c] > W ? W : class_genes[c]) - 1 : 0;   // add_gene_count (:66-72)
        for (uint64_t j = x.ec_offset[c]; j < x.ec_offset[c + 1]; ++j) {          // :146-150
            const size_t t = x.ec_ids[j];
            if (tx_mult) tx_mult[t * W + tslot] += num_kmer;
            if (gene_mult) gene_mult[t * W + gslot] += num_kmer;
        }
    }
    return PA_OK;
}

// Rust's `{}` of an f64: shortest digits that round-trip, never scientific
static std::string rust_f64(double v) {
    if (v != v) return "NaN";
    char buf[400];
    const auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::fixed);
    return std::string(buf, r.ptr);
}

int pa_write_mappability_tsv(const pa_host_index* h, const char* path) {
    if (!h || !path) return fail(PA_ERR_INVALID_ARG, "null argument");
    const size_t ntx = h->h.num_transcripts, W = PA_MAPPABILITY_COUNTS_LEN;
    std::vector<uint64_t> tm(ntx * W), gm(ntx * W);
    const int rc = pa_host_index_mappability(h, tm.data(), gm.data());
    if (rc != PA_OK) return rc;
    FILE* f = fopen(path, "w");
    if (!f) return fail(PA_ERR_IO, "cannot write %s", path);
    fputs("tx_name\tgene_name\ttx_kmer_count\tfrac_kmer_unique_tx\tfrac_kmer_unique_gene\n", f);   // :32-33
    for (size_t t = 0; t < ntx; ++t) {
        uint64_t total = 0;                                                       // total_kmer_count (:54-56)
        for (size_t j = 0; j < W; ++j) total += tm[t * W + j];
        const double ft = (double)tm[t * W] / (double)total, fg = (double)gm[t * W] / (double)total;   // :74-80
        fprintf(f, "%s\t%s\t%llu\t%s\t%s\n", h->h.tx_names[t].c_str(), h->h.tx_genes[t].c_str(), (unsigned long long)total,
                rust_f64(ft).c_str(), rust_f64(fg).c_str());
    }
    if (fclose(f) != 0) return fail(PA_ERR_IO, "cannot write %s", path);
    return PA_OK;
}

int pa_host_index_transcripts(const pa_host_index* h, const uint64_t** packed, const uint64_t** tx_start, uint32_t* num_tx) {
    if (!h) return fail(PA_ERR_INVALID_ARG, "null argument");
    if (packed) *packed = h->h.tx_packed.data();
    if (tx_start) *tx_start = h->h.tx_start.data();
    if (num_tx) *num_tx = (uint32_t)(h->h.tx_start.size() - 1);
    return PA_OK;
}
void pa_host_index_destroy(pa_host_index* h) { delete h; }

}  // extern "C"

// ---- own container: magic, then length-prefixed little-endian arrays (NOT the reference's bincode, utils.rs:22-43) ----
namespace {
constexpr char MAGIC[8] = {'P', 'A', 'A', 'M', 'D', 'I', 'X', '1'};
template <class T>
bool wr(FILE* f, const std::vector<T>& v) {
    const uint64_t n = v.size();
    return fwrite(&n, 8, 1, f) == 1 && (n == 0 || fwrite(v.data(), sizeof(T), n, f) == n);
}
template <class T>
bool rd(FILE* f, std::vector<T>& v) {
    uint64_t n;
    if (fread(&n, 8, 1, f) != 1 || n > (1ull << 40)) return false;
    v.resize(n);
    return n == 0 || fread(v.data(), sizeof(T), n, f) == n;
}
bool wr_strs(FILE* f, const std::vector<std::string>& v) {
    std::vector<char> blob;
    std::vector<uint64_t> off{0};
    for (auto& s : v) { blob.insert(blob.end(), s.begin(), s.end()); off.push_back(blob.size()); }
    return wr(f, off) && wr(f, blob);
}
bool rd_strs(FILE* f, std::vector<std::string>& v) {
    std::vector<char> blob;
    std::vector<uint64_t> off;
    if (!rd(f, off) || !rd(f, blob) || off.empty()) return false;
    v.clear();
    for (size_t i = 0; i + 1 < off.size(); ++i) {
        if (off[i + 1] < off[i] || off[i + 1] > blob.size()) return false;
        v.emplace_back(blob.data() + off[i], blob.data() + off[i + 1]);
    }
    return true;
}
}  // namespace

extern "C" {

int pa_host_index_save(const pa_host_index* h, const char* path) {
    if (!h || !path) return fail(PA_ERR_INVALID_ARG, "null argument");
    FILE* f = fopen(path, "wb");
    if (!f) return fail(PA_ERR_IO, "cannot create %s: %s", path, strerror(errno));
    const HostIndex& x = h->h;
    const uint32_t hdr[2] = {x.k, x.num_transcripts};
    bool ok = fwrite(MAGIC, 8, 1, f) == 1 && fwrite(hdr, 4, 2, f) == 2 && wr(f, x.node_seq) && wr(f, x.node_start) &&
              wr(f, x.node_len) && wr(f, x.node_exts) && wr(f, x.node_colour) && wr(f, x.node_redge) && wr(f, x.node_ledge) &&
              wr(f, x.ec_offset) && wr(f, x.ec_ids) && wr_strs(f, x.tx_names) && wr_strs(f, x.tx_genes) &&
              wr(f, x.tx_packed) && wr(f, x.tx_start);
    ok = (fclose(f) == 0) && ok;
    return ok ? PA_OK : fail(PA_ERR_IO, "short write to %s", path);
}

int pa_host_index_load(const char* path, pa_host_index** out) {
    if (!path || !out) return fail(PA_ERR_INVALID_ARG, "null argument");
    FILE* f = fopen(path, "rb");
    if (!f) return fail(PA_ERR_IO, "cannot open %s: %s", path, strerror(errno));
    pa_host_index* h = new pa_host_index();
    HostIndex& x = h->h;
    char magic[8];
    uint32_t hdr[2];
    bool ok = fread(magic, 8, 1, f) == 1 && memcmp(magic, MAGIC, 8) == 0 && fread(hdr, 4, 2, f) == 2;
    if (ok) {
        x.k = hdr[0];
        x.num_transcripts = hdr[1];
        ok = rd(f, x.node_seq) && rd(f, x.node_start) && rd(f, x.node_len) && rd(f, x.node_exts) && rd(f, x.node_colour) &&
             rd(f, x.node_redge) && rd(f, x.node_ledge) && rd(f, x.ec_offset) && rd(f, x.ec_ids) && rd_strs(f, x.tx_names) &&
             rd_strs(f, x.tx_genes) && rd(f, x.tx_packed) && rd(f, x.tx_start);
    }
    fclose(f);
    if (ok) {
        pa_flat_index v;
        ok = !x.node_start.empty() && !x.ec_offset.empty() && x.node_start.size() == x.node_len.size() + 1 &&
             x.node_exts.size() == x.node_len.size() && x.node_colour.size() == x.node_len.size() &&
             x.node_seq.size() >= (x.node_start.back() + 31) / 32 + 1 && x.ec_ids.size() == x.ec_offset.back();
        if (ok) { pa_host_index_view(h, &v); ok = check_flat(&v) == PA_OK; }
    }
    if (!ok) { delete h; return fail(PA_ERR_FORMAT, "%s is not a valid index container", path); }
    *out = h;
    return PA_OK;
}

}  // extern "C"

// DnaString::from_dna_string (src/pseudoaligner.rs:449-450) for a batch, into the device tile layout (host copy of
// pa_encode_kernel; used to prepare device-resident batches from the host and by the tests).
extern "C" int pa_encode_reads_host(const uint8_t* ascii, const uint64_t* offsets, uint64_t n_reads, uint32_t words_per_read,
                                    uint64_t* tiles, uint32_t* lens) {
    if (!offsets || !tiles || !lens || (n_reads && !ascii) || words_per_read == 0) return fail(PA_ERR_INVALID_ARG, "null argument");
    const uint64_t ntiles = (n_reads + 63) / 64;
    std::memset(tiles, 0, ntiles * words_per_read * 64 * sizeof(uint64_t));
    for (uint64_t i = 0; i < n_reads; ++i) {
        uint64_t len = offsets[i + 1] - offsets[i];
        if (len > 32ull * words_per_read) return fail(PA_ERR_INVALID_ARG, "read %llu longer than words_per_read allows", (unsigned long long)i);
        lens[i] = (uint32_t)len;
        const uint8_t* s = ascii + offsets[i];
        uint64_t* dst = tiles + ((i >> 6) * words_per_read) * 64 + (i & 63);
        for (uint64_t j = 0; j < len; ++j) {
            uint32_t b = base_code(s[j]);
            if (b > 3) b = 0;   // non-ACGT -> A (unpinned)
            dst[(j >> 5) * 64] |= (uint64_t)b << ((j & 31) * 2);
        }
    }
    return PA_OK;
}
