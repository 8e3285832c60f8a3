// Host driver mirroring process_reads (src/pseudoaligner.rs:420-514): FASTQ in, one Rust-Debug-formatted tuple per
// read out. The reference's per-record mutex (src/utils.rs:152-157), bounded channel (:430,:464) and serial println
// consumer (:490) are replaced by: parse a batch -> one pa_map_batch call (GPU) -> format in parallel -> write in INPUT
// order. The flag rule of :455 is kept as is (true iff coverage >= 32 and the class is EMPTY).
#include <cerrno>
#include <cstdio>
#include <thread>

#include "pa_common.hpp"

using namespace pa;

namespace {

constexpr size_t BATCH_READS = 1u << 20;

struct Batch {
    std::vector<uint8_t> seq;
    std::vector<uint64_t> off{0};
    std::vector<std::string> ids;
    void clear() { seq.clear(); off.assign(1, 0); ids.clear(); }
};

// Rust `impl Debug for str`: quotes, backslash escapes for \t \r \n \\ \" and \u{..} for other control bytes
void debug_str(std::string& out, const std::string& s) {
    out.push_back('"');
    for (unsigned char c : s) {
        switch (c) {
            case '\t': out += "\\t"; break;
            case '\r': out += "\\r"; break;
            case '\n': out += "\\n"; break;
            case '\\': out += "\\\\"; break;
            case '"': out += "\\\""; break;
            default:
                if (c < 0x20 || c == 0x7f) { char b[16]; snprintf(b, sizeof b, "\\u{%x}", c); out += b; }
                else out.push_back((char)c);
        }
    }
    out.push_back('"');
}

bool read_line(FILE* f, std::string& line) {
    line.clear();
    int c;
    bool any = false;
    while ((c = fgetc_unlocked(f)) != EOF) {
        any = true;
        if (c == '\n') break;
        line.push_back((char)c);
    }
    if (!line.empty() && line.back() == '\r') line.pop_back();
    return any;
}

}  // namespace

extern "C" int pa_process_reads(pa_index* idx, const char* fastq_path, const char* out_path, int num_threads, uint64_t* n_reads_out,
                                uint64_t* n_flagged_out) {
    if (!idx || !fastq_path || !out_path) return fail(PA_ERR_INVALID_ARG, "null argument");
    if (num_threads < 1) num_threads = 1;
    FILE* in = fopen(fastq_path, "rb");
    if (!in) return fail(PA_ERR_IO, "cannot open %s: %s", fastq_path, strerror(errno));
    FILE* out = strcmp(out_path, "-") == 0 ? stdout : fopen(out_path, "wb");
    if (!out) { fclose(in); return fail(PA_ERR_IO, "cannot create %s: %s", out_path, strerror(errno)); }
    std::vector<char> iobuf(1 << 22);
    setvbuf(in, iobuf.data(), _IOFBF, iobuf.size());

    Batch b;
    std::string l1, l2, l3, l4;
    uint64_t read_counter = 0, flagged = 0, next_report = 1000000;
    int rc = PA_OK;
    bool eof = false;
    std::vector<pa_read_result> results;
    std::vector<uint64_t> coff;
    while (!eof && rc == PA_OK) {
        b.clear();
        while (b.ids.size() < BATCH_READS) {
            if (!read_line(in, l1)) { eof = true; break; }
            if (l1.empty()) continue;
            if (l1[0] != '@' || !read_line(in, l2) || !read_line(in, l3) || l3.empty() || l3[0] != '+' || !read_line(in, l4)) {
                rc = fail(PA_ERR_FORMAT, "%s: malformed FASTQ record %llu", fastq_path, (unsigned long long)(read_counter + b.ids.size()));
                break;
            }
            const size_t sp = l1.find_first_of(" \t");
            b.ids.emplace_back(l1, 1, sp == std::string::npos ? std::string::npos : sp - 1);   // record.id() (:456)
            b.seq.insert(b.seq.end(), l2.begin(), l2.end());
            b.off.push_back(b.seq.size());
        }
        if (rc != PA_OK || b.ids.empty()) break;
        const uint64_t n = b.ids.size();
        results.resize(n);
        coff.resize(n + 1);
        const uint32_t* cids = nullptr;
        rc = pa_map_batch(idx, b.seq.data(), b.off.data(), n, PA_DEFAULT_ALLOWED_MISMATCHES, results.data(), coff.data(), &cids);   // index.map_read (:451)
        if (rc != PA_OK) break;
        std::vector<std::string> parts((size_t)num_threads);
        std::vector<uint64_t> flags((size_t)num_threads, 0);
        auto fmt = [&](int t) {
            std::string& o = parts[t];
            char num[32];
            for (uint64_t i = n * t / num_threads; i < n * (t + 1) / num_threads; ++i) {
                const pa_read_result& r = results[i];
                const bool mapped = r.mismatches & PA_MAPPED_BIT;
                const bool flag = mapped && r.coverage >= PA_READ_COVERAGE_THRESHOLD && r.class_len == 0;   // :455
                flags[t] += flag;
                o += flag ? "(true, " : "(false, ";
                debug_str(o, b.ids[i]);
                o += ", [";
                for (uint32_t j = 0; j < r.class_len; ++j) {
                    if (j) o += ", ";
                    snprintf(num, sizeof num, "%u", cids[coff[i] + j]);
                    o += num;
                }
                snprintf(num, sizeof num, "], %u)\n", mapped ? r.coverage : 0u);   // None -> (false, id, [], 0) (:461)
                o += num;
            }
        };
        if (num_threads == 1) fmt(0);
        else {
            std::vector<std::thread> th;
            for (int t = 0; t < num_threads; ++t) th.emplace_back(fmt, t);
            for (auto& t : th) t.join();
        }
        for (int t = 0; t < num_threads; ++t) {
            if (fwrite(parts[t].data(), 1, parts[t].size(), out) != parts[t].size()) { rc = fail(PA_ERR_IO, "short write to %s", out_path); break; }
            flagged += flags[t];
        }
        read_counter += n;
        while (read_counter >= next_report) {   // :497-503
            fprintf(stderr, "\rDone Mapping %llu reads w/ Rate: %g", (unsigned long long)next_report,
                    (double)((float)flagged * 100.0f / (float)read_counter));
            next_report += 1000000;
        }
    }
    fclose(in);
    if (out != stdout) { if (fclose(out) != 0 && rc == PA_OK) rc = fail(PA_ERR_IO, "close %s: %s", out_path, strerror(errno)); }
    else fflush(stdout);
    if (n_reads_out) *n_reads_out = read_counter;
    if (n_flagged_out) *n_flagged_out = flagged;
    return rc;
}
