/* A plain-C client of include/pseudoaligner_amd.h that calls EVERY entry point of the header once, with the argument types a
 * foreign binding (integration/rust/src/amd_ffi.rs, rust-pseudoaligner_amd/_ffi.py) assumes. Compiled by __graft_entry__.build()
 * with gcc -Wall -Werror against the header, so a drifting prototype is a build error, not a run-time surprise; run by the
 * tests: without a GPU the host half runs and every device entry point must fail with PA_ERR_NO_DEVICE (no CPU fallback),
 * with a GPU the whole sequence runs on a toy transcriptome.
 *   usage: abi_check <fasta> <fastq> <scratch dir>          exit code 0 = every call behaved
 */
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pseudoaligner_amd.h"

static int failures = 0;
#define EXPECT(cond)                                                                      \
    do {                                                                                  \
        if (!(cond)) { fprintf(stderr, "abi_check: %s:%d: %s (last error: %s)\n", __FILE__, __LINE__, #cond, pa_last_error()); ++failures; } \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s <fasta> <fastq> <scratch dir>\n", argv[0]); return 2; }
    const char *fasta = argv[1], *fastq = argv[2], *dir = argv[3];
    char path[1024], report[256];

    /* the struct layouts as THIS compiler sees them, for the test that checks the #[repr(C)] structs of integration/rust/src/amd_ffi.rs
     * against them (tests/test_abi.py): "layout <struct> sizeof <n>" and "layout <struct>.<field> <offset>" */
#define LAYOUT_STRUCT(T) printf("layout %s sizeof %zu\n", #T, sizeof(T))
#define LAYOUT_FIELD(T, f) printf("layout %s.%s %zu\n", #T, #f, offsetof(T, f))
    LAYOUT_STRUCT(pa_flat_index);
    LAYOUT_FIELD(pa_flat_index, k); LAYOUT_FIELD(pa_flat_index, num_nodes); LAYOUT_FIELD(pa_flat_index, num_classes); LAYOUT_FIELD(pa_flat_index, num_transcripts);
    LAYOUT_FIELD(pa_flat_index, seq_bases); LAYOUT_FIELD(pa_flat_index, node_seq); LAYOUT_FIELD(pa_flat_index, node_start); LAYOUT_FIELD(pa_flat_index, node_len);
    LAYOUT_FIELD(pa_flat_index, node_exts); LAYOUT_FIELD(pa_flat_index, node_colour); LAYOUT_FIELD(pa_flat_index, ec_offset); LAYOUT_FIELD(pa_flat_index, ec_ids);
    LAYOUT_FIELD(pa_flat_index, node_redge); LAYOUT_FIELD(pa_flat_index, node_ledge);
    LAYOUT_STRUCT(pa_read_result);
    LAYOUT_FIELD(pa_read_result, coverage); LAYOUT_FIELD(pa_read_result, mismatches); LAYOUT_FIELD(pa_read_result, class_off); LAYOUT_FIELD(pa_read_result, class_len);
    LAYOUT_STRUCT(pa_index_stats);
    LAYOUT_FIELD(pa_index_stats, num_kmers); LAYOUT_FIELD(pa_index_stats, table_slots); LAYOUT_FIELD(pa_index_stats, bytes_table); LAYOUT_FIELD(pa_index_stats, bytes_graph);
    LAYOUT_FIELD(pa_index_stats, bytes_classes); LAYOUT_FIELD(pa_index_stats, bytes_total); LAYOUT_FIELD(pa_index_stats, num_nodes); LAYOUT_FIELD(pa_index_stats, num_classes);
    LAYOUT_FIELD(pa_index_stats, k); LAYOUT_FIELD(pa_index_stats, max_class_len);

    /* ---- host half ---- */
    EXPECT(pa_abi_version() == PA_ABI_VERSION);
    EXPECT(pa_last_error() != NULL);
    const int ndev = pa_device_count();
    pa_host_index *h = NULL, *h2 = NULL, *h3 = NULL, *h4 = NULL;
    EXPECT(pa_host_index_build_fasta(fasta, 20, 2, &h) == PA_OK && h);
    if (!h) return 1;
    pa_flat_index flat;
    EXPECT(pa_host_index_view(h, &flat) == PA_OK && flat.k == 20 && flat.num_nodes > 0);
    EXPECT(pa_host_index_from_flat(&flat, &h2) == PA_OK);
    EXPECT(pa_host_index_compare(h, h2, 1u << 20, report, sizeof report) == 0);
    snprintf(path, sizeof path, "%s/abi_check.idx", dir);
    EXPECT(pa_host_index_save(h, path) == PA_OK);
    EXPECT(pa_host_index_load(path, &h3) == PA_OK);
    const uint32_t ntx = pa_host_index_num_transcripts(h);
    EXPECT(ntx == flat.num_transcripts && pa_host_index_tx_name(h, 0) != NULL && pa_host_index_tx_gene(h, 0) != NULL);
    uint32_t ngenes = 0;
    uint32_t* tx_gene = (uint32_t*)calloc(ntx ? ntx : 1, 4);
    EXPECT(pa_host_index_genes(h, tx_gene, &ngenes) == PA_OK && ngenes >= 1 && pa_host_index_gene_name(h, 0) != NULL);
    const uint64_t counts_len = (uint64_t)flat.num_classes + 3;
    uint64_t* counts = (uint64_t*)calloc(counts_len, 8);
    uint64_t* gene_counts = (uint64_t*)calloc(ngenes + 1, 8);
    counts[0] = 5;
    EXPECT(pa_counts_collapse_genes(h, counts, counts_len, gene_counts) == PA_OK);
    uint64_t* mult = (uint64_t*)calloc((size_t)ntx * PA_MAPPABILITY_COUNTS_LEN, 8);
    EXPECT(pa_host_index_mappability(h, mult, NULL) == PA_OK);
    snprintf(path, sizeof path, "%s/abi_check_mappability.tsv", dir);
    EXPECT(pa_write_mappability_tsv(h, path) == PA_OK);
    const uint64_t *packed = NULL, *tx_start = NULL;
    uint32_t ntx2 = 0;
    EXPECT(pa_host_index_transcripts(h, &packed, &tx_start, &ntx2) == PA_OK && ntx2 == ntx);
    EXPECT(pa_host_index_build_packed(packed, tx_start, ntx2, 20, 1, &h4) == PA_OK);
    EXPECT(pa_host_index_compare(h, h4, 1u << 20, report, sizeof report) == 0);

    const uint8_t ascii[] = "ACGTACGTACGTTTGACCAGTNNAC";
    const uint64_t offsets[3] = {0, 12, 25};
    const uint32_t wpr = pa_words_per_read(13);
    EXPECT(wpr == 1 && pa_tiles_words(2, wpr) == 64);
    uint64_t tiles[64];
    uint32_t lens[2];
    EXPECT(pa_encode_reads_host(ascii, offsets, 2, wpr, tiles, lens) == PA_OK && lens[0] == 12 && lens[1] == 13);
    {   /* the scan stage of process_reads, no GPU needed */
        uint64_t nrec = 0, starts[4];
        uint32_t hdr[4], sq[4];
        int kind = -1;
        EXPECT(pa_fastq_scan_host(fastq, 2, &nrec, starts, hdr, sq, 4, &kind) == PA_OK && nrec > 4 && kind == 0 && starts[0] == 0 && sq[0] > 0 &&
               starts[1] > starts[0] + hdr[0] + sq[0]);
    }

    pa_txome *tx = NULL, *tx2 = NULL, *tx3 = NULL;
    EXPECT(pa_txome_synthesize(50, 120, 7, &tx) == PA_OK);
    {   /* the same genes with repeat families and low-complexity tracts in their last exons */
        pa_txome* txr = NULL;
        const pa_synth_repeats rep = {4, 100, 100000, 150000, 1, 30000, 60000, 500000, 3};
        uint32_t ntx_plain = 0, ntx_rep = 0;
        const uint64_t *st_plain = NULL, *st_rep = NULL;
        EXPECT(pa_txome_synthesize_repeats(50, 120, 7, &rep, &txr) == PA_OK);
        EXPECT(pa_txome_view(tx, NULL, &st_plain, &ntx_plain) == PA_OK && pa_txome_view(txr, NULL, &st_rep, &ntx_rep) == PA_OK);
        EXPECT(ntx_plain == ntx_rep && st_rep[ntx_rep] > st_plain[ntx_plain]);
        pa_txome_destroy(txr);
    }
    EXPECT(pa_txome_from_host_index(h, &tx2) == PA_OK);
    EXPECT(pa_txome_from_fasta(fasta, &tx3) == PA_OK);
    EXPECT(pa_txome_view(tx2, &packed, &tx_start, &ntx2) == PA_OK && ntx2 == ntx);
    const uint64_t nsim = 256;
    const uint32_t sim_wpr = pa_words_per_read(60);
    uint64_t* sim_tiles = (uint64_t*)calloc(pa_tiles_words(nsim, sim_wpr), 8);
    uint32_t* sim_lens = (uint32_t*)calloc(nsim, 4);
    EXPECT(pa_simulate_reads_host(tx2, 60, 1, 10000, 0, nsim, sim_wpr, sim_tiles, sim_lens) == PA_OK && sim_lens[0] == 60);

    const uint32_t ovf_a[] = {1, 7, 2, 3, 0, 4, 9}, ovf_b[] = {1, 7, 2, 1, 0, 4, 9};
    const uint32_t* bufs[2] = {ovf_a, ovf_b};
    const uint64_t nwords[2] = {7, 7};
    uint32_t merged[16];
    uint64_t merged_words = 0;
    EXPECT(pa_overflow_merge(bufs, nwords, 2, merged, 16, &merged_words) == PA_OK && merged_words == 7 && merged[3] == 4);

    /* ---- device half: with a GPU it runs, without one every entry point refuses (there is no CPU fallback) ---- */
    pa_index* idx = NULL;
    int rc = pa_index_create(&flat, 0, &idx);
    if (ndev < 1) {
        void *p = NULL, *ev = NULL;
        pa_overflow* o = NULL;
        pa_txome_device* td = NULL;
        EXPECT(rc == PA_ERR_NO_DEVICE && idx == NULL);
        EXPECT(pa_device_malloc(0, 64, &p) == PA_ERR_NO_DEVICE);
        EXPECT(pa_overflow_create(0, 16, 64, &o) == PA_ERR_NO_DEVICE);
        EXPECT(pa_txome_upload(tx2, 60, 0, &td) == PA_ERR_NO_DEVICE);
        EXPECT(pa_event_create(&ev) < 0);
        const int devs0[1] = {0};
        pa_index* multi0[1] = {NULL};
        EXPECT(pa_index_create_multi(&flat, devs0, 1, multi0) == PA_ERR_NO_DEVICE && multi0[0] == NULL);
        pa_host_index* hg = NULL;
        EXPECT(pa_host_index_build_fasta_device(fasta, 20, 0, &hg) == PA_ERR_NO_DEVICE && hg == NULL);
        EXPECT(pa_host_index_build_packed_device(packed, tx_start, ntx2, 20, 0, &hg) == PA_ERR_NO_DEVICE && hg == NULL);
        printf("abi_check: host half ok, no device: %d failures\n", failures);
    } else {
        EXPECT(rc == PA_OK && idx);
        /* one call for the GPUs of a single-process host (here: the one GPU, and a refused duplicate) */
        const int devs[2] = {0, 0};
        pa_index* multi[2] = {NULL, NULL};
        EXPECT(pa_index_create_multi(&flat, devs, 1, multi) == PA_OK && multi[0] != NULL);
        pa_index_stats mst;
        EXPECT(pa_index_get_stats(multi[0], &mst) == PA_OK && mst.num_nodes == flat.num_nodes);
        pa_index_destroy(multi[0]);
        multi[0] = NULL;
        EXPECT(pa_index_create_multi(&flat, devs, 2, multi) == PA_ERR_INVALID_ARG && multi[0] == NULL && multi[1] == NULL);
        /* the graph built on the GPU is the graph the CPU builder gives */
        pa_host_index *hg = NULL, *hg2 = NULL;
        EXPECT(pa_host_index_build_fasta_device(fasta, 20, 0, &hg) == PA_OK && hg);
        EXPECT(pa_host_index_compare(h, hg, 1u << 20, report, sizeof report) == 0);
        EXPECT(pa_host_index_build_packed_device(packed, tx_start, ntx2, 20, 0, &hg2) == PA_OK && hg2);
        EXPECT(pa_host_index_compare(h, hg2, 1u << 20, report, sizeof report) == 0);
        pa_host_index_destroy(hg);
        pa_host_index_destroy(hg2);
        pa_index_stats st;
        EXPECT(pa_index_get_stats(idx, &st) == PA_OK && st.k == 20 && st.num_nodes == flat.num_nodes);
        EXPECT(pa_counts_len(idx) == counts_len);
        /* single reads and host batches */
        const char* ex1 = "GGCTGTCAACCAGTCCATAGGCAGGGCCATCAGGCACCAAAGGGATTCTGCCAGCATAGT";
        uint32_t cls[64], ncls = 0, cov = 0, mm = 0, nodes[256], nn = 0;
        rc = pa_map_read(idx, (const uint8_t*)ex1, 60, cls, 64, &ncls, &cov);
        EXPECT(rc == 0 || (rc == 1 && cov <= 60));
        EXPECT(pa_map_read_with_mismatch(idx, (const uint8_t*)ex1, 60, 2, cls, 64, &ncls, &cov, &mm) >= 0);
        EXPECT(pa_map_read_to_nodes(idx, (const uint8_t*)ex1, 60, 2, nodes, 256, &nn, &cov, &mm) >= 0);
        pa_read_result res[2];
        uint64_t coff[3];
        const uint32_t* cids = NULL;
        EXPECT(pa_map_batch(idx, ascii, offsets, 2, 2, res, coff, &cids) == PA_OK);
        /* the same two reads as a DnaString would hold them: 2-bit words (LSB-first: the host encoder's tiles of reads 0 and 1) */
        {
            uint64_t pw[8];
            uint32_t plen[2] = {lens[0], lens[1]};
            uint64_t poff[3] = {0, wpr, 2ull * wpr};
            for (uint32_t w = 0; w < wpr && w < 4; ++w) { pw[w] = tiles[(uint64_t)w * 64]; pw[wpr + w] = tiles[(uint64_t)w * 64 + 1]; }
            pa_read_result pres[2];
            uint64_t pcoff[3];
            const uint32_t* pcids = NULL;
            EXPECT(pa_map_batch_packed(idx, pw, poff, plen, 2, PA_PACKED_LSB_FIRST, 2, pres, pcoff, &pcids) == PA_OK);
            EXPECT(pres[0].coverage == res[0].coverage && pres[1].coverage == res[1].coverage && pcoff[2] == coff[2]);
            uint32_t pcls[64], pn = 0, pcov = 0, pmm = 0;
            EXPECT(pa_map_read_packed(idx, pw, plen[0], PA_PACKED_LSB_FIRST, 2, pcls, 64, &pn, &pcov, &pmm) >= 0 && pcov == res[0].coverage);
        }
        /* process_reads for a caller that holds the reader: push records, pull the tuples */
        {
            pa_record_stream* rs = NULL;
            EXPECT(pa_record_stream_create(idx, 2, 64, &rs) == PA_OK && rs);
            const uint8_t ids2[] = "r0r1";
            const uint64_t ioff[3] = {0, 2, 4};
            EXPECT(pa_records_push(rs, ids2, ioff, ascii, offsets, 2) == PA_OK);
            char text[512];
            size_t nb = 1;
            EXPECT(pa_records_pull(rs, text, sizeof text, &nb) == PA_OK && nb == 0);   /* nothing rendered yet: the batch is still being filled */
            EXPECT(pa_records_flush(rs) == PA_OK);
            EXPECT(pa_records_pull(rs, text, sizeof text, &nb) == PA_OK && nb > 0 && text[nb - 1] == '\n' && text[0] == '(');
            uint64_t rn = 0, rf = 0;
            EXPECT(pa_record_stream_stats(rs, &rn, &rf) == PA_OK && rn == 2);
            { double st[PA_INGEST_STAGES]; EXPECT(pa_record_stream_stage_seconds(rs, st) == PA_OK && st[7] == 2.0 && st[6] >= 0.0);
              EXPECT(pa_process_reads_stage_seconds(st) == PA_OK); }
            pa_record_stream_destroy(rs);
            {   /* the same stream over two lanes of the handle: the same tuples */
                pa_index* two[2]; char text2[4096]; size_t nb2 = 0;
                two[0] = idx; two[1] = idx;
                EXPECT(pa_record_stream_create_multi(two, 2, 2, 64, &rs) == PA_OK && rs);
                EXPECT(pa_records_push(rs, ids2, ioff, ascii, offsets, 2) == PA_OK && pa_records_flush(rs) == PA_OK);
                EXPECT(pa_records_pull(rs, text2, sizeof text2, &nb2) == PA_OK && nb2 == nb && memcmp(text, text2, nb) == 0);
                pa_record_stream_destroy(rs);
            }
        }
        uint32_t nodes_flat[2 * 64], nodes_len[2];
        EXPECT(pa_map_batch_nodes(idx, ascii, offsets, 2, 2, res, nodes_flat, 64, nodes_len) == PA_OK);
        snprintf(path, sizeof path, "%s/abi_check_tuples.txt", dir);
        uint64_t nreads = 0, nflag = 0;
        EXPECT(pa_process_reads(idx, fastq, path, 2, &nreads, &nflag) == PA_OK && nreads > 0);
        { pa_index* two[2]; uint64_t n2 = 0, f2 = 0; two[0] = idx; two[1] = idx;   /* two lanes on one GPU */
          EXPECT(pa_process_reads_multi(two, 2, fastq, path, 2, &n2, &f2) == PA_OK && n2 == nreads && f2 == nflag); }
        /* device-resident batch: simulate on the device, map with the fused count table, overflow table, RCCL world of one */
        pa_txome_device* td = NULL;
        EXPECT(pa_txome_upload(tx2, 60, 0, &td) == PA_OK);
        void *d_tiles = NULL, *d_lens = NULL, *d_res = NULL, *d_arena = NULL, *d_counts = NULL, *d_colour = NULL, *d_ascii = NULL, *d_off = NULL;
        const uint64_t arena_cap = pa_map_arena_hint(idx, nsim);
        EXPECT(pa_device_malloc(0, pa_tiles_words(nsim, sim_wpr) * 8, &d_tiles) == PA_OK);
        EXPECT(pa_device_malloc(0, nsim * 4, &d_lens) == PA_OK && pa_device_malloc(0, nsim * sizeof(pa_read_result), &d_res) == PA_OK);
        EXPECT(pa_device_malloc(0, arena_cap * 4, &d_arena) == PA_OK && pa_device_malloc(0, counts_len * 8, &d_counts) == PA_OK);
        EXPECT(pa_device_malloc(0, nsim * 4, &d_colour) == PA_OK && pa_device_malloc(0, 64, &d_ascii) == PA_OK && pa_device_malloc(0, 64, &d_off) == PA_OK);
        EXPECT(pa_memset_device(d_counts, 0, counts_len * 8, NULL) == PA_OK);
        void *ev0 = NULL, *ev1 = NULL;
        float ms = -1.0f;
        EXPECT(pa_event_create(&ev0) == PA_OK && pa_event_create(&ev1) == PA_OK);
        EXPECT(pa_simulate_reads_device(td, 1, 10000, 0, nsim, sim_wpr, (uint64_t*)d_tiles, (uint32_t*)d_lens, NULL) == PA_OK);
        pa_overflow* ovf = NULL;
        EXPECT(pa_overflow_create(0, 1024, 1 << 16, &ovf) == PA_OK && pa_index_set_overflow(idx, ovf) == PA_OK);
        EXPECT(pa_event_record(ev0, NULL) == PA_OK);
        EXPECT(pa_index_set_timing(idx, 1) == PA_OK);
        EXPECT(pa_map_count_batch_device(idx, (const uint64_t*)d_tiles, (const uint32_t*)d_lens, nsim, sim_wpr, 2, (pa_read_result*)d_res,
                                         (uint32_t*)d_arena, arena_cap, (uint64_t*)d_counts, NULL) == PA_OK);
        EXPECT(pa_event_record(ev1, NULL) == PA_OK);
        uint64_t used = 0, need = 0;
        EXPECT(pa_map_finish(idx, NULL, &used, &need) == PA_OK);
        { float kms = -1.0f, st[3] = {-1.0f, -1.0f, -1.0f}; EXPECT(pa_map_kernel_ms(idx, NULL, &kms) == PA_OK && kms > 0.0f);
          EXPECT(pa_map_stage_ms(idx, NULL, st) == PA_OK && st[0] > 0.0f && st[1] >= 0.0f && st[2] >= 0.0f); EXPECT(pa_index_set_timing(idx, 0) == PA_OK); }
        {   /* the records in their 8-byte form + the packed stream of the classes that are no index classes */
            void *d_compact = NULL, *d_packed = NULL, *d_pw = NULL, *d_scr = NULL;
            const size_t scr = pa_compact_scratch_bytes(nsim);
            uint64_t* hc = (uint64_t*)malloc(nsim * 8);
            pa_read_result* hr = (pa_read_result*)malloc(nsim * sizeof(pa_read_result));
            uint64_t pw = ~0ull;
            EXPECT(pa_device_malloc(0, nsim * 8, &d_compact) == PA_OK && pa_device_malloc(0, (arena_cap + 16) * 4, &d_packed) == PA_OK &&
                   pa_device_malloc(0, 8, &d_pw) == PA_OK && pa_device_malloc(0, scr, &d_scr) == PA_OK);
            EXPECT(pa_results_compact_device(idx, (const pa_read_result*)d_res, (const uint32_t*)d_arena, arena_cap, nsim, (uint64_t*)d_compact, (uint32_t*)d_packed,
                                             arena_cap + 16, (uint64_t*)d_pw, d_scr, scr, NULL) == PA_OK);
            EXPECT(pa_memcpy_d2h(hc, d_compact, nsim * 8, NULL) == PA_OK && pa_memcpy_d2h(&pw, d_pw, 8, NULL) == PA_OK &&
                   pa_memcpy_d2h(hr, d_res, nsim * sizeof(pa_read_result), NULL) == PA_OK && pa_stream_synchronize(NULL) == PA_OK);
            for (uint64_t i = 0; i < nsim; ++i) {
                const uint32_t lo = (uint32_t)hc[i];
                EXPECT((lo & 0x3FFFu) == hr[i].coverage && ((lo >> 14) & 0x3FFFu) == (hr[i].mismatches & 0x3FFFu) &&
                       ((lo & PA_COMPACT_MAPPED) != 0) == ((hr[i].mismatches & PA_MAPPED_BIT) != 0));
                if (lo & PA_COMPACT_BY_REF) EXPECT((uint32_t)(hc[i] >> 32) == (hr[i].class_off & ~PA_CLASS_REF) && (hr[i].class_off & PA_CLASS_REF));
            }
            EXPECT(pw <= arena_cap + 16);
            free(hc); free(hr);
            pa_device_free(d_compact); pa_device_free(d_packed); pa_device_free(d_pw); pa_device_free(d_scr);
        }
        {   /* host to host: the simulated batch from pinned host tiles to compact records + count table, chunks of 64 reads on two streams */
            void *ph_tiles = NULL, *ph_compact = NULL, *ph_packed = NULL, *ph_counts = NULL;
            const uint64_t clen = pa_counts_len(idx);
            uint64_t pwords = ~0ull, total = 0;
            EXPECT(pa_host_alloc_pinned(pa_tiles_words(nsim, sim_wpr) * 8, &ph_tiles) == PA_OK && pa_host_alloc_pinned(nsim * 8, &ph_compact) == PA_OK &&
                   pa_host_alloc_pinned((arena_cap + 16) * 4, &ph_packed) == PA_OK && pa_host_alloc_pinned(clen * 8, &ph_counts) == PA_OK);
            memcpy(ph_tiles, sim_tiles, pa_tiles_words(nsim, sim_wpr) * 8);
            EXPECT(pa_map_tiles_host(idx, (const uint64_t*)ph_tiles, sim_lens, 0, nsim, sim_wpr, 2, (uint64_t*)ph_compact, (uint32_t*)ph_packed, arena_cap + 16, &pwords,
                                     (uint64_t*)ph_counts, 64, 2) == PA_OK);
            for (uint64_t i = 0; i < clen; ++i) total += ((uint64_t*)ph_counts)[i];
            EXPECT(total == nsim && pwords <= arena_cap + 16);
            EXPECT(pa_map_tiles_host(idx, (const uint64_t*)ph_tiles, NULL, 60, nsim, sim_wpr, 2, (uint64_t*)ph_compact, (uint32_t*)ph_packed, arena_cap + 16, &pwords, NULL, 0, 0) == PA_OK);
            EXPECT(pa_host_free_pinned(ph_tiles) == PA_OK && pa_host_free_pinned(ph_compact) == PA_OK && pa_host_free_pinned(ph_packed) == PA_OK && pa_host_free_pinned(ph_counts) == PA_OK);
        }
        {   /* the same batch as a uniform one (the simulator's reads all have 60 bases): the same records without a length array */
            void* d_res2 = NULL;
            pa_read_result *r1 = (pa_read_result*)malloc(nsim * sizeof(pa_read_result)), *r2 = (pa_read_result*)malloc(nsim * sizeof(pa_read_result));
            EXPECT(pa_device_malloc(0, nsim * sizeof(pa_read_result), &d_res2) == PA_OK);
            EXPECT(pa_map_count_batch_uniform_device(idx, (const uint64_t*)d_tiles, 0, nsim, sim_wpr, 2, (pa_read_result*)d_res2, (uint32_t*)d_arena, arena_cap,
                                                     (uint64_t*)d_counts, NULL) == PA_ERR_INVALID_ARG);
            EXPECT(pa_memcpy_d2h(r1, d_res, nsim * sizeof(pa_read_result), NULL) == PA_OK);
            EXPECT(pa_map_count_batch_uniform_device(idx, (const uint64_t*)d_tiles, 60, nsim, sim_wpr, 2, (pa_read_result*)d_res2, (uint32_t*)d_arena, arena_cap,
                                                     (uint64_t*)d_counts, NULL) == PA_OK);
            EXPECT(pa_map_finish(idx, NULL, &used, &need) == PA_OK);
            EXPECT(pa_memcpy_d2h(r2, d_res2, nsim * sizeof(pa_read_result), NULL) == PA_OK && pa_stream_synchronize(NULL) == PA_OK);
            uint64_t same = 0;
            for (uint64_t i = 0; i < nsim; ++i)
                same += r1[i].coverage == r2[i].coverage && r1[i].mismatches == r2[i].mismatches && r1[i].class_len == r2[i].class_len &&
                        ((r1[i].class_off & PA_CLASS_REF) ? r1[i].class_off == r2[i].class_off : !(r2[i].class_off & PA_CLASS_REF));
            EXPECT(same == nsim);
            EXPECT(pa_device_free(d_res2) == PA_OK);
            free(r1); free(r2);
        }
        EXPECT(pa_index_release_stream(idx, NULL) == PA_OK);   /* the null stream's launch context goes; the next launch makes a new one */
        EXPECT(pa_event_elapsed_ms(ev0, ev1, &ms) == PA_OK && ms >= 0.0f);
        EXPECT(pa_map_batch_device(idx, (const uint64_t*)d_tiles, (const uint32_t*)d_lens, nsim, sim_wpr, 2, (pa_read_result*)d_res, (uint32_t*)d_arena,
                                   arena_cap, (uint32_t*)d_colour, NULL) == PA_OK);
        EXPECT(pa_map_finish(idx, NULL, &used, &need) == PA_OK);
        EXPECT(pa_counts_accumulate_device(idx, (const pa_read_result*)d_res, (const uint32_t*)d_arena, (const uint32_t*)d_colour, nsim, (uint64_t*)d_counts, NULL) == PA_OK);
        EXPECT(pa_stream_synchronize(NULL) == PA_OK);
        {   /* per-barcode counts: four cells' worth of barcodes over the same records */
            void *d_bc = NULL, *d_keys = NULL, *d_vals = NULL;
            uint32_t* bc = (uint32_t*)calloc(nsim, 4);
            for (uint64_t i = 0; i < nsim; ++i) bc[i] = (uint32_t)(i & 3);
            uint64_t cells = 0;
            EXPECT(pa_device_malloc(0, nsim * 4, &d_bc) == PA_OK && pa_device_malloc(0, nsim * 8, &d_keys) == PA_OK && pa_device_malloc(0, nsim * 4, &d_vals) == PA_OK);
            EXPECT(pa_memcpy_h2d(d_bc, bc, nsim * 4, NULL) == PA_OK);
            EXPECT(pa_counts_by_barcode_device(idx, (const pa_read_result*)d_res, (const uint32_t*)d_arena, (const uint32_t*)d_bc, nsim, 2, (uint64_t*)d_keys,
                                               (uint32_t*)d_vals, &cells, NULL) == PA_OK && cells >= 4 && cells <= nsim);
            EXPECT(pa_device_free(d_bc) == PA_OK && pa_device_free(d_keys) == PA_OK && pa_device_free(d_vals) == PA_OK);
            free(bc);
        }
        uint64_t* h_counts = (uint64_t*)calloc(counts_len, 8);
        EXPECT(pa_memcpy_d2h(h_counts, d_counts, counts_len * 8, NULL) == PA_OK);
        uint64_t total = 0;
        for (uint64_t i = 0; i < counts_len; ++i) total += h_counts[i];
        EXPECT(total == 3 * nsim);                                   /* counted by the two fused launches (ragged form, uniform form) and once by the count kernel */
        const uint32_t* words = NULL;
        uint64_t nw = 0;
        EXPECT(pa_overflow_fetch(ovf, NULL, &words, &nw) == PA_OK && nw >= 2 && words[1] == nw);
        uint8_t id[128];
        pa_comm* comm = NULL;
        if (pa_comm_unique_id(id) == PA_OK) {                        /* RCCL present: a world of one */
            EXPECT(pa_comm_create(0, 1, 0, id, &comm) == PA_OK && pa_comm_rank(comm) == 0 && pa_comm_size(comm) == 1);
            EXPECT(pa_counts_allreduce(idx, (uint64_t*)d_counts, comm, NULL) == PA_OK);
            EXPECT(pa_overflow_allgather(ovf, comm, NULL, &words, &nw) == PA_OK && words[1] == nw);
            pa_comm_destroy(comm);
        } else {
            EXPECT(pa_counts_allreduce(idx, (uint64_t*)d_counts, NULL, NULL) == PA_OK);
            EXPECT(pa_overflow_allgather(ovf, NULL, NULL, &words, &nw) == PA_OK);
            EXPECT(pa_comm_rank(NULL) == 0 && pa_comm_size(NULL) == 1);
        }
        EXPECT(pa_overflow_reset(ovf, NULL) == PA_OK && pa_index_set_overflow(idx, NULL) == PA_OK);
        pa_overflow_destroy(ovf);
        /* the encode kernel */
        EXPECT(pa_memcpy_h2d(d_ascii, ascii, 25, NULL) == PA_OK && pa_memcpy_h2d(d_off, offsets, 24, NULL) == PA_OK);
        EXPECT(pa_encode_reads_device(idx, (const uint8_t*)d_ascii, (const uint64_t*)d_off, 2, wpr, (uint64_t*)d_tiles, (uint32_t*)d_lens, NULL) == PA_OK);
        uint64_t t2[64];
        EXPECT(pa_memcpy_d2h(t2, d_tiles, sizeof t2, NULL) == PA_OK && memcmp(t2, tiles, sizeof t2) == 0);
        EXPECT(pa_event_destroy(ev0) == PA_OK && pa_event_destroy(ev1) == PA_OK);
        void* bufs_d[] = {d_tiles, d_lens, d_res, d_arena, d_counts, d_colour, d_ascii, d_off};
        for (size_t i = 0; i < sizeof bufs_d / sizeof *bufs_d; ++i) EXPECT(pa_device_free(bufs_d[i]) == PA_OK);
        pa_txome_device_destroy(td);
        pa_index_destroy(idx);
        free(h_counts);
        printf("abi_check: host and device halves ok on %d device(s): %d failures\n", ndev, failures);
    }
    pa_txome_destroy(tx);
    pa_txome_destroy(tx2);
    pa_txome_destroy(tx3);
    pa_host_index_destroy(h);
    pa_host_index_destroy(h2);
    pa_host_index_destroy(h3);
    pa_host_index_destroy(h4);
    free(tx_gene); free(counts); free(gene_counts); free(mult); free(sim_tiles); free(sim_lens);
    return failures ? 1 : 0;
}
