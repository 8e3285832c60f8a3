//! `extern "C"` binding of include/pseudoaligner_amd.h — the part of the C ABI a Rust host needs for
//! `Pseudoaligner::map_read` / `process_reads` (src/pseudoaligner.rs:381-384, :420-514), the fused class-count table and the
//! reduction over GPUs. Field order and widths mirror the header exactly; `integration/c/abi_check.c` compiles the same
//! declarations from C against the header (this file cannot be compiled in the image of this repository: no rustc).
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int, c_void};

pub const PA_OK: c_int = 0;
pub const PA_ERR_ARENA_FULL: c_int = -7;
pub const PA_ERR_BUFFER_TOO_SMALL: c_int = -10;
pub const PA_MAPPED_BIT: u32 = 0x8000_0000;
pub const PA_CLASS_REF: u32 = 0x8000_0000;
pub const PA_MAX_ARENA_ENTRIES: u64 = 0x7FFF_FFFF;
pub const PA_COMPACT_MAPPED: u32 = 0x1000_0000;     // pa_results_compact_device: bit 28 of a record's low word
pub const PA_COMPACT_BY_REF: u32 = 0x2000_0000;     // the class is index class (record >> 32)
pub const PA_COMPACT_PACKED: u32 = 0x4000_0000;     // the class is the next {length, ids...} entry of the packed stream

#[repr(C)]
pub struct PaFlatIndex {            // pa_flat_index
    pub k: u32, pub num_nodes: u32, pub num_classes: u32, pub num_transcripts: u32,
    pub seq_bases: u64,
    pub node_seq: *const u64, pub node_start: *const u64, pub node_len: *const u32,
    pub node_exts: *const u8, pub node_colour: *const u32,
    pub ec_offset: *const u64, pub ec_ids: *const u32,
    pub node_redge: *const u32, pub node_ledge: *const u32,   // may be null: the library derives the edges
}

#[repr(C)]
#[derive(Clone, Copy, Default, Debug)]
pub struct PaReadResult { pub coverage: u32, pub mismatches: u32, pub class_off: u32, pub class_len: u32 }

#[repr(C)]
#[derive(Clone, Copy, Default, Debug)]
pub struct PaIndexStats {           // pa_index_stats
    pub num_kmers: u64, pub table_slots: u64,
    pub bytes_table: u64, pub bytes_graph: u64, pub bytes_classes: u64, pub bytes_total: u64,
    pub num_nodes: u32, pub num_classes: u32, pub k: u32, pub max_class_len: u32,
}

#[repr(C)] pub struct PaIndex { _private: [u8; 0] }
#[repr(C)] pub struct PaHostIndex { _private: [u8; 0] }
#[repr(C)] pub struct PaOverflow { _private: [u8; 0] }
#[repr(C)] pub struct PaComm { _private: [u8; 0] }
#[repr(C)] pub struct PaRecordStream { _private: [u8; 0] }

pub const PA_PACKED_LSB_FIRST: c_int = 0;   // base j in bits 2 (j % 32) of word j / 32
pub const PA_PACKED_MSB_FIRST: c_int = 1;

extern "C" {
    pub fn pa_abi_version() -> u32;
    pub fn pa_last_error() -> *const c_char;
    pub fn pa_device_count() -> c_int;

    // index: flat form of `pub struct Pseudoaligner<K>` (src/pseudoaligner.rs:26-33) -> GPU
    pub fn pa_index_create(flat: *const PaFlatIndex, device: c_int, out: *mut *mut PaIndex) -> c_int;
    pub fn pa_index_create_multi(flat: *const PaFlatIndex, devices: *const c_int, ndev: c_int, out: *mut *mut PaIndex) -> c_int;
    pub fn pa_index_get_stats(idx: *const PaIndex, stats: *mut PaIndexStats) -> c_int;
    pub fn pa_index_destroy(idx: *mut PaIndex);
    pub fn pa_host_index_from_flat(flat: *const PaFlatIndex, out: *mut *mut PaHostIndex) -> c_int;
    pub fn pa_host_index_build_fasta(fasta_path: *const c_char, k: u32, num_threads: c_int, out: *mut *mut PaHostIndex) -> c_int;
    pub fn pa_host_index_build_fasta_device(fasta_path: *const c_char, k: u32, device: c_int, out: *mut *mut PaHostIndex) -> c_int;
    pub fn pa_host_index_save(h: *const PaHostIndex, path: *const c_char) -> c_int;
    pub fn pa_host_index_compare(a: *const PaHostIndex, b: *const PaHostIndex, max_kmers: u64, report: *mut c_char, report_cap: usize) -> c_int;
    pub fn pa_host_index_destroy(h: *mut PaHostIndex);

    // map_read / process_reads
    pub fn pa_map_batch(idx: *mut PaIndex, ascii: *const u8, offsets: *const u64, n_reads: u64, allowed_mismatches: u32,
                        results: *mut PaReadResult, class_offsets: *mut u64, class_ids: *mut *const u32) -> c_int;
    pub fn pa_map_read(idx: *mut PaIndex, ascii: *const u8, len: u32, class_buf: *mut u32, class_cap: u32,
                       class_len: *mut u32, coverage: *mut u32) -> c_int;
    pub fn pa_map_read_with_mismatch(idx: *mut PaIndex, ascii: *const u8, len: u32, allowed_mismatches: u32, class_buf: *mut u32,
                                     class_cap: u32, class_len: *mut u32, coverage: *mut u32, mismatches: *mut u32) -> c_int;
    // reads the caller holds 2-bit packed (a DnaString): no ASCII round trip
    pub fn pa_map_batch_packed(idx: *mut PaIndex, words: *const u64, word_offsets: *const u64, lens: *const u32, n_reads: u64, layout: c_int,
                               allowed_mismatches: u32, results: *mut PaReadResult, class_offsets: *mut u64, class_ids: *mut *const u32) -> c_int;
    pub fn pa_map_read_packed(idx: *mut PaIndex, words: *const u64, len: u32, layout: c_int, allowed_mismatches: u32, class_buf: *mut u32,
                              class_cap: u32, class_len: *mut u32, coverage: *mut u32, mismatches: *mut u32) -> c_int;
    // process_reads for a caller that holds the reader: push records, pull the Debug tuples (overlapped batch pipeline inside)
    pub fn pa_record_stream_create(idx: *mut PaIndex, num_threads: c_int, batch_reads: u64, out: *mut *mut PaRecordStream) -> c_int;
    pub fn pa_record_stream_create_multi(idx: *const *mut PaIndex, n_idx: c_int, num_threads: c_int, batch_reads: u64, out: *mut *mut PaRecordStream) -> c_int;
    pub fn pa_records_push(s: *mut PaRecordStream, ids: *const u8, id_offsets: *const u64, seqs: *const u8, seq_offsets: *const u64, n_records: u64) -> c_int;
    pub fn pa_records_pull(s: *mut PaRecordStream, buf: *mut c_char, cap: usize, n_bytes: *mut usize) -> c_int;
    pub fn pa_records_flush(s: *mut PaRecordStream) -> c_int;
    pub fn pa_record_stream_stats(s: *const PaRecordStream, n_reads: *mut u64, n_flagged: *mut u64) -> c_int;
    pub fn pa_record_stream_stage_seconds(s: *const PaRecordStream, out: *mut f64) -> c_int;      // double out[8]
    pub fn pa_process_reads_stage_seconds(out: *mut f64) -> c_int;
    pub fn pa_record_stream_destroy(s: *mut PaRecordStream);
    pub fn pa_process_reads(idx: *mut PaIndex, fastq_path: *const c_char, out_path: *const c_char, num_threads: c_int,
                            n_reads: *mut u64, n_flagged: *mut u64) -> c_int;
    pub fn pa_process_reads_multi(idx: *const *mut PaIndex, n_idx: c_int, fastq_path: *const c_char, out_path: *const c_char, num_threads: c_int,
                                  n_reads: *mut u64, n_flagged: *mut u64) -> c_int;
    pub fn pa_fastq_scan_host(fastq_path: *const c_char, num_threads: c_int, n_records: *mut u64, starts: *mut u64, header_len: *mut u32,
                              seq_len: *mut u32, capacity: u64, text_kind: *mut c_int) -> c_int;

    // device-resident batches + the fused class-count table
    pub fn pa_words_per_read(max_read_len: u32) -> u32;
    pub fn pa_tiles_words(n_reads: u64, words_per_read: u32) -> usize;
    pub fn pa_encode_reads_device(idx: *const PaIndex, d_ascii: *const u8, d_offsets: *const u64, n_reads: u64, words_per_read: u32,
                                  d_tiles: *mut u64, d_lens: *mut u32, stream: *mut c_void) -> c_int;
    pub fn pa_map_count_batch_device(idx: *mut PaIndex, d_tiles: *const u64, d_lens: *const u32, n_reads: u64,
                                     words_per_read: u32, allowed_mismatches: u32, d_results: *mut PaReadResult,
                                     d_arena: *mut u32, arena_cap: u64, d_counts: *mut u64, stream: *mut c_void) -> c_int;
    pub fn pa_map_count_batch_uniform_device(idx: *mut PaIndex, d_tiles: *const u64, read_len: u32, n_reads: u64,
                                             words_per_read: u32, allowed_mismatches: u32, d_results: *mut PaReadResult,
                                             d_arena: *mut u32, arena_cap: u64, d_counts: *mut u64, stream: *mut c_void) -> c_int;
    // host to host: a batch in (pinned) host memory -> compact 8-byte records + packed classes + count table, chunks pipelined over streams
    pub fn pa_map_tiles_host(idx: *mut PaIndex, h_tiles: *const u64, h_lens: *const u32, uniform_len: u32, n_reads: u64, words_per_read: u32,
                             allowed_mismatches: u32, h_compact: *mut u64, h_packed: *mut u32, packed_cap: u64, packed_words: *mut u64, h_counts: *mut u64,
                             chunk_reads: u64, n_streams: c_int) -> c_int;
    pub fn pa_host_alloc_pinned(bytes: usize, out: *mut *mut c_void) -> c_int;
    pub fn pa_host_free_pinned(p: *mut c_void) -> c_int;
    pub fn pa_compact_scratch_bytes(n_reads: u64) -> usize;
    pub fn pa_results_compact_device(idx: *mut PaIndex, d_results: *const PaReadResult, d_arena: *const u32, arena_cap: u64, n_reads: u64, d_compact: *mut u64,
                                     d_packed: *mut u32, packed_cap: u64, d_packed_words: *mut u64, d_scratch: *mut c_void, scratch_bytes: usize,
                                     stream: *mut c_void) -> c_int;
    pub fn pa_map_finish(idx: *mut PaIndex, stream: *mut c_void, arena_used: *mut u64, arena_needed: *mut u64) -> c_int;
    pub fn pa_index_release_stream(idx: *mut PaIndex, stream: *mut c_void) -> c_int;
    pub fn pa_index_set_timing(idx: *mut PaIndex, on: c_int) -> c_int;
    pub fn pa_map_kernel_ms(idx: *mut PaIndex, stream: *mut c_void, ms: *mut f32) -> c_int;
    pub fn pa_map_stage_ms(idx: *mut PaIndex, stream: *mut c_void, ms: *mut f32) -> c_int;        // float ms[3]
    pub fn pa_map_arena_hint(idx: *const PaIndex, n_reads: u64) -> u64;
    pub fn pa_counts_len(idx: *const PaIndex) -> u64;

    // novel classes + the reduction over GPUs
    pub fn pa_overflow_create(device: c_int, max_classes: u64, max_ids: u64, out: *mut *mut PaOverflow) -> c_int;
    pub fn pa_overflow_destroy(ovf: *mut PaOverflow);
    pub fn pa_index_set_overflow(idx: *mut PaIndex, ovf: *mut PaOverflow) -> c_int;
    pub fn pa_overflow_fetch(ovf: *mut PaOverflow, stream: *mut c_void, words: *mut *const u32, n_words: *mut u64) -> c_int;
    pub fn pa_comm_unique_id(id: *mut u8) -> c_int;                               // 128 bytes
    pub fn pa_comm_create(device: c_int, nranks: c_int, rank: c_int, id: *const u8, out: *mut *mut PaComm) -> c_int;
    pub fn pa_comm_destroy(comm: *mut PaComm);
    pub fn pa_counts_allreduce(idx: *mut PaIndex, d_counts: *mut u64, comm: *mut PaComm, stream: *mut c_void) -> c_int;
    pub fn pa_overflow_allgather(ovf: *mut PaOverflow, comm: *mut PaComm, stream: *mut c_void, words: *mut *const u32, n_words: *mut u64) -> c_int;

    // plumbing for hosts without a HIP binding of their own
    pub fn pa_device_malloc(device: c_int, bytes: usize, out: *mut *mut c_void) -> c_int;
    pub fn pa_device_free(p: *mut c_void) -> c_int;
    pub fn pa_memcpy_h2d(dst: *mut c_void, src: *const c_void, bytes: usize, stream: *mut c_void) -> c_int;
    pub fn pa_memcpy_d2h(dst: *mut c_void, src: *const c_void, bytes: usize, stream: *mut c_void) -> c_int;
    pub fn pa_memset_device(dst: *mut c_void, value: c_int, bytes: usize, stream: *mut c_void) -> c_int;
}
