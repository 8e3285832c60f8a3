"""ctypes binding of include/pseudoaligner_amd.h (the same symbols a Rust `extern "C"` block would bind)."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

from . import _build

u8p, u32p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
vp = C.c_void_p

PA_OK = 0
PA_ERR_NO_DEVICE = -4
PA_ERR_ARENA_FULL = -7
PA_ERR_BUFFER_TOO_SMALL = -10
PA_MAPPED_BIT = 0x80000000
PA_DEFAULT_ALLOWED_MISMATCHES = 2
PA_READ_COVERAGE_THRESHOLD = 32
PA_MAX_READ_LEN = 1048575


class PaError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__("pseudoaligner_amd error %d: %s" % (code, message))
        self.code = code


class FlatIndex(C.Structure):
    _fields_ = [("k", C.c_uint32), ("num_nodes", C.c_uint32), ("num_classes", C.c_uint32), ("num_transcripts", C.c_uint32),
                ("seq_bases", C.c_uint64), ("node_seq", vp), ("node_start", vp), ("node_len", vp), ("node_exts", vp),
                ("node_colour", vp), ("ec_offset", vp), ("ec_ids", vp), ("node_redge", vp), ("node_ledge", vp)]


class ReadResult(C.Structure):
    _fields_ = [("coverage", C.c_uint32), ("mismatches", C.c_uint32), ("class_off", C.c_uint32), ("class_len", C.c_uint32)]


class SynthRepeats(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("families", "element_len", "div_lo_ppm", "div_hi_ppm", "young_families", "young_div_lo_ppm", "young_div_hi_ppm",
                                          "gene_fraction_ppm", "low_complexity_genes")]


class IndexStats(C.Structure):
    _fields_ = [("num_kmers", C.c_uint64), ("table_slots", C.c_uint64), ("bytes_table", C.c_uint64), ("bytes_graph", C.c_uint64),
                ("bytes_classes", C.c_uint64), ("bytes_total", C.c_uint64), ("num_nodes", C.c_uint32), ("num_classes", C.c_uint32),
                ("k", C.c_uint32), ("max_class_len", C.c_uint32)]


# name -> (restype, argtypes); every symbol declared in include/pseudoaligner_amd.h
SIGNATURES = {
    "pa_abi_version": (C.c_uint32, []),
    "pa_device_count": (C.c_int, []),
    "pa_last_error": (C.c_char_p, []),
    "pa_host_index_build_fasta": (C.c_int, [C.c_char_p, C.c_uint32, C.c_int, C.POINTER(vp)]),
    "pa_host_index_build_packed": (C.c_int, [vp, vp, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(vp)]),
    "pa_host_index_build_fasta_device": (C.c_int, [C.c_char_p, C.c_uint32, C.c_int, C.POINTER(vp)]),
    "pa_host_index_build_packed_device": (C.c_int, [vp, vp, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(vp)]),
    "pa_host_index_from_flat": (C.c_int, [C.POINTER(FlatIndex), C.POINTER(vp)]),
    "pa_host_index_view": (C.c_int, [vp, C.POINTER(FlatIndex)]),
    "pa_host_index_compare": (C.c_int, [vp, vp, C.c_uint64, C.c_char_p, C.c_size_t]),
    "pa_host_index_save": (C.c_int, [vp, C.c_char_p]),
    "pa_host_index_load": (C.c_int, [C.c_char_p, C.POINTER(vp)]),
    "pa_host_index_num_transcripts": (C.c_uint32, [vp]),
    "pa_host_index_tx_name": (C.c_char_p, [vp, C.c_uint32]),
    "pa_host_index_tx_gene": (C.c_char_p, [vp, C.c_uint32]),
    "pa_host_index_genes": (C.c_int, [vp, vp, u32p]),
    "pa_host_index_gene_name": (C.c_char_p, [vp, C.c_uint32]),
    "pa_counts_collapse_genes": (C.c_int, [vp, vp, C.c_uint64, vp]),
    "pa_host_index_mappability": (C.c_int, [vp, vp, vp]),
    "pa_write_mappability_tsv": (C.c_int, [vp, C.c_char_p]),
    "pa_host_index_transcripts": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), u32p]),
    "pa_host_index_destroy": (None, [vp]),
    "pa_index_create": (C.c_int, [C.POINTER(FlatIndex), C.c_int, C.POINTER(vp)]),
    "pa_index_create_multi": (C.c_int, [C.POINTER(FlatIndex), C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]),
    "pa_index_get_stats": (C.c_int, [vp, C.POINTER(IndexStats)]),
    "pa_index_destroy": (None, [vp]),
    "pa_tiles_words": (C.c_size_t, [C.c_uint64, C.c_uint32]),
    "pa_words_per_read": (C.c_uint32, [C.c_uint32]),
    "pa_encode_reads_device": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_uint32, vp, vp, vp]),
    "pa_encode_reads_host": (C.c_int, [vp, vp, C.c_uint64, C.c_uint32, vp, vp]),
    "pa_map_batch_device": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, vp, vp, C.c_uint64, vp, vp]),
    "pa_map_count_batch_device": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, vp, vp, C.c_uint64, vp, vp]),
    "pa_map_count_batch_uniform_device": (C.c_int, [vp, vp, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, vp, vp, C.c_uint64, vp, vp]),
    "pa_map_finish": (C.c_int, [vp, vp, u64p, u64p]),
    "pa_index_release_stream": (C.c_int, [vp, vp]),
    "pa_map_batch_packed": (C.c_int, [vp, vp, vp, vp, C.c_uint64, C.c_int, C.c_uint32, vp, vp, C.POINTER(vp)]),
    "pa_map_read_packed": (C.c_int, [vp, vp, C.c_uint32, C.c_int, C.c_uint32, vp, C.c_uint32, u32p, u32p, u32p]),
    "pa_record_stream_create": (C.c_int, [vp, C.c_int, C.c_uint64, C.POINTER(vp)]),
    "pa_record_stream_create_multi": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, C.c_uint64, C.POINTER(vp)]),
    "pa_records_push": (C.c_int, [vp, vp, vp, vp, vp, C.c_uint64]),
    "pa_records_pull": (C.c_int, [vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "pa_records_flush": (C.c_int, [vp]),
    "pa_record_stream_stats": (C.c_int, [vp, u64p, u64p]),
    "pa_record_stream_destroy": (None, [vp]),
    "pa_index_set_timing": (C.c_int, [vp, C.c_int]),
    "pa_map_kernel_ms": (C.c_int, [vp, vp, C.POINTER(C.c_float)]),
    "pa_map_stage_ms": (C.c_int, [vp, vp, C.POINTER(C.c_float)]),
    "pa_process_reads_stage_seconds": (C.c_int, [C.POINTER(C.c_double)]),
    "pa_record_stream_stage_seconds": (C.c_int, [vp, C.POINTER(C.c_double)]),
    "pa_map_arena_hint": (C.c_uint64, [vp, C.c_uint64]),
    "pa_map_tiles_host": (C.c_int, [vp, vp, vp, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, vp, vp, C.c_uint64, u64p, vp, C.c_uint64, C.c_int]),
    "pa_host_alloc_pinned": (C.c_int, [C.c_size_t, C.POINTER(vp)]),
    "pa_host_free_pinned": (C.c_int, [vp]),
    "pa_compact_scratch_bytes": (C.c_size_t, [C.c_uint64]),
    "pa_results_compact_device": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_uint64, vp, vp, C.c_uint64, vp, vp, C.c_size_t, vp]),
    "pa_map_batch": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_uint32, vp, vp, C.POINTER(vp)]),
    "pa_map_read": (C.c_int, [vp, C.c_char_p, C.c_uint32, vp, C.c_uint32, u32p, u32p]),
    "pa_map_read_with_mismatch": (C.c_int, [vp, C.c_char_p, C.c_uint32, C.c_uint32, vp, C.c_uint32, u32p, u32p, u32p]),
    "pa_map_read_to_nodes": (C.c_int, [vp, C.c_char_p, C.c_uint32, C.c_uint32, vp, C.c_uint32, u32p, u32p, u32p]),
    "pa_map_batch_nodes": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_uint32, vp, vp, C.c_uint32, vp]),
    "pa_process_reads": (C.c_int, [vp, C.c_char_p, C.c_char_p, C.c_int, u64p, u64p]),
    "pa_process_reads_multi": (C.c_int, [C.POINTER(vp), C.c_int, C.c_char_p, C.c_char_p, C.c_int, u64p, u64p]),
    "pa_fastq_scan_host": (C.c_int, [C.c_char_p, C.c_int, u64p, u64p, u32p, u32p, C.c_uint64, C.POINTER(C.c_int)]),
    "pa_counts_len": (C.c_uint64, [vp]),
    "pa_counts_accumulate_device": (C.c_int, [vp, vp, vp, vp, C.c_uint64, vp, vp]),
    "pa_counts_by_barcode_device": (C.c_int, [vp, vp, vp, vp, C.c_uint64, C.c_uint32, vp, vp, u64p, vp]),
    "pa_overflow_create": (C.c_int, [C.c_int, C.c_uint64, C.c_uint64, C.POINTER(vp)]),
    "pa_overflow_destroy": (None, [vp]),
    "pa_overflow_reset": (C.c_int, [vp, vp]),
    "pa_index_set_overflow": (C.c_int, [vp, vp]),
    "pa_overflow_fetch": (C.c_int, [vp, vp, C.POINTER(vp), u64p]),
    "pa_overflow_merge": (C.c_int, [vp, vp, C.c_int, vp, C.c_uint64, u64p]),
    "pa_comm_unique_id": (C.c_int, [vp]),
    "pa_comm_create": (C.c_int, [C.c_int, C.c_int, C.c_int, vp, C.POINTER(vp)]),
    "pa_comm_destroy": (None, [vp]),
    "pa_comm_rank": (C.c_int, [vp]),
    "pa_comm_size": (C.c_int, [vp]),
    "pa_counts_allreduce": (C.c_int, [vp, vp, vp, vp]),
    "pa_overflow_allgather": (C.c_int, [vp, vp, vp, C.POINTER(vp), u64p]),
    "pa_txome_synthesize": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(vp)]),
    "pa_txome_synthesize_repeats": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(SynthRepeats), C.POINTER(vp)]),
    "pa_txome_from_host_index": (C.c_int, [vp, C.POINTER(vp)]),
    "pa_txome_from_fasta": (C.c_int, [C.c_char_p, C.POINTER(vp)]),
    "pa_txome_view": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), u32p]),
    "pa_txome_destroy": (None, [vp]),
    "pa_simulate_reads_host": (C.c_int, [vp, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32, vp, vp]),
    "pa_txome_upload": (C.c_int, [vp, C.c_uint32, C.c_int, C.POINTER(vp)]),
    "pa_txome_device_destroy": (None, [vp]),
    "pa_simulate_reads_device": (C.c_int, [vp, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32, vp, vp, vp]),
    "pa_event_create": (C.c_int, [C.POINTER(vp)]),
    "pa_event_record": (C.c_int, [vp, vp]),
    "pa_event_elapsed_ms": (C.c_int, [vp, vp, C.POINTER(C.c_float)]),
    "pa_event_destroy": (C.c_int, [vp]),
    "pa_device_malloc": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(vp)]),
    "pa_device_free": (C.c_int, [vp]),
    "pa_memcpy_h2d": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "pa_memcpy_d2h": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "pa_memset_device": (C.c_int, [vp, C.c_int, C.c_size_t, vp]),
    "pa_stream_synchronize": (C.c_int, [vp]),
}

_lib = None


def library_path() -> Path:
    """the in-tree product library; PA_PRODUCT_SO (A/B runs of bench.py against an older build, tools/baseline) overrides it"""
    import os
    alt = os.environ.get("PA_PRODUCT_SO")
    return Path(alt) if alt else _build.PRODUCT_SO


def lib() -> C.CDLL:
    """Load libpseudoaligner_amd.so. There is no fallback: a missing library is an error."""
    global _lib
    if _lib is None:
        path = library_path()
        if not path.exists():
            raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
        handle = C.CDLL(str(path))
        for name, (res, args) in SIGNATURES.items():
            if path != _build.PRODUCT_SO and not hasattr(handle, name):
                continue                 # an older build under PA_PRODUCT_SO may lack newer entry points
            fn = getattr(handle, name)   # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc: int) -> int:
    if rc < 0:
        raise PaError(rc, (lib().pa_last_error() or b"").decode("utf-8", "replace"))
    return rc
