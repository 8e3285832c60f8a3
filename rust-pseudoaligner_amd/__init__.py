"""MI355X-native pseudoalignment hot path behind the API surface of rust-pseudoaligner's `Pseudoaligner`.

Python here is plumbing over the C ABI (include/pseudoaligner_amd.h): the product is libpseudoaligner_amd.so
(hand-written HIP for gfx950 + a C++ host runtime). Nothing in this package imports the CPU oracle and there is no
CPU fallback: without the built library or without a GPU the mapping entry points raise.

Mirrors (names and argument meaning follow the reference):
    Pseudoaligner.map_read / map_read_with_mismatch / map_read_to_nodes   src/pseudoaligner.rs:381, 361, 54
    process_reads                                                          src/pseudoaligner.rs:420
    build_index                                                            src/build_index.rs:27 (CPU)
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _build, _ffi
from ._ffi import (PA_DEFAULT_ALLOWED_MISMATCHES, PA_ERR_ARENA_FULL, PA_MAPPED_BIT, PA_OK, PA_READ_COVERAGE_THRESHOLD, FlatIndex,
                   IndexStats, PaError, ReadResult, check, lib, vp)

__all__ = ["HostIndex", "Txome", "Pseudoaligner", "build_index", "process_reads", "process_reads_multi", "PaError", "lib", "concat_reads",
           "gather_classes", "unpack_compact", "unpack_tiles", "RESULT_DTYPE", "PA_MAPPED_BIT", "PA_DEFAULT_ALLOWED_MISMATCHES",
           "PA_READ_COVERAGE_THRESHOLD", "PA_CLASS_REF", "Overflow", "Comm", "parse_overflow", "serialise_overflow", "overflow_merge"]

PA_CLASS_REF = 0x80000000
RESULT_DTYPE = np.dtype([("coverage", "<u4"), ("mismatches", "<u4"), ("class_off", "<u4"), ("class_len", "<u4")])


def _np_view(ptr: int, n: int, dtype, owner=None) -> np.ndarray:
    """numpy view of library-owned memory; `owner` (the wrapper object whose handle owns it) is kept alive by the view"""
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_uint8 * (int(n) * np.dtype(dtype).itemsize)).from_address(ptr)
    buf._pa_owner = owner          # ndarray.base -> buf -> owner: the index cannot be destroyed under a live view
    return np.frombuffer(buf, dtype=dtype)


def concat_reads(reads: Sequence) -> Tuple[np.ndarray, np.ndarray]:
    """ASCII reads -> (concatenated uint8, offsets[n+1] uint64)."""
    bs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
    offsets = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offsets[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    data = np.frombuffer(b"".join(bs), dtype=np.uint8) if bs else np.zeros(0, np.uint8)
    return np.ascontiguousarray(data), offsets


class HostIndex:
    """CPU-side index: the flat form of `pub struct Pseudoaligner<K>` (src/pseudoaligner.rs:26-33)."""

    def __init__(self, handle: int):
        self._h = vp(handle)

    @classmethod
    def build_fasta(cls, fasta_path: str, k: int, num_threads: int = 0) -> "HostIndex":
        h = vp()
        check(lib().pa_host_index_build_fasta(str(fasta_path).encode(), k, num_threads, C.byref(h)))
        return cls(h.value)

    @classmethod
    def build_packed(cls, packed: np.ndarray, tx_start: np.ndarray, k: int, num_threads: int = 0) -> "HostIndex":
        packed = np.ascontiguousarray(packed, dtype=np.uint64)
        tx_start = np.ascontiguousarray(tx_start, dtype=np.uint64)
        h = vp()
        check(lib().pa_host_index_build_packed(packed.ctypes.data, tx_start.ctypes.data, len(tx_start) - 1, k, num_threads, C.byref(h)))
        return cls(h.value)

    @classmethod
    def from_txome(cls, txome: "Txome", k: int, num_threads: int = 0) -> "HostIndex":
        packed, tx_start = txome.arrays()
        return cls.build_packed(packed, tx_start, k, num_threads)

    # the same index with the graph construction on the GPU (csrc/index_build.hip)
    @classmethod
    def build_fasta_device(cls, fasta_path: str, k: int, device: int = 0) -> "HostIndex":
        h = vp()
        check(lib().pa_host_index_build_fasta_device(str(fasta_path).encode(), k, device, C.byref(h)))
        return cls(h.value)

    @classmethod
    def build_packed_device(cls, packed: np.ndarray, tx_start: np.ndarray, k: int, device: int = 0) -> "HostIndex":
        packed = np.ascontiguousarray(packed, dtype=np.uint64)
        tx_start = np.ascontiguousarray(tx_start, dtype=np.uint64)
        h = vp()
        check(lib().pa_host_index_build_packed_device(packed.ctypes.data, tx_start.ctypes.data, len(tx_start) - 1, k, device, C.byref(h)))
        return cls(h.value)

    @classmethod
    def from_txome_device(cls, txome: "Txome", k: int, device: int = 0) -> "HostIndex":
        packed, tx_start = txome.arrays()
        return cls.build_packed_device(packed, tx_start, k, device)

    @classmethod
    def from_flat(cls, flat: FlatIndex) -> "HostIndex":
        h = vp()
        check(lib().pa_host_index_from_flat(C.byref(flat), C.byref(h)))
        return cls(h.value)

    @classmethod
    def load(cls, path: str) -> "HostIndex":
        h = vp()
        check(lib().pa_host_index_load(str(path).encode(), C.byref(h)))
        return cls(h.value)

    def save(self, path: str) -> None:
        check(lib().pa_host_index_save(self._h, str(path).encode()))

    def compare(self, other: "HostIndex", max_kmers: int = 1 << 26) -> Tuple[int, str]:
        """pa_host_index_compare: (0 equivalent | 1 different | 2 undecided, one line of explanation)"""
        buf = C.create_string_buffer(512)
        rc = check(lib().pa_host_index_compare(self._h, other._h, max_kmers, buf, len(buf)))
        return rc, buf.value.decode()

    def flat(self) -> FlatIndex:
        f = FlatIndex()
        check(lib().pa_host_index_view(self._h, C.byref(f)))
        return f

    def arrays(self) -> dict:
        """numpy views (no copy) of the flat arrays; valid while this object is alive."""
        f = self.flat()
        n, c = f.num_nodes, f.num_classes
        ec_offset = _np_view(f.ec_offset, c + 1, np.uint64, owner=self)
        return dict(k=f.k, num_nodes=n, num_classes=c, num_transcripts=f.num_transcripts,
                    node_seq=_np_view(f.node_seq, (f.seq_bases + 31) // 32 + 1, np.uint64, owner=self),
                    node_start=_np_view(f.node_start, n + 1, np.uint64, owner=self), node_len=_np_view(f.node_len, n, np.uint32, owner=self),
                    node_exts=_np_view(f.node_exts, n, np.uint8, owner=self), node_colour=_np_view(f.node_colour, n, np.uint32, owner=self),
                    ec_offset=ec_offset, ec_ids=_np_view(f.ec_ids, int(ec_offset[-1]) if c else 0, np.uint32, owner=self))

    @property
    def k(self) -> int:
        return self.flat().k

    @property
    def num_transcripts(self) -> int:
        return lib().pa_host_index_num_transcripts(self._h)

    def tx_names(self) -> List[str]:
        return [lib().pa_host_index_tx_name(self._h, i).decode() for i in range(self.num_transcripts)]

    def tx_genes(self) -> List[str]:
        return [lib().pa_host_index_tx_gene(self._h, i).decode() for i in range(self.num_transcripts)]

    def genes(self) -> Tuple[np.ndarray, List[str]]:
        """(gene id of each transcript, gene names): tx_gene_mapping (src/pseudoaligner.rs:32), genes numbered by first appearance"""
        n = C.c_uint32()
        tx_gene = np.zeros(max(self.num_transcripts, 1), np.uint32)
        check(lib().pa_host_index_genes(self._h, tx_gene.ctypes.data, C.byref(n)))
        return tx_gene[: self.num_transcripts], [lib().pa_host_index_gene_name(self._h, g).decode() for g in range(n.value)]

    def collapse_to_genes(self, class_counts: np.ndarray) -> np.ndarray:
        """gene-level counts (last entry: classes spanning several genes) from a class-count table of counts_len entries"""
        n = C.c_uint32()
        check(lib().pa_host_index_genes(self._h, None, C.byref(n)))
        cc = np.ascontiguousarray(class_counts, np.uint64)
        out = np.zeros(n.value + 1, np.uint64)
        check(lib().pa_counts_collapse_genes(self._h, cc.ctypes.data, len(cc), out.ctypes.data))
        return out

    def mappability(self) -> Tuple[np.ndarray, np.ndarray]:
        """analyze_graph (src/mappability.rs:120-156): (tx_multiplicity, gene_multiplicity), each [num_transcripts, 11]"""
        W = 11   # PA_MAPPABILITY_COUNTS_LEN
        tm = np.zeros((max(self.num_transcripts, 1), W), np.uint64)
        gm = np.zeros_like(tm)
        check(lib().pa_host_index_mappability(self._h, tm.ctypes.data, gm.ctypes.data))
        return tm[: self.num_transcripts], gm[: self.num_transcripts]

    def write_mappability_tsv(self, path: str) -> None:
        """write_mappability_tsv (src/mappability.rs:91-104)"""
        check(lib().pa_write_mappability_tsv(self._h, str(path).encode()))

    def transcripts(self) -> Tuple[np.ndarray, np.ndarray]:
        p, s, n = vp(), vp(), C.c_uint32()
        check(lib().pa_host_index_transcripts(self._h, C.byref(p), C.byref(s), C.byref(n)))
        tx_start = _np_view(s.value, n.value + 1, np.uint64, owner=self)
        return _np_view(p.value, (int(tx_start[-1]) + 31) // 32 + 1, np.uint64, owner=self), tx_start

    def __del__(self):
        try:
            if self._h:
                lib().pa_host_index_destroy(self._h)
                self._h = vp()
        except Exception:
            pass


class Txome:
    """A transcript set (packed) used to build indices and to simulate reads."""

    def __init__(self, handle: int):
        self._h = vp(handle)
        self._dev = {}

    @classmethod
    def synthesize(cls, num_genes: int, target_transcripts: int, seed: int) -> "Txome":
        h = vp()
        check(lib().pa_txome_synthesize(num_genes, target_transcripts, seed, C.byref(h)))
        return cls(h.value)

    # bench.py's "config3r": 40 old (10-15 % diverged) + 10 young (3-6 %) repeat families of 300 bases in the last exon of 13 % of the genes, 200 low-complexity tracts
    REPEATS_DEFAULT = dict(families=40, element_len=300, div_lo_ppm=100000, div_hi_ppm=150000, young_families=10, young_div_lo_ppm=30000, young_div_hi_ppm=60000,
                           gene_fraction_ppm=130000, low_complexity_genes=200)

    @classmethod
    def synthesize_repeats(cls, num_genes: int, target_transcripts: int, seed: int, **repeats) -> "Txome":
        """pa_txome_synthesize_repeats: the transcriptome of synthesize() with interspersed repeats and low-complexity tracts"""
        cfg = dict(cls.REPEATS_DEFAULT, **repeats)
        r = _ffi.SynthRepeats(**cfg)
        h = vp()
        check(lib().pa_txome_synthesize_repeats(num_genes, target_transcripts, seed, C.byref(r), C.byref(h)))
        return cls(h.value)

    @classmethod
    def from_fasta(cls, path: str) -> "Txome":
        h = vp()
        check(lib().pa_txome_from_fasta(str(path).encode(), C.byref(h)))
        return cls(h.value)

    @classmethod
    def from_host_index(cls, index: HostIndex) -> "Txome":
        h = vp()
        check(lib().pa_txome_from_host_index(index._h, C.byref(h)))
        return cls(h.value)

    def arrays(self) -> Tuple[np.ndarray, np.ndarray]:
        p, s, n = vp(), vp(), C.c_uint32()
        check(lib().pa_txome_view(self._h, C.byref(p), C.byref(s), C.byref(n)))
        tx_start = _np_view(s.value, n.value + 1, np.uint64, owner=self)
        return _np_view(p.value, (int(tx_start[-1]) + 31) // 32 + 1, np.uint64, owner=self), tx_start

    @property
    def num_transcripts(self) -> int:
        return len(self.arrays()[1]) - 1

    def simulate_host(self, read_len: int, seed: int, n_reads: int, sub_rate_ppm: int = 0, first_read: int = 0,
                      words_per_read: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray]:
        wpr = words_per_read or lib().pa_words_per_read(read_len)
        tiles = np.zeros(lib().pa_tiles_words(n_reads, wpr), dtype=np.uint64)
        lens = np.zeros(n_reads, dtype=np.uint32)
        check(lib().pa_simulate_reads_host(self._h, read_len, seed, sub_rate_ppm, first_read, n_reads, wpr, tiles.ctypes.data, lens.ctypes.data))
        return tiles, lens

    def device_handle(self, read_len: int, device: int = 0) -> vp:
        key = (read_len, device)
        if key not in self._dev:
            h = vp()
            check(lib().pa_txome_upload(self._h, read_len, device, C.byref(h)))
            self._dev[key] = h
        return self._dev[key]

    def simulate_device(self, read_len: int, seed: int, n_reads: int, d_tiles: int, d_lens: int, sub_rate_ppm: int = 0,
                        first_read: int = 0, words_per_read: Optional[int] = None, device: int = 0, stream: int = 0) -> None:
        wpr = words_per_read or lib().pa_words_per_read(read_len)
        check(lib().pa_simulate_reads_device(self.device_handle(read_len, device), seed, sub_rate_ppm, first_read, n_reads, wpr,
                                             d_tiles, d_lens, stream or None))

    def __del__(self):
        try:
            for h in self._dev.values():
                lib().pa_txome_device_destroy(h)
            self._dev = {}
            if self._h:
                lib().pa_txome_destroy(self._h)
                self._h = vp()
        except Exception:
            pass


def build_index(fasta_path: str, k: int, num_threads: int = 0) -> HostIndex:
    """build_index (src/build_index.rs:27-32) over utils::read_transcripts (src/utils.rs:61); CPU."""
    return HostIndex.build_fasta(fasta_path, k, num_threads)


def encode_reads_host(reads: Sequence, words_per_read: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray, int]:
    """ASCII reads -> (tiles, lens, words_per_read) in the device tile layout, packed on the host."""
    data, offsets = reads if isinstance(reads, tuple) else concat_reads(reads)
    n = len(offsets) - 1
    maxlen = int((offsets[1:] - offsets[:-1]).max()) if n else 1
    wpr = words_per_read or lib().pa_words_per_read(maxlen)
    tiles = np.zeros(lib().pa_tiles_words(n, wpr), dtype=np.uint64)
    lens = np.zeros(max(n, 1), dtype=np.uint32)
    check(lib().pa_encode_reads_host(data.ctypes.data, offsets.ctypes.data, n, wpr, tiles.ctypes.data, lens.ctypes.data))
    return tiles, lens[:n], wpr


def unpack_tiles(tiles: np.ndarray, lens: np.ndarray, words_per_read: int) -> List[str]:
    """tile layout -> ASCII strings (tests / debugging)."""
    t = tiles.reshape(-1, words_per_read, 64)
    out = []
    for i, L in enumerate(lens):
        words = t[i >> 6, :, i & 63]
        out.append("".join("ACGT"[(int(words[j >> 5]) >> (2 * (j & 31))) & 3] for j in range(int(L))))
    return out


class Pseudoaligner:
    """GPU-resident index with the reference's mapping API (src/pseudoaligner.rs:35-385)."""

    def __init__(self, index: HostIndex, device: int = 0):
        self.host = index
        self.device = device
        flat = index.flat()
        h = vp()
        check(lib().pa_index_create(C.byref(flat), device, C.byref(h)))
        self._h = h

    def stats(self) -> IndexStats:
        s = IndexStats()
        check(lib().pa_index_get_stats(self._h, C.byref(s)))
        return s

    # ---- single reads -------------------------------------------------------------------------------
    def map_read_with_mismatch(self, read_seq, allowed_mismatches: int) -> Optional[Tuple[List[int], int, int]]:
        """-> Some((eq_class, read_coverage, mismatches)) | None (src/pseudoaligner.rs:361-376)"""
        seq = read_seq.encode() if isinstance(read_seq, str) else bytes(read_seq)
        cap = max(64, self.stats().max_class_len)
        buf = (C.c_uint32 * cap)()
        n, cov, mm = C.c_uint32(), C.c_uint32(), C.c_uint32()
        rc = check(lib().pa_map_read_with_mismatch(self._h, seq, len(seq), allowed_mismatches, buf, cap, C.byref(n), C.byref(cov), C.byref(mm)))
        if rc == 0:
            return None
        return list(buf[: n.value]), cov.value, mm.value

    def map_read(self, read_seq) -> Optional[Tuple[List[int], int]]:
        """-> Some((eq_class, read_coverage)) | None with allowed mismatches = 2 (src/pseudoaligner.rs:381-384)"""
        r = self.map_read_with_mismatch(read_seq, PA_DEFAULT_ALLOWED_MISMATCHES)
        return None if r is None else (r[0], r[1])

    def map_read_to_nodes(self, read_seq, allowed_mismatches: int = PA_DEFAULT_ALLOWED_MISMATCHES) -> Optional[Tuple[List[int], int]]:
        """-> Some((nodes in visit order, read_coverage)) | None (src/pseudoaligner.rs:54-61)"""
        seq = read_seq.encode() if isinstance(read_seq, str) else bytes(read_seq)
        cap = 2 * len(seq) + 8
        buf = (C.c_uint32 * cap)()
        n, cov, mm = C.c_uint32(), C.c_uint32(), C.c_uint32()
        rc = check(lib().pa_map_read_to_nodes(self._h, seq, len(seq), allowed_mismatches, buf, cap, C.byref(n), C.byref(cov), C.byref(mm)))
        if rc == 0:
            return None
        return list(buf[: n.value]), cov.value

    # ---- batches ------------------------------------------------------------------------------------
    def map_batch(self, reads: Sequence, allowed_mismatches: int = PA_DEFAULT_ALLOWED_MISMATCHES):
        """-> (results[n] RESULT_DTYPE, class_offsets[n+1], class_ids) in input order; `mismatches` has bit 31 = mapped."""
        data, offsets = reads if isinstance(reads, tuple) else concat_reads(reads)
        n = len(offsets) - 1
        results = np.zeros(n, dtype=RESULT_DTYPE)
        coff = np.zeros(n + 1, dtype=np.uint64)
        ids = vp()
        check(lib().pa_map_batch(self._h, data.ctypes.data, offsets.ctypes.data, n, allowed_mismatches, results.ctypes.data,
                                 coff.ctypes.data, C.byref(ids)))
        class_ids = _np_view(ids.value, int(coff[-1]), np.uint32).copy()
        return results, coff, class_ids

    def map_batch_packed(self, words: np.ndarray, word_offsets: np.ndarray, lens: np.ndarray, layout: int = 0,
                         allowed_mismatches: int = PA_DEFAULT_ALLOWED_MISMATCHES):
        """map_batch for reads held 2-bit packed (a DnaString, :450): read i = lens[i] bases in words[word_offsets[i]:word_offsets[i+1]];
        layout 0 = LSB-first words, 1 = MSB-first words"""
        words = np.ascontiguousarray(words, dtype=np.uint64)
        word_offsets = np.ascontiguousarray(word_offsets, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        n = len(lens)
        results = np.zeros(n, dtype=RESULT_DTYPE)
        coff = np.zeros(n + 1, dtype=np.uint64)
        ids = vp()
        check(lib().pa_map_batch_packed(self._h, words.ctypes.data, word_offsets.ctypes.data, lens.ctypes.data, n, layout, allowed_mismatches,
                                        results.ctypes.data, coff.ctypes.data, C.byref(ids)))
        return results, coff, _np_view(ids.value, int(coff[-1]), np.uint32).copy()

    def map_read_packed(self, words: np.ndarray, length: int, layout: int = 0, allowed_mismatches: int = PA_DEFAULT_ALLOWED_MISMATCHES, cap: int = 1 << 16):
        """map_read_with_mismatch (:361) of one packed read -> (ids, coverage, mismatches) or None"""
        words = np.ascontiguousarray(words, dtype=np.uint64)
        buf = np.zeros(cap, dtype=np.uint32)
        clen, cov, mm = C.c_uint32(), C.c_uint32(), C.c_uint32()
        rc = lib().pa_map_read_packed(self._h, words.ctypes.data, length, layout, allowed_mismatches, buf.ctypes.data, cap, C.byref(clen), C.byref(cov), C.byref(mm))
        if rc < 0:
            check(rc)
        return (buf[: clen.value].tolist(), cov.value, mm.value) if rc == 1 else None

    def map_batch_nodes(self, reads: Sequence, allowed_mismatches: int = PA_DEFAULT_ALLOWED_MISMATCHES):
        data, offsets = reads if isinstance(reads, tuple) else concat_reads(reads)
        n = len(offsets) - 1
        maxlen = int((offsets[1:] - offsets[:-1]).max()) if n else 1
        stride = 2 * maxlen + 8
        results = np.zeros(n, dtype=RESULT_DTYPE)
        nodes = np.zeros((n, stride), dtype=np.uint32)
        nlen = np.zeros(n, dtype=np.uint32)
        check(lib().pa_map_batch_nodes(self._h, data.ctypes.data, offsets.ctypes.data, n, allowed_mismatches, results.ctypes.data,
                                       nodes.ctypes.data, stride, nlen.ctypes.data))
        return results, nodes, nlen

    def map_batch_device(self, d_tiles: int, d_lens: int, n_reads: int, words_per_read: int, d_results: int, d_arena: int,
                         arena_cap: int, allowed_mismatches: int = PA_DEFAULT_ALLOWED_MISMATCHES, d_colour: int = 0, stream: int = 0) -> None:
        """Asynchronous launch of the hot path on device-resident tiles (raw device pointers)."""
        check(lib().pa_map_batch_device(self._h, d_tiles, d_lens, n_reads, words_per_read, allowed_mismatches, d_results, d_arena,
                                        arena_cap, d_colour or None, stream or None))

    def map_count_batch_device(self, d_tiles: int, d_lens: int, n_reads: int, words_per_read: int, d_results: int, d_arena: int,
                               arena_cap: int, d_counts: int, allowed_mismatches: int = PA_DEFAULT_ALLOWED_MISMATCHES, stream: int = 0) -> None:
        """map_batch_device with the class-count table (u64[counts_len()]) accumulated in the same kernel."""
        check(lib().pa_map_count_batch_device(self._h, d_tiles, d_lens, n_reads, words_per_read, allowed_mismatches, d_results, d_arena,
                                              arena_cap, d_counts, stream or None))

    def map_count_batch_uniform_device(self, d_tiles: int, read_len: int, n_reads: int, words_per_read: int, d_results: int, d_arena: int,
                                       arena_cap: int, d_counts: int, allowed_mismatches: int = PA_DEFAULT_ALLOWED_MISMATCHES, stream: int = 0) -> None:
        """map_count_batch_device for a batch whose reads all have read_len bases: no length array"""
        check(lib().pa_map_count_batch_uniform_device(self._h, d_tiles, read_len, n_reads, words_per_read, allowed_mismatches, d_results, d_arena,
                                                      arena_cap, d_counts, stream or None))

    def map_tiles_host(self, h_tiles: int, n_reads: int, words_per_read: int, h_compact: int, h_packed: int, packed_cap: int, h_lens: int = 0, uniform_len: int = 0,
                       h_counts: int = 0, allowed_mismatches: int = PA_DEFAULT_ALLOWED_MISMATCHES, chunk_reads: int = 0, n_streams: int = 0) -> int:
        """pa_map_tiles_host: a batch in (pinned) host memory -> compact records + packed classes (+ count table) on the host, chunks pipelined
        over several streams; returns the words of the packed stream. Raw host pointers."""
        pw = C.c_uint64()
        check(lib().pa_map_tiles_host(self._h, h_tiles, h_lens or None, uniform_len, n_reads, words_per_read, allowed_mismatches, h_compact, h_packed or None, packed_cap,
                                      C.byref(pw), h_counts or None, chunk_reads, n_streams))
        return pw.value

    def map_finish(self, stream: int = 0) -> Tuple[int, int]:
        used, need = C.c_uint64(), C.c_uint64()
        check(lib().pa_map_finish(self._h, stream or None, C.byref(used), C.byref(need)))
        return used.value, need.value

    def set_timing(self, on: bool = True) -> None:
        """HIP events around the mapping kernel of every launch (pa_map_kernel_ms)"""
        check(lib().pa_index_set_timing(self._h, 1 if on else 0))

    def map_kernel_ms(self, stream: int = 0) -> float:
        ms = C.c_float()
        check(lib().pa_map_kernel_ms(self._h, stream or None, C.byref(ms)))
        return ms.value

    def map_stage_ms(self, stream: int = 0):
        """(mapping kernel, resolve kernel, count kernels) of the last timed launch, ms"""
        ms = (C.c_float * 3)()
        check(lib().pa_map_stage_ms(self._h, stream or None, ms))
        return ms[0], ms[1], ms[2]

    def release_stream(self, stream: int = 0) -> None:
        check(lib().pa_index_release_stream(self._h, stream or None))

    def arena_hint(self, n_reads: int) -> int:
        return lib().pa_map_arena_hint(self._h, n_reads)

    def counts_len(self) -> int:
        return lib().pa_counts_len(self._h)

    def counts_by_barcode_device(self, d_results: int, d_arena: int, d_barcode: int, n_reads: int, d_keys: int, d_vals: int,
                                 barcode_bits: int = 0, stream: int = 0) -> int:
        """sparse (barcode, class) -> reads matrix as sorted keys (barcode << 32 | column) + counts; returns the number of cells"""
        n = C.c_uint64()
        check(lib().pa_counts_by_barcode_device(self._h, d_results, d_arena, d_barcode, n_reads, barcode_bits, d_keys, d_vals, C.byref(n), stream or None))
        return n.value

    def set_overflow(self, overflow: Optional["Overflow"]) -> None:
        """attach the table that remembers WHICH novel classes the fused count launches met (None detaches)"""
        check(lib().pa_index_set_overflow(self._h, overflow._h if overflow else None))
        self._overflow = overflow   # keep it alive while attached

    def counts_allreduce(self, d_counts: int, comm: Optional["Comm"], stream: int = 0) -> None:
        """RCCL all-reduce (sum) of the dense count table, in place"""
        check(lib().pa_counts_allreduce(self._h, d_counts, comm._h if comm else None, stream or None))

    def counts_accumulate_device(self, d_results: int, d_arena: int, d_colour: int, n_reads: int, d_counts: int, stream: int = 0) -> None:
        check(lib().pa_counts_accumulate_device(self._h, d_results, d_arena, d_colour or None, n_reads, d_counts, stream or None))

    def encode_reads_device(self, d_ascii: int, d_offsets: int, n_reads: int, words_per_read: int, d_tiles: int, d_lens: int, stream: int = 0):
        check(lib().pa_encode_reads_device(self._h, d_ascii, d_offsets, n_reads, words_per_read, d_tiles, d_lens, stream or None))

    def __del__(self):
        try:
            if self._h:
                lib().pa_index_destroy(self._h)
                self._h = vp()
        except Exception:
            pass


def parse_overflow(words: np.ndarray) -> dict:
    """serialised overflow table (include/pseudoaligner_amd.h) -> {tuple(ids): count}"""
    out = {}
    if len(words) < 2:
        return out
    p = 2
    for _ in range(int(words[0])):
        n = int(words[p])
        out[tuple(int(x) for x in words[p + 3:p + 3 + n])] = int(words[p + 1]) | (int(words[p + 2]) << 32)
        p += 3 + n
    assert p == int(words[1]), "overflow table: header says %d words, records end at %d" % (int(words[1]), p)
    return out


def serialise_overflow(classes: dict) -> np.ndarray:
    """{tuple(ids): count} -> the serialised form (canonical order), e.g. to feed pa_overflow_merge"""
    w = [0, 0]
    for ids in sorted(classes):
        c = int(classes[ids])
        w += [len(ids), c & 0xFFFFFFFF, c >> 32] + list(ids)
    w[0], w[1] = len(classes), len(w)
    return np.array(w, dtype=np.uint32)


def overflow_merge(buffers: Sequence[np.ndarray]) -> np.ndarray:
    """pa_overflow_merge: serialised tables of several GPUs -> one canonical table (host side)"""
    bufs = [np.ascontiguousarray(b, dtype=np.uint32) for b in buffers]
    ptrs = (C.c_void_p * max(len(bufs), 1))(*[b.ctypes.data for b in bufs])
    sizes = (C.c_uint64 * max(len(bufs), 1))(*[len(b) for b in bufs])
    need = C.c_uint64()
    rc = lib().pa_overflow_merge(ptrs, sizes, len(bufs), None, 0, C.byref(need))
    if rc not in (PA_OK, PA_ERR_ARENA_FULL):
        check(rc)
    out = np.zeros(max(need.value, 2), dtype=np.uint32)
    check(lib().pa_overflow_merge(ptrs, sizes, len(bufs), out.ctypes.data, len(out), C.byref(need)))
    return out[: need.value]


class Overflow:
    """Per-GPU table of the NOVEL classes (results that are no index class), keyed by content."""

    def __init__(self, device: int = 0, max_classes: int = 1 << 20, max_ids: int = 1 << 24):
        h = vp()
        check(lib().pa_overflow_create(device, max_classes, max_ids, C.byref(h)))
        self._h = h

    def reset(self, stream: int = 0) -> None:
        check(lib().pa_overflow_reset(self._h, stream or None))

    def fetch(self, stream: int = 0) -> np.ndarray:
        w, n = vp(), C.c_uint64()
        check(lib().pa_overflow_fetch(self._h, stream or None, C.byref(w), C.byref(n)))
        return _np_view(w.value, n.value, np.uint32).copy()

    def allgather(self, comm: Optional["Comm"], stream: int = 0) -> np.ndarray:
        w, n = vp(), C.c_uint64()
        check(lib().pa_overflow_allgather(self._h, comm._h if comm else None, stream or None, C.byref(w), C.byref(n)))
        return _np_view(w.value, n.value, np.uint32).copy()

    def __del__(self):
        try:
            if self._h:
                lib().pa_overflow_destroy(self._h)
                self._h = vp()
        except Exception:
            pass


class RecordStream:
    """pa_record_stream: process_reads for a caller that holds the reader — push records, pull the reference's Debug tuples"""

    def __init__(self, aligner, num_threads: int = 0, batch_reads: int = 0):
        """aligner: a Pseudoaligner, or a list of them (replicas of one index: pa_record_stream_create_multi — batches round-robin over the handles)"""
        h = vp()
        if isinstance(aligner, (list, tuple)):
            hs = (vp * len(aligner))(*[a._h for a in aligner])
            check(lib().pa_record_stream_create_multi(hs, len(aligner), num_threads, batch_reads, C.byref(h)))
        else:
            check(lib().pa_record_stream_create(aligner._h, num_threads, batch_reads, C.byref(h)))
        self._h = h
        self._aligner = aligner   # (the index must outlive the stream)

    def push(self, ids: Sequence, seqs: Sequence) -> None:
        i_data, i_off = concat_reads(ids)
        s_data, s_off = concat_reads(seqs)
        check(lib().pa_records_push(self._h, i_data.ctypes.data, i_off.ctypes.data, s_data.ctypes.data, s_off.ctypes.data, len(i_off) - 1))

    def pull(self, cap: int = 1 << 20, grow: bool = True) -> bytes:
        """rendered tuples, whole lines. A tuple longer than `cap` (a class of ~100 k ids) makes the library answer PA_ERR_BUFFER_TOO_SMALL with the
        bytes it needs: the buffer is grown to that and the pull repeated (grow=False: the error is raised instead)"""
        from ._ffi import PA_ERR_BUFFER_TOO_SMALL
        for _ in range(4):
            buf = C.create_string_buffer(cap)
            n = C.c_size_t()
            rc = lib().pa_records_pull(self._h, buf, cap, C.byref(n))
            if rc == PA_ERR_BUFFER_TOO_SMALL and grow and n.value > cap:
                cap = int(n.value)
                continue
            check(rc)
            return buf.raw[: n.value]
        raise PaError(PA_ERR_BUFFER_TOO_SMALL, "pa_records_pull keeps asking for a larger buffer")

    def flush(self) -> None:
        check(lib().pa_records_flush(self._h))

    def drain(self, cap: int = 1 << 20) -> bytes:
        out = []
        while True:
            b = self.pull(cap)
            if not b:
                return b"".join(out)
            out.append(b)

    def stats(self) -> Tuple[int, int]:
        n, f = C.c_uint64(), C.c_uint64()
        check(lib().pa_record_stream_stats(self._h, C.byref(n), C.byref(f)))
        return n.value, f.value

    def stage_seconds(self) -> dict:
        st = (C.c_double * 8)()
        check(lib().pa_record_stream_stage_seconds(self._h, st))
        return dict(zip(INGEST_STAGES, st))

    def close(self) -> None:
        if self._h:
            lib().pa_record_stream_destroy(self._h)
            self._h = vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """RCCL communicator owned by the library: one rank per GPU."""

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * 128)()
        check(lib().pa_comm_unique_id(buf))
        return bytes(buf)

    def __init__(self, device: int, nranks: int, rank: int, unique_id: bytes):
        assert len(unique_id) == 128
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        h = vp()
        check(lib().pa_comm_create(device, nranks, rank, buf, C.byref(h)))
        self._h = h

    @property
    def rank(self) -> int:
        return lib().pa_comm_rank(self._h)

    @property
    def size(self) -> int:
        return lib().pa_comm_size(self._h)

    def __del__(self):
        try:
            if self._h:
                lib().pa_comm_destroy(self._h)
                self._h = vp()
        except Exception:
            pass


INGEST_STAGES = ("scan_s", "pack_s", "gpu_wait_s", "launch_s", "render_s", "writer_wait_s", "total_s", "reads")


def process_reads_stage_seconds() -> dict:
    """host-stage wall seconds of this thread's last process_reads call (pa_process_reads_stage_seconds)"""
    st = (C.c_double * 8)()
    check(lib().pa_process_reads_stage_seconds(st))
    return dict(zip(INGEST_STAGES, st))


def process_reads(fastq_path: str, index: Pseudoaligner, out_path: str = "-", num_threads: int = 2) -> Tuple[int, int]:
    """process_reads (src/pseudoaligner.rs:420-425): one `(flag, "id", [ids], coverage)` line per read, input order.
    Returns (reads, reads flagged true by the rule at :455)."""
    n, flagged = C.c_uint64(), C.c_uint64()
    check(lib().pa_process_reads(index._h, str(fastq_path).encode(), str(out_path).encode(), num_threads, C.byref(n), C.byref(flagged)))
    return n.value, flagged.value


def process_reads_multi(fastq_path: str, indexes: Sequence[Pseudoaligner], out_path: str = "-", num_threads: int = 2) -> Tuple[int, int]:
    """pa_process_reads_multi: process_reads over several handles of one index (the GPUs of a node; a handle listed twice = two lanes
    on its GPU); the output is byte for byte that of process_reads on one handle"""
    n, flagged = C.c_uint64(), C.c_uint64()
    hs = (vp * len(indexes))(*[a._h for a in indexes])
    check(lib().pa_process_reads_multi(hs, len(indexes), str(fastq_path).encode(), str(out_path).encode(), num_threads, C.byref(n), C.byref(flagged)))
    return n.value, flagged.value


def fastq_scan(fastq_path: str, num_threads: int = 2) -> Tuple[np.ndarray, np.ndarray, np.ndarray, int]:
    """pa_fastq_scan_host: the scan stage of process_reads alone (no GPU) -> (starts, header_len, seq_len, text_kind)"""
    n, kind = C.c_uint64(), C.c_int()
    check(lib().pa_fastq_scan_host(str(fastq_path).encode(), num_threads, C.byref(n), None, None, None, 0, C.byref(kind)))
    starts, hdr, seq = np.zeros(n.value, np.uint64), np.zeros(n.value, np.uint32), np.zeros(n.value, np.uint32)
    if n.value:
        check(lib().pa_fastq_scan_host(str(fastq_path).encode(), num_threads, C.byref(n), starts.ctypes.data_as(_ffi.u64p),
                                       hdr.ctypes.data_as(_ffi.u32p), seq.ctypes.data_as(_ffi.u32p), len(starts), C.byref(kind)))
    return starts, hdr, seq, kind.value


PA_COMPACT_MAPPED, PA_COMPACT_BY_REF, PA_COMPACT_PACKED = 0x10000000, 0x20000000, 0x40000000


def unpack_compact(compact: np.ndarray, packed: np.ndarray, index: "HostIndex") -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """pa_results_compact_device's 8-byte records + packed stream -> (results RESULT_DTYPE with class_off = 0, class_offsets[n+1], class_ids)
    in read order (vectorised): a by-reference class is eq_classes[id] of the flat index, a packed one the next {length, ids...} entry"""
    a = index.arrays()
    lo = (compact & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hi = (compact >> np.uint64(32)).astype(np.int64)
    n = len(compact)
    res = np.zeros(n, RESULT_DTYPE)
    res["coverage"] = lo & 0x3FFF
    res["mismatches"] = ((lo >> 14) & 0x3FFF) | np.where(lo & PA_COMPACT_MAPPED, np.uint32(PA_MAPPED_BIT), np.uint32(0))
    by_ref = (lo & PA_COMPACT_BY_REF) != 0
    pk = (lo & PA_COMPACT_PACKED) != 0
    assert not (by_ref & pk).any(), "a class was lost to a full arena"
    ec_offset = a["ec_offset"].astype(np.int64)
    lens = np.zeros(n, np.int64)
    lens[by_ref] = (ec_offset[1:] - ec_offset[:-1])[hi[by_ref]]
    packed = np.asarray(packed, np.uint32)
    # packed entries in read order: walk their lengths
    pidx = np.flatnonzero(pk)
    pstart = np.zeros(len(pidx), np.int64)
    pos = 0
    plen = np.zeros(len(pidx), np.int64)
    for j in range(len(pidx)):     # (sequential by construction: entry j starts where entry j - 1 ends)
        plen[j] = int(packed[pos])
        pstart[j] = pos + 1
        pos += 1 + int(plen[j])
    assert pos == len(packed), "packed stream: %d words walked, %d given" % (pos, len(packed))
    lens[pidx] = plen
    res["class_len"] = lens
    coff = np.zeros(n + 1, np.uint64)
    coff[1:] = np.cumsum(lens)
    total = int(coff[-1])
    if total == 0:
        return res, coff, np.zeros(0, np.uint32)
    src = np.concatenate([packed, a["ec_ids"]])
    start = np.zeros(n, np.int64)
    start[by_ref] = len(packed) + ec_offset[hi[by_ref]]
    start[pidx] = pstart
    within = np.arange(total, dtype=np.int64) - np.repeat(coff[:-1].astype(np.int64), lens)
    return res, coff, src[np.repeat(start, lens) + within].astype(np.uint32)


def gather_classes(results: np.ndarray, arena: np.ndarray, index: "HostIndex") -> Tuple[np.ndarray, np.ndarray]:
    """Device-format results -> (class_offsets[n+1], class_ids) in read order (vectorised). A class is either given by
    reference (class_off has PA_CLASS_REF set: it is eq_classes[class_off & 0x7FFFFFFF] of the flat index) or by its
    offset in the arena."""
    a = index.arrays()
    lens = results["class_len"].astype(np.int64)
    coff = np.zeros(len(results) + 1, dtype=np.uint64)
    coff[1:] = np.cumsum(lens)
    total = int(coff[-1])
    if total == 0:
        return coff, np.zeros(0, np.uint32)
    off = results["class_off"].astype(np.int64)
    is_ref = (off & 0x80000000) != 0
    ec_offset = a["ec_offset"].astype(np.int64)
    src = np.concatenate([np.asarray(arena, dtype=np.uint32), a["ec_ids"]])
    cid = np.where(is_ref & (lens > 0), off & 0x7FFFFFFF, 0)
    start = np.where(is_ref, len(arena) + ec_offset[cid], off)
    starts = np.repeat(start, lens)
    within = np.arange(total, dtype=np.int64) - np.repeat(coff[:-1].astype(np.int64), lens)
    return coff, src[starts + within].astype(np.uint32)
