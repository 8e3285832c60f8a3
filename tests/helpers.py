"""Test-side plumbing: loads the product package, and wraps the two CHECKERS (oracle/ C restatement, tests/emu lane
emulator) with ctypes. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use this module."""
from __future__ import annotations

import ctypes as C
import hashlib
import importlib
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

pa = importlib.import_module("rust-pseudoaligner_amd")
_build = importlib.import_module("rust-pseudoaligner_amd._build")


def _load_recipe(path, name):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, str(path))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


_oracle_build = _load_recipe(ROOT / "oracle" / "build.py", "pa_oracle_build")
_emu_build = _load_recipe(ROOT / "tests" / "emu" / "build.py", "pa_emu_build")


def build_all(force: bool = False):
    """product (hipcc, gfx950) + the two checkers (gcc / g++)"""
    return _build.build_product(force), _oracle_build.build_oracle(force), _emu_build.build_emu(force)

GOLDEN = ROOT / "tests" / "golden"
FASTA = GOLDEN / "gencode_small.fa"
FASTQ = GOLDEN / "small.fq"

ORACLE_RESULT = np.dtype([("mapped", "<u4"), ("coverage", "<u4"), ("mismatches", "<u4"), ("class_len", "<u4")])
COUNTER_NAMES = ["reads", "mapped", "probes", "node_visits", "bases_compared", "class_sizes", "result_sizes",
                 "left_extensions", "reseeks"]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in COUNTER_NAMES]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in COUNTER_NAMES}


_oracle_lib = None
_emu_lib = None


def oracle_lib():
    global _oracle_lib
    if _oracle_lib is None:
        so = _oracle_build.build_oracle()
        L = C.CDLL(str(so))
        L.oracle_index_new.restype = C.c_void_p
        L.oracle_index_new.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_uint32, C.c_void_p, C.c_void_p]
        L.oracle_index_free.argtypes = [C.c_void_p]
        L.oracle_last_error.restype = C.c_char_p
        L.oracle_intersect.restype = C.c_size_t
        L.oracle_intersect.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.oracle_map_read.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        for name in ("oracle_map_batch", "oracle_map_batch_tiles"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int,
                                         C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p]
        L.oracle_free.argtypes = [C.c_void_p]
        L.oracle_last_batch_seconds.restype = C.c_double
        L.oracle_lookup_kmer.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
        _oracle_lib = L
    return _oracle_lib


def emu_lib():
    global _emu_lib
    if _emu_lib is None:
        so = _emu_build.build_emu()
        L = C.CDLL(str(so))
        L.emu_index_new.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.emu_index_free.argtypes = [C.c_void_p]
        L.emu_last_error.restype = C.c_char_p
        L.emu_index_info.restype = C.c_uint64
        L.emu_index_info.argtypes = [C.c_void_p, C.c_int]
        L.emu_map_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p,
                                    C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.emu_free.argtypes = [C.c_void_p]
        _emu_lib = L
    return _emu_lib


def pack_read(seq: str) -> np.ndarray:
    """ASCII -> packed words (+2 pad words), LSB-first; non-ACGT -> A."""
    lut = np.zeros(256, np.uint64)
    for ch, v in (("C", 1), ("G", 2), ("T", 3), ("c", 1), ("g", 2), ("t", 3)):
        lut[ord(ch)] = v
    codes = lut[np.frombuffer(seq.encode(), np.uint8)]
    words = np.zeros((len(seq) + 31) // 32 + 2, np.uint64)
    idx = np.arange(len(seq))
    np.bitwise_or.at(words, idx >> 5, codes << ((idx & 31) * 2).astype(np.uint64))
    return words


def pack_reads_tiles(reads, words_per_read=None):
    """Independent (numpy) packer of the checker: ASCII reads -> (tiles, lens, words_per_read) in the tile layout the oracle's
    batch entry reads (word-major tiles of 64 reads: tiles[(t*W + w)*64 + r], 2 bits per base LSB-first, A=0 C=1 G=2 T=3,
    anything else -> A, as src/pseudoaligner.rs:450 / SURVEY appendix A). Shares no code with the product's encoders, so a
    packing bug common to pa_encode_reads_host and pa_encode_kernel cannot hide in the batch parity tests."""
    bs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
    n = len(bs)
    lens = np.array([len(b) for b in bs], np.uint32)
    wpr = int(words_per_read or max(1, (int(lens.max()) + 31) // 32 if n else 1))
    tiles = np.zeros(((n + 63) // 64) * wpr * 64, np.uint64)
    lut = np.zeros(256, np.uint64)
    for ch, v in ((b"C", 1), (b"G", 2), (b"T", 3), (b"c", 1), (b"g", 2), (b"t", 3)):
        lut[ch[0]] = v
    shifts = (2 * np.arange(32, dtype=np.uint64))[None, None, :]
    t3 = tiles.reshape(-1, wpr, 64)
    for lo in range(0, n, 65536):
        hi = min(n, lo + 65536)
        m = np.zeros((hi - lo, wpr * 32), np.uint8)
        flat = np.frombuffer(b"".join(bs[lo:hi]), np.uint8)
        ll = lens[lo:hi].astype(np.int64)
        row = np.repeat(np.arange(hi - lo), ll)
        col = np.arange(len(flat)) - np.repeat(np.cumsum(ll) - ll, ll)
        keep = col < wpr * 32
        m[row[keep], col[keep]] = flat[keep]
        words = (lut[m].reshape(hi - lo, wpr, 32) << shifts).sum(axis=2, dtype=np.uint64)      # disjoint bit fields: sum == or
        rid = np.arange(lo, hi)
        t3[rid >> 6, :, rid & 63] = words
    return tiles, lens, wpr


class Oracle:
    """oracle/pa_oracle.c over the flat arrays of a HostIndex (graph + classes only: the oracle builds its own
    dictionary and edges)."""

    def __init__(self, host_index):
        self.host = host_index
        a = host_index.arrays()
        self.a = a
        L = oracle_lib()
        self._h = L.oracle_index_new(a["k"], a["num_nodes"], a["node_seq"].ctypes.data, a["node_start"].ctypes.data,
                                     a["node_len"].ctypes.data, a["node_exts"].ctypes.data, a["node_colour"].ctypes.data,
                                     a["num_classes"], a["ec_offset"].ctypes.data, a["ec_ids"].ctypes.data)
        if not self._h:
            raise RuntimeError("oracle_index_new: %s" % L.oracle_last_error().decode())
        self.max_class = int((a["ec_offset"][1:] - a["ec_offset"][:-1]).max()) if a["num_classes"] else 1

    def map_read(self, seq: str, allowed: int = 2):
        """-> (rc, class ids, coverage, mismatches, nodes in visit order); rc 1 = Some, 0 = None"""
        w = pack_read(seq)
        cls = np.zeros(max(self.max_class, 1), np.uint32)
        nodes = np.zeros(2 * len(seq) + 8, np.uint32)
        cl, cov, mm, nn = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        rc = oracle_lib().oracle_map_read(self._h, w.ctypes.data, len(seq), allowed, cls.ctypes.data, len(cls), C.byref(cl),
                                          C.byref(cov), C.byref(mm), nodes.ctypes.data, len(nodes), C.byref(nn), None)
        assert rc >= 0, rc
        return rc, cls[: cl.value].tolist(), cov.value, mm.value, nodes[: nn.value].tolist()

    def map_tiles(self, tiles, lens, wpr, allowed=2, nthreads=1):
        n = len(lens)
        res = np.zeros(n, ORACLE_RESULT)
        coff = np.zeros(n + 1, np.uint64)
        ids = C.c_void_p()
        ctr = Counters()
        lens = np.ascontiguousarray(lens, np.uint32)
        rc = oracle_lib().oracle_map_batch_tiles(self._h, tiles.ctypes.data, wpr, lens.ctypes.data, n, allowed, nthreads, res.ctypes.data,
                                                 coff.ctypes.data, C.byref(ids), C.byref(ctr))
        assert rc == 0, rc
        self.last_seconds = oracle_lib().oracle_last_batch_seconds()
        total = int(coff[-1])
        out = np.frombuffer((C.c_uint32 * max(total, 1)).from_address(ids.value), np.uint32)[:total].copy()
        oracle_lib().oracle_free(ids)
        return res, coff, out, ctr.as_dict()

    def map_reads(self, reads, allowed=2, nthreads=1):
        tiles, lens, wpr = pack_reads_tiles(reads)   # the checker's own packer, not the product's encoder
        return self.map_tiles(tiles, lens, wpr, allowed, nthreads)

    def lookup(self, kmer: int):
        n, o = C.c_uint32(), C.c_uint32()
        ok = oracle_lib().oracle_lookup_kmer(self._h, kmer & 0xFFFFFFFFFFFFFFFF, kmer >> 64, C.byref(n), C.byref(o))
        return (n.value, o.value) if ok else None

    def __del__(self):
        try:
            if self._h:
                oracle_lib().oracle_index_free(self._h)
                self._h = None
        except Exception:
            pass


def oracle_intersect(v1, v2):
    a = np.array(v1, np.uint32)
    b = np.array(v2, np.uint32)
    n = oracle_lib().oracle_intersect(a.ctypes.data if len(a) else None, len(a), b.ctypes.data if len(b) else None, len(b))
    return a[:n].tolist()


class Emu:
    """tests/emu: the product's lane state machine + GPU index layout executed on the host."""

    def __init__(self, host_index, threads=4):
        self.host = host_index
        flat = host_index.flat()
        h = C.c_void_p()
        rc = emu_lib().emu_index_new(C.byref(flat), threads, C.byref(h))
        if rc != 0:
            raise RuntimeError("emu_index_new: %s" % emu_lib().emu_last_error().decode())
        self._h = h

    def info(self):
        L = emu_lib()
        return dict(num_kmers=L.emu_index_info(self._h, 0), nbuckets=L.emu_index_info(self._h, 1), blob_bytes=L.emu_index_info(self._h, 2),
                    max_class_len=L.emu_index_info(self._h, 3), num_chains=L.emu_index_info(self._h, 4), num_segs=L.emu_index_info(self._h, 5),
                    bad_blocks=L.emu_index_info(self._h, 6), branch_records=L.emu_index_info(self._h, 7),
                    num_bitmaps=L.emu_index_info(self._h, 8), bitmap_min=L.emu_index_info(self._h, 9))

    def map_tiles(self, tiles, lens, wpr, allowed=2, col_cap=8, want_nodes=False):
        n = len(lens)
        res = np.zeros(n, pa.RESULT_DTYPE)
        coff = np.zeros(n + 1, np.uint64)
        ids = C.c_void_p()
        colour = np.zeros(max(n, 1), np.uint32)
        steps = np.zeros(5, np.uint64)
        lens = np.ascontiguousarray(lens, np.uint32)
        stride = 2 * int(lens.max() if n else 1) + 8
        nodes = np.zeros((n, stride), np.uint32) if want_nodes else None
        nlen = np.zeros(max(n, 1), np.uint32)
        rc = emu_lib().emu_map_batch(self._h, tiles.ctypes.data, wpr, lens.ctypes.data, n, allowed, col_cap, res.ctypes.data, coff.ctypes.data,
                                     C.byref(ids), colour.ctypes.data, steps.ctypes.data, nodes.ctypes.data if want_nodes else None, stride,
                                     nlen.ctypes.data)
        assert rc == 0, (rc, emu_lib().emu_last_error())
        total = int(coff[-1])
        out = np.frombuffer((C.c_uint32 * max(total, 1)).from_address(ids.value), np.uint32)[:total].copy()
        emu_lib().emu_free(ids)
        return dict(results=res, coff=coff, ids=out, colour=colour[:n], steps=steps, nodes=nodes, nodes_len=nlen[:n])

    def __del__(self):
        try:
            if self._h:
                emu_lib().emu_index_free(self._h)
                self._h = None
        except Exception:
            pass


def read_fastq(path=FASTQ):
    ids, seqs = [], []
    with open(path) as f:
        lines = f.read().split("\n")
    for i in range(0, len(lines) - 3, 4):
        ids.append(lines[i][1:].split()[0])
        seqs.append(lines[i + 1])
    return ids, seqs


def read_fasta(path=FASTA):
    names, seqs, cur = [], [], []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line.startswith(">"):
                if names:
                    seqs.append("".join(cur))
                names.append(line[1:])
                cur = []
            elif line:
                cur.append(line)
    seqs.append("".join(cur))
    return names, seqs


def result_lines(ids, mapped, cov, mm, coff, cids):
    """The line format hashed in SURVEY.md appendix B."""
    out = []
    for i, rid in enumerate(ids):
        if not mapped[i]:
            out.append("%s\tNone\n" % rid)
        else:
            cl = ",".join(str(int(x)) for x in cids[int(coff[i]):int(coff[i + 1])])
            out.append("%s\t%s\t%d\t%d\n" % (rid, cl, cov[i], mm[i]))
    return out


def sha256_lines(lines):
    return hashlib.sha256("".join(lines).encode()).hexdigest()


def assert_same_as_oracle(got_results, got_coff, got_ids, o_res, o_coff, o_ids, what=""):
    """Bit-exact comparison of a product-format result (bit 31 of mismatches = mapped) with the oracle's."""
    mapped = (got_results["mismatches"] >> 31).astype(np.uint32)
    mm = got_results["mismatches"] & np.uint32(0x7FFFFFFF)
    bad = np.nonzero((mapped != o_res["mapped"]) | (got_results["coverage"] != o_res["coverage"]) | (mm != o_res["mismatches"]) |
                     (got_results["class_len"] != o_res["class_len"]))[0]
    assert len(bad) == 0, "%s: %d reads differ in (mapped, coverage, mismatches, class_len); first: read %d got %s oracle %s" % (
        what, len(bad), bad[0], (mapped[bad[0]], got_results[bad[0]]), o_res[bad[0]])
    assert np.array_equal(got_coff, o_coff), what + ": class offsets differ"
    assert np.array_equal(got_ids, o_ids), what + ": class ids differ"


def counts_reference(results, coff, cids, host_index):
    """Checker for the class-count table (pa_counts_*): counts[c] = reads whose class equals index class c, then
    [novel non-empty, mapped-but-empty, unmapped]."""
    a = host_index.arrays()
    nc = a["num_classes"]
    off = a["ec_offset"].astype(np.int64)
    table = {tuple(a["ec_ids"][off[c]:off[c + 1]].tolist()): c for c in range(nc)}
    counts = np.zeros(nc + 3, np.int64)
    mapped = (results["mismatches"] >> 31).astype(bool) if "mapped" not in results.dtype.names else results["mapped"].astype(bool)
    for i in range(len(results)):
        if not mapped[i]:
            counts[nc + 2] += 1
        elif results["class_len"][i] == 0:
            counts[nc + 1] += 1
        else:
            c = table.get(tuple(cids[int(coff[i]):int(coff[i + 1])].tolist()))
            counts[nc if c is None else c] += 1
    return counts


def novel_reference(results, coff, cids, host_index):
    """Checker for the overflow table (pa_overflow_*): {tuple(ids): reads} over the mapped reads whose non-empty class is
    NO index class — the reads the dense table lumps into counts[num_classes]."""
    a = host_index.arrays()
    off = a["ec_offset"].astype(np.int64)
    known = {tuple(a["ec_ids"][off[c]:off[c + 1]].tolist()) for c in range(a["num_classes"])}
    mapped = (results["mismatches"] >> 31).astype(bool) if "mapped" not in results.dtype.names else results["mapped"].astype(bool)
    out = {}
    for i in np.flatnonzero(mapped & (results["class_len"] > 0)):
        ids = tuple(cids[int(coff[i]):int(coff[i + 1])].tolist())
        if ids not in known:
            out[ids] = out.get(ids, 0) + 1
    return out


def counts_reference_fast(results, coff, cids, host_index):
    """counts_reference for millions of reads: reads are grouped by (class length, 64-bit content hash), every group is checked
    id by id against its first member (so a hash collision cannot merge two classes), and only one dictionary lookup per
    group is done in Python."""
    a = host_index.arrays()
    nc = a["num_classes"]
    off = a["ec_offset"].astype(np.int64)
    ec_ids = a["ec_ids"]
    n = len(results)
    coff = np.asarray(coff, np.int64)
    cids = np.asarray(cids, np.uint32)
    mapped = (results["mismatches"] >> 31).astype(bool) if "mapped" not in results.dtype.names else results["mapped"].astype(bool)
    clen = (coff[1:] - coff[:-1])[:n]
    counts = np.zeros(nc + 3, np.int64)
    counts[nc + 2] = int((~mapped).sum())
    counts[nc + 1] = int((mapped & (clen == 0)).sum())
    sel = np.flatnonzero(mapped & (clen > 0))
    if len(sel) == 0:
        return counts

    def content_hash(ids, starts, lens_):
        x = ids.astype(np.uint64)
        x = (x + np.uint64(0x9E3779B97F4A7C15)) * np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(31)
        x *= np.uint64(0x94D049BB133111EB)
        within = np.arange(len(ids), dtype=np.uint64) - np.repeat(starts.astype(np.uint64), lens_)
        x *= (within * np.uint64(2) + np.uint64(1))                  # position-dependent: order matters
        csum = np.concatenate([[np.uint64(0)], np.cumsum(x, dtype=np.uint64)])
        return csum[starts + lens_] - csum[starts]

    with np.errstate(over="ignore"):
        h_read = content_hash(cids, coff[sel], clen[sel])
        h_cls = content_hash(ec_ids, off[:-1], off[1:] - off[:-1])
    key = np.stack([h_read, clen[sel].astype(np.uint64)], axis=1)
    uniq, first, inv = np.unique(key, axis=0, return_index=True, return_inverse=True)
    inv = inv.reshape(-1)
    # exactness: every read of a group has the ids of the group's first member
    rep = sel[first][inv]
    owner = np.repeat(np.arange(len(sel)), clen[sel])
    within = np.arange(int(clen[sel].sum())) - np.repeat(np.cumsum(clen[sel]) - clen[sel], clen[sel])
    assert np.array_equal(cids[coff[sel][owner] + within], cids[coff[rep][owner] + within]), "hash collision between result classes"
    table = {}
    cl_len = off[1:] - off[:-1]
    for c in np.flatnonzero(np.isin(h_cls, uniq[:, 0])):
        table.setdefault((int(h_cls[c]), int(cl_len[c])), []).append(int(c))
    group_class = np.full(len(uniq), nc, np.int64)
    for g in range(len(uniq)):
        r = sel[first[g]]
        ids = cids[coff[r]:coff[r + 1]]
        for c in table.get((int(uniq[g, 0]), int(uniq[g, 1])), ()):
            if np.array_equal(ec_ids[off[c]:off[c + 1]], ids):
                group_class[g] = c
                break
    np.add.at(counts, group_class[inv], 1)
    return counts


def random_txome_case(seed, tmp_path, big=False, max_read=250):
    """differential-fuzz input: a small random transcriptome built from shared segments (repeats, cycles on a two-letter
    alphabet, transcripts shorter than k), a k from 8 to 64, 0..3 allowed mismatches and 3000 reads: substrings with
    substitutions, chimeras of two transcripts, random sequence, N and lower case. Returns (host index or None, k, reads,
    reads as the reference encodes them, allowed)."""
    rng = np.random.RandomState(100 + seed)
    k = int(rng.choice([8, 11, 16, 21, 32, 33, 47, 64]))
    alphabet = "ACGT" if seed % 3 else "AC"
    # big: hundreds of transcripts over few segments: classes of many ids spread over more than two 32-id windows, i.e.
    # list mode with long lists, many classes per read, bases of more than 8 ids
    nseg, ntx = (int(rng.randint(12, 40)), int(rng.randint(150, 500))) if big else (16, int(rng.randint(3, 60)))
    segs = ["".join(rng.choice(list(alphabet), rng.randint(5, 90))) for _ in range(nseg)]
    txs = ["".join(segs[j] for j in rng.choice(len(segs), rng.randint(1, 9))) for _ in range(ntx)]
    fa = tmp_path / "r.fa"
    fa.write_text("".join(">t%d|g%d\n%s\n" % (i, i // 3, s) for i, s in enumerate(txs)))
    host = pa.HostIndex.build_fasta(str(fa), k, 3)
    if host.arrays()["num_nodes"] == 0:
        return None, k, [], [], 0
    reads = []
    for _ in range(3000):
        kind = rng.randint(0, 10)
        t = txs[rng.randint(len(txs))]
        if kind < 6:                                   # substring with substitutions
            lo = rng.randint(0, max(1, len(t) - 10))
            r = list(t[lo:lo + rng.randint(1, max(2, max_read * 4 // 5))])
            for j in range(len(r)):
                if rng.rand() < 0.02:
                    r[j] = "ACGT"[rng.randint(4)]
            reads.append("".join(r))
        elif kind < 8:                                 # chimera
            u = txs[rng.randint(len(txs))]
            reads.append((t[: rng.randint(1, len(t) + 1)] + u[rng.randint(0, len(u)):] + (t + u) * (max_read // 250 - 1))[:max_read])
        elif kind == 8:
            reads.append("".join(rng.choice(list(alphabet), rng.randint(0, 150))))
        else:
            r = t[: rng.randint(1, len(t) + 1)]
            reads.append((r[: len(r) // 2] + "N" + r[len(r) // 2 + 1:]).lower() if rng.rand() < 0.5 else r.lower())
    allowed = int(rng.randint(0, 4))
    return host, k, reads, [r.upper().replace("N", "A") for r in reads], allowed


def many_classes_case(seed, tmp_path, k=11, ntx=300, read_len=1500, nreads=400, ordered=False):
    """list-mode stress: transcripts that are long chains of short shared segments, so that every node is in ~half of the
    transcripts (classes far beyond two windows) and a long read crosses more than 64 nodes of DIFFERENT classes.
    Returns (host index, reads)."""
    rng = np.random.RandomState(7000 + seed)
    if ordered:   # transcripts = windows of ONE long cycle of segments: every node is in ~90 transcripts, a sliding set, so
                  # that even the shortest class of a read has far more than 8 ids (the cooperative step, > 64 lists)
        segs = ["".join(rng.choice(list("ACGT"), rng.randint(12, 22))) for _ in range(400)]
        txs = []
        for _ in range(ntx):
            st = rng.randint(len(segs))
            txs.append("".join(segs[(st + j) % len(segs)] for j in range(120)))
    else:
        segs = ["".join(rng.choice(list("ACGT"), rng.randint(12, 22))) for _ in range(90)]
        txs = ["".join(segs[j] for j in rng.choice(len(segs), 120)) for _ in range(ntx)]
    fa = tmp_path / "mc.fa"
    fa.write_text("".join(">t%d|g%d\n%s\n" % (i, i // 4, s) for i, s in enumerate(txs)))
    host = pa.HostIndex.build_fasta(str(fa), k, 4)
    reads = []
    for _ in range(nreads):
        t = txs[rng.randint(len(txs))]
        n = int(rng.randint(40, read_len + 1))               # tens of classes for the short reads, hundreds for the long
        lo = rng.randint(0, len(t) - n)
        r = list(t[lo:lo + n])
        for j in range(len(r)):
            if rng.rand() < 0.002:
                r[j] = "ACGT"[rng.randint(4)]
        reads.append("".join(r))
    return host, reads


def thousands_of_classes_case(tmp_path, k=20, length=16383, step=3):
    """ONE long transcript T and every suffix T[step*i:] as a transcript of its own: the k-mer at position p belongs to the
    transcripts {i : step*i <= p}, so the colour changes every `step` positions and a read = T passes ~length/step unitigs
    with pairwise DIFFERENT classes (5 400 for the defaults: more than the 12-bit class counter of round 2 could hold;
    the reference has no limit). Returns (host index, [reads])."""
    rng = np.random.RandomState(77)
    T = "".join("ACGT"[i] for i in rng.randint(0, 4, length))
    p = tmp_path / "suffixes.fa"
    with open(p, "w") as f:
        i = 0
        while step * i + k <= length:
            f.write(">s%d gene=G\n%s\n" % (i, T[step * i:]))
            i += 1
    host = pa.build_index(str(p), k, 8)
    reads = [T, T[5000:], T[: 3 * k], T[100:9000], T[:6000] + "A" + T[6001:]]
    return host, reads


def long_chain_case(seed, tmp_path):
    """fuzz family for the LEFT path across chain blocks: long transcripts, nested suffix / infix transcripts whose cuts (colour-only
    node boundaries) fall on and around multiples of 64, reads of 300..500 bases with dense errors at their start (so that the first
    hit is far into the read and the left extension walks > 192 bases), allowed in {6, 12}."""
    rng = np.random.RandomState(9000 + seed)
    k = int(rng.choice([16, 24, 31, 40]))
    base = ["".join(rng.choice(list("ACGT"), rng.randint(600, 1500))) for _ in range(3)]
    txs = list(base)
    for b in base:
        for _ in range(rng.randint(2, 6)):
            cut = int(rng.choice([64, 128, 192, 256, 320, 63, 65, 127, 129, rng.randint(1, len(b) - k - 1)]))
            cut = min(cut, len(b) - k - 1)
            txs.append(b[cut:] if rng.rand() < 0.7 else b[cut: cut + rng.randint(k + 1, len(b) - cut)])
    fa = tmp_path / ("lc%d.fa" % seed)
    fa.write_text("".join(">t%d|g%d\n%s\n" % (i, i, s) for i, s in enumerate(txs)))
    host = pa.HostIndex.build_fasta(str(fa), k, 3)
    allowed = int(rng.choice([6, 12]))
    reads = []
    for _ in range(300):
        t = txs[rng.randint(len(txs))]
        n = int(rng.randint(300, 501))
        if len(t) < n + 2:
            t = base[rng.randint(3)]
        lo = rng.randint(0, len(t) - n)
        r = list(t[lo:lo + n])
        head = rng.randint(100, n - k - 5)            # errors spaced closer than k over the head: first hit behind it
        step = rng.randint(max(2, k // 2), k)
        for j in range(rng.randint(0, step), head, step):
            r[j] = "ACGT"[("ACGT".index(r[j]) + 1 + rng.randint(3)) % 4]
        for j in range(head, n):
            if rng.rand() < 0.01:
                r[j] = "ACGT"[rng.randint(4)]
        reads.append("".join(r))
    return host, reads, allowed


# ---- foreign flat indexes (the arrays a Rust exporter hands over are NOT laid out by the product's builders) ----
def unpack_bases(words, n):
    """2-bit packed words (LSB-first) -> uint8 codes[n]"""
    pos = np.arange(n, dtype=np.int64)
    return ((np.asarray(words)[pos >> 5] >> ((pos & 31) * 2).astype(np.uint64)) & np.uint64(3)).astype(np.uint8)


def pack_bases(codes):
    """uint8 codes -> packed words (+2 pad words)"""
    n = len(codes)
    padded = np.zeros(((n + 31) // 32 + 2) * 32, np.uint64)
    padded[:n] = codes
    return (padded.reshape(-1, 32) << (2 * np.arange(32, dtype=np.uint64))[None, :]).sum(axis=1, dtype=np.uint64)


def flat_from_arrays(arr, k, num_tx):
    """a FlatIndex over numpy arrays (kept alive by the returned dict), as an exporter on the Rust side would fill it
    (integration/rust/src/amd.rs: export_flat)"""
    f = pa._ffi.FlatIndex()
    keep = {n: np.ascontiguousarray(arr[n]) for n in ("node_seq", "node_start", "node_len", "node_exts", "node_colour", "ec_offset", "ec_ids")}
    f.k, f.num_nodes, f.num_classes, f.num_transcripts = k, len(keep["node_len"]), len(keep["ec_offset"]) - 1, num_tx
    f.seq_bases = int(keep["node_start"][-1])
    for n, v in keep.items():
        setattr(f, n, v.ctypes.data)
    f.node_redge = f.node_ledge = None
    return f, keep


def foreign_index(host, seed, cut_frac=0.5, max_cuts=3):
    """The SAME coloured De Bruijn graph as `host` in a layout the product's builders never produce: nodes in a random order,
    classes renumbered by a random permutation, and about `cut_frac` of the unitigs that hold more than one k-mer cut at random
    k-mer boundaries into 2..max_cuts+1 same-colour pieces that overlap by k-1 bases (non-maximal unitigs: a different builder's
    break points; every piece gets the one extension bit that leads to its neighbour piece). Returns a HostIndex made by
    pa_host_index_from_flat — the oracle and the product are then both built from THESE arrays."""
    a = host.arrays()
    k, nn, nc = int(a["k"]), int(a["num_nodes"]), int(a["num_classes"])
    rng = np.random.RandomState(seed)
    start, ln = a["node_start"].astype(np.int64), a["node_len"].astype(np.int64)
    bases = unpack_bases(a["node_seq"], int(start[-1]))
    nk = ln - k + 1
    ncut = np.where((rng.rand(nn) < cut_frac) & (nk > 1), rng.randint(1, max_cuts + 1, nn), 0)
    ncut = np.minimum(ncut, nk - 1)
    p_node, p_lo, p_hi = [], [], []          # piece = k-mers [lo, hi) of node p_node
    for n in np.flatnonzero(ncut):
        cuts = np.sort(rng.choice(np.arange(1, nk[n]), int(ncut[n]), replace=False))
        b = np.r_[0, cuts, nk[n]]
        p_node += [n] * (len(b) - 1); p_lo += b[:-1].tolist(); p_hi += b[1:].tolist()
    whole = np.flatnonzero(ncut == 0)
    p_node = np.r_[np.array(p_node, np.int64), whole]
    p_lo = np.r_[np.array(p_lo, np.int64), np.zeros(len(whole), np.int64)]
    p_hi = np.r_[np.array(p_hi, np.int64), nk[whole]]
    order = rng.permutation(len(p_node))
    p_node, p_lo, p_hi = p_node[order], p_lo[order], p_hi[order]
    p_len = p_hi - p_lo + k - 1
    p_src = start[p_node] + p_lo
    exts = a["node_exts"][p_node].astype(np.int64)
    inner_l, inner_r = p_lo > 0, p_hi < nk[p_node]
    e = np.where(inner_l, 16 << bases[np.maximum(p_src - 1, 0)].astype(np.int64), exts & 0xF0)
    e |= np.where(inner_r, 1 << bases[np.minimum(p_src + p_len, len(bases) - 1)].astype(np.int64), exts & 0x0F)
    new_start = np.zeros(len(p_node) + 1, np.uint64)
    new_start[1:] = np.cumsum(p_len)
    idx = np.repeat(p_src, p_len) + (np.arange(int(p_len.sum()), dtype=np.int64) - np.repeat(new_start[:-1].astype(np.int64), p_len))
    perm = rng.permutation(nc)                                     # class c is renumbered perm[c]
    off = a["ec_offset"].astype(np.int64)
    clen = off[1:] - off[:-1]
    inv = np.argsort(perm)                                         # new class j is old class inv[j]
    ec_offset = np.zeros(nc + 1, np.uint64)
    ec_offset[1:] = np.cumsum(clen[inv])
    gidx = np.repeat(off[inv], clen[inv]) + (np.arange(int(clen.sum()), dtype=np.int64) - np.repeat(ec_offset[:-1].astype(np.int64), clen[inv]))
    arr = dict(node_seq=pack_bases(bases[idx]), node_start=new_start, node_len=p_len.astype(np.uint32), node_exts=e.astype(np.uint8),
               node_colour=perm[a["node_colour"][p_node]].astype(np.uint32), ec_offset=ec_offset, ec_ids=a["ec_ids"][gidx].astype(np.uint32))
    f, keep = flat_from_arrays(arr, k, int(a["num_transcripts"]))
    out = pa.HostIndex.from_flat(f)
    return out, int((ncut > 0).sum())


def index_from_node_set(nodes, k, num_tx, seed=0):
    """a flat index from a SET of (sequence, class id tuple, left ext letters, right ext letters) — the form
    tests/test_reference_build_order.reference_order_nodes emulates the reference's two-pass build in — nodes in the set's
    (shuffled) order, classes numbered by first appearance"""
    rng = np.random.RandomState(seed)
    nodes = sorted(nodes)
    nodes = [nodes[i] for i in rng.permutation(len(nodes))]
    lut = {c: i for i, c in enumerate("ACGT")}
    classes, colour, exts = {}, [], []
    for seq, cls, le, re in nodes:
        colour.append(classes.setdefault(cls, len(classes)))
        exts.append(sum(16 << lut[c] for c in le) | sum(1 << lut[c] for c in re))
    ln = np.array([len(n[0]) for n in nodes], np.int64)
    start = np.zeros(len(nodes) + 1, np.uint64)
    start[1:] = np.cumsum(ln)
    l8 = np.zeros(256, np.uint8)
    for c, v in lut.items():
        l8[ord(c)] = v
    codes = l8[np.frombuffer("".join(n[0] for n in nodes).encode(), np.uint8)]
    lists = sorted(classes, key=classes.get)
    ec_offset = np.zeros(len(lists) + 1, np.uint64)
    ec_offset[1:] = np.cumsum([len(l) for l in lists])
    arr = dict(node_seq=pack_bases(codes), node_start=start, node_len=ln.astype(np.uint32), node_exts=np.array(exts, np.uint8),
               node_colour=np.array(colour, np.uint32), ec_offset=ec_offset, ec_ids=np.array([i for l in lists for i in l], np.uint32))
    f, keep = flat_from_arrays(arr, k, num_tx)
    return pa.HostIndex.from_flat(f)


def error_reads(host, read_len, n, ppm, seed):
    """n simulated reads (tiles, lens, wpr) of the index's own transcripts with ppm substitutions per million bases"""
    tx = pa.Txome.from_host_index(host)
    tiles, lens = tx.simulate_host(read_len, seed, n, ppm)
    return tiles, lens, pa.lib().pa_words_per_read(read_len)


def _mutate(rng, r, rate):
    r = list(r)
    for j in range(len(r)):
        if rng.rand() < rate:
            r[j] = "ACGT"[("ACGT".index(r[j]) + 1 + rng.randint(3)) % 4]
    return r


def branch_case(seed, tmp_path, nreads=600):
    """fuzz family: BRANCH POINTS on and around multiples of 64 with skewed multiplicities, and bubbles. Every locus is
    prefix + one of 2..4 branches + (for a bubble) a common suffix; the prefix's length puts the branch point at 64 j + d, d in -2..2, of
    its unitig (the chain blocks' seams); branch 0 is carried by most transcripts (the flattener's FAVOURED branch: its copy follows the
    branch record), the others by one or two. Reads start before a branch point and run over it along favoured and other branches, some
    to 700 bases over several loci, with clustered errors (2..5 substitutions inside ten bases), dense errors at their head, or none;
    allowed in {0, 1, 2, 3, 6, 12}; K in {12 .. 64}. Returns (host index, reads, allowed)."""
    rng = np.random.RandomState(31000 + seed)
    k = int(rng.choice([12, 16, 21, 24, 31, 32, 33, 47, 64]))
    rnd = lambda n: "".join(rng.choice(list("ACGT"), n))
    nloci = int(rng.randint(2, 5))
    loci = []
    for _ in range(nloci):
        j, d = int(rng.randint(1, 5)), int(rng.randint(-2, 3))
        plen = max(k + 1, 64 * j + d + int(rng.choice([0, k - 1, k])))     # the branch point (or its last k-mer) on / beside a multiple of 64
        prefix = rnd(plen)
        nb = int(rng.randint(2, 5))
        kind = rng.randint(3)                                               # 0: branches of any length; 1: SNP bubble; 2: short indel-like bubble
        if kind == 1:
            base = rnd(int(rng.randint(1, 40)))
            branches = [("ACGT"[i] + base) for i in rng.permutation(4)[:nb]]
        elif kind == 2:
            branches = [rnd(int(rng.randint(0, 6))) for _ in range(nb)]
            branches = list(dict.fromkeys(branches))
        else:
            branches = [rnd(int(rng.randint(1, 150))) for _ in range(nb)]
        suffix = rnd(int(rng.randint(k, 200))) if rng.rand() < 0.7 else ""   # a bubble closes again, a fork does not
        mult = [int(rng.randint(4, 12))] + [int(rng.randint(1, 3)) for _ in branches[1:]]
        loci.append((prefix, branches, suffix, mult))
    txs = []
    ntx = int(rng.randint(6, 30))
    for t in range(ntx):
        s = ""
        for prefix, branches, suffix, mult in loci:
            w = np.array(mult[:len(branches)], float)
            b = rng.choice(len(branches), p=w / w.sum())
            s += prefix + branches[b] + suffix
            if rng.rand() < 0.3:
                break
        txs.append(s)
    fa = tmp_path / ("br%d.fa" % seed)
    fa.write_text("".join(">t%d|g%d\n%s\n" % (i, i // 2, s) for i, s in enumerate(txs)))
    host = pa.HostIndex.build_fasta(str(fa), k, 3)
    if host.arrays()["num_nodes"] == 0:
        return None, [], 0
    allowed = int(rng.choice([0, 1, 2, 3, 6, 12]))
    reads = []
    for _ in range(nreads):
        t = txs[rng.randint(len(txs))]
        n = int(rng.randint(k, min(len(t), 700) + 1)) if len(t) >= k else len(t)
        lo = rng.randint(0, len(t) - n + 1)
        r = list(t[lo:lo + n])
        kind = rng.randint(4)
        if kind == 0 and n > 12:                                            # a cluster of errors
            at = rng.randint(0, n - 10)
            for j in rng.choice(10, rng.randint(2, 6), replace=False):
                r[at + j] = "ACGT"[("ACGT".index(r[at + j]) + 1 + rng.randint(3)) % 4]
        elif kind == 1 and n > 2 * k:                                       # dense errors at the head: the first hit lies behind them
            for j in range(rng.randint(0, k // 2 + 1), min(n - k, rng.randint(k, 3 * k)), max(2, k // 2)):
                r[j] = "ACGT"[("ACGT".index(r[j]) + 1 + rng.randint(3)) % 4]
        elif kind == 2:
            r = _mutate(rng, r, 0.02)
        reads.append("".join(r))
    return host, reads, allowed


def tandem_case(seed, tmp_path, nreads=500):
    """fuzz family: TANDEM REPEATS and low-complexity sequence over 1..4-letter alphabets — k-mer cycles, self-loops, nodes
    of exactly one k-mer — with reads of exactly K, K + 1, K + 2 ... 1 200 bases. Transcripts are runs of a short unit (1..12 bases) of
    different lengths between random flanks, and some pure runs. K in {8 .. 64}. Returns (host index, reads, allowed)."""
    rng = np.random.RandomState(47000 + seed)
    k = int(rng.choice([8, 11, 16, 20, 24, 31, 32, 33, 48, 64]))
    alpha = list("ACGT"[: int(rng.randint(1, 5))])
    rnd = lambda n, a=alpha: "".join(rng.choice(a, n))
    units = [rnd(int(rng.randint(1, 13))) for _ in range(int(rng.randint(1, 4)))]
    txs = []
    for _ in range(int(rng.randint(3, 25))):
        u = units[rng.randint(len(units))]
        run = (u * (int(rng.randint(k, 4 * k + 200)) // len(u) + 2))[: int(rng.randint(k, 4 * k + 200))]
        kind = rng.randint(4)
        if kind == 0:
            txs.append(run)                                                   # a pure run (a k-mer cycle when it is long enough)
        elif kind == 1:
            txs.append(rnd(int(rng.randint(0, 80)), list("ACGT")) + run + rnd(int(rng.randint(0, 80)), list("ACGT")))
        elif kind == 2:
            v = units[rng.randint(len(units))]
            txs.append(run + (v * 60)[: int(rng.randint(k, 300))] + run[: int(rng.randint(1, len(run) + 1))])
        else:
            txs.append("".join(_mutate(rng, run, 0.01)))                      # an imperfect repeat
    fa = tmp_path / ("td%d.fa" % seed)
    fa.write_text("".join(">t%d|g%d\n%s\n" % (i, i // 3, s) for i, s in enumerate(txs)))
    host = pa.HostIndex.build_fasta(str(fa), k, 3)
    if host.arrays()["num_nodes"] == 0:
        return None, [], 0
    allowed = int(rng.choice([0, 1, 2, 3]))
    reads = []
    for i in range(nreads):
        t = txs[rng.randint(len(txs))]
        if len(t) < k:
            reads.append(t)
            continue
        n = k + i % 3 if i % 4 == 0 else int(rng.randint(k, min(len(t), 1200) + 1))   # exactly K, K + 1, K + 2 bases, and anything up to 1 200
        n = min(n, len(t))
        lo = rng.randint(0, len(t) - n + 1)
        r = list(t[lo:lo + n])
        if rng.rand() < 0.4:
            r = _mutate(rng, r, 0.03)
        reads.append("".join(r))
    return host, reads, allowed


def long_transcript_case(tmp_path, k=24, long_len=120000, seed=5):
    """validate_dbg's self-mapping property (src/build_index.rs:300-367) beyond 2^14 bases: a small synthetic transcriptome plus two LONG
    transcripts — one of `long_len` random bases, one that shares stretches with it and with the short ones (so that the walk crosses many
    nodes and classes) — every transcript is also a read. Returns (host index, transcript strings)."""
    rng = np.random.RandomState(seed)
    tx = pa.Txome.synthesize(40, 130, 11)
    packed, tx_start = tx.arrays()
    all_codes = unpack_bases(packed, int(tx_start[-1]))
    lut = np.frombuffer(b"ACGT", np.uint8)
    seqs = [lut[all_codes[int(tx_start[t]):int(tx_start[t + 1])]].tobytes().decode() for t in range(len(tx_start) - 1)]
    big = "".join(rng.choice(list("ACGT"), long_len))
    pieces = [big[20000:52000], seqs[3], big[70000:100000], seqs[7][:900], "".join(rng.choice(list("ACGT"), 9000)), big[5000:9000]]
    seqs += [big, "".join(pieces)]
    fa = tmp_path / "long.fa"
    fa.write_text("".join(">t%d|g%d\n%s\n" % (i, i // 3, s) for i, s in enumerate(seqs)))
    return pa.HostIndex.build_fasta(str(fa), k, 4), seqs
