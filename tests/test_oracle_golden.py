"""Pins the oracle (oracle/pa_oracle.c) and the CPU index builder against everything the reference's own tests hold for
this path: the test_alignment literals (src/build_index.rs:429-441), validate_dbg's properties on test/gencode_small.fa
(src/build_index.rs:262-368), plus the independent Python model of SURVEY.md appendix B and the committed fixtures."""
import numpy as np
import pytest

import helpers

# SHA-256 of the per-read result lines of an independent (throw-away Python) model of the path
APPENDIX_B = {
    20: "ba9f91d597f6a78b4178e912a615ee55401c2ce6aee7121c35cef52094f905ff",
    24: "39e558dc51737e8c558b375a4ceef6dd8a70e38d67ed0614e1970e3ff4f2a266",
}
APPENDIX_B_SUBSETS_K20 = {"exact": "c8c29204872334b5", "err": "92ebc1bbf008a6b6", "rev": "c8644efa8eb73781"}

EX1 = "GGCTGTCAACCAGTCCATAGGCAGGGCCATCAGGCACCAAAGGGATTCTGCCAGCATAGT"          # src/build_index.rs:430
SINGLE_SNP = "GGCTGTCAACCAGTCCATAGGCGGGGCCATCAGGCACCAAAGGGATTCTGCCAGCATAGT"   # src/build_index.rs:437


def test_alignment_literals(small_index):
    o = helpers.Oracle(small_index(20))
    rc, cls, cov, mm, _ = o.map_read(EX1)
    assert (rc, cls, cov) == (1, [1, 30], len(EX1))          # :433-434
    rc, cls, cov, mm, _ = o.map_read(SINGLE_SNP)
    assert (rc, cls, cov, mm) == (1, [1, 30], len(SINGLE_SNP), 1)   # :440-441


@pytest.mark.parametrize("k", [20, 24])
def test_small_fq_matches_independent_model(small_index, k):
    ids, seqs = helpers.read_fastq()
    res, coff, cids, ctr = helpers.Oracle(small_index(k)).map_reads(seqs, 2, 4)
    lines = helpers.result_lines(ids, res["mapped"], res["coverage"], res["mismatches"], coff, cids)
    assert helpers.sha256_lines(lines) == APPENDIX_B[k]
    if k == 20:
        sub = {"exact": [l for l, i in zip(lines, ids) if "_err" not in i and not i.endswith("_rev")],
               "err": [l for l, i in zip(lines, ids) if "_err" in i], "rev": [l for l, i in zip(lines, ids) if i.endswith("_rev")]}
        for name, want in APPENDIX_B_SUBSETS_K20.items():
            assert helpers.sha256_lines(sub[name])[:16] == want
        # appendix B counters of the same model
        assert (ctr["probes"], ctr["node_visits"], ctr["left_extensions"], ctr["reseeks"]) == (53648, 10329, 569, 28)
        assert int(res["mapped"].sum()) == 6211 and int(res["coverage"].sum()) == 369128


@pytest.mark.parametrize("k", [20, 24, 31])
def test_small_fq_matches_committed_fixture(small_index, k):
    ids, seqs = helpers.read_fastq()
    res, coff, cids, _ = helpers.Oracle(small_index(k)).map_reads(seqs, 2, 2)
    lines = helpers.result_lines(ids, res["mapped"], res["coverage"], res["mismatches"], coff, cids)
    assert "".join(lines) == (helpers.GOLDEN / ("small_fq_k%d.tsv" % k)).read_text()


KMER = np.dtype([("hi", "<u8"), ("lo", "<u8")])   # k-mers of up to 64 bases, sortable (hi first)


def _kmers_of(codes: np.ndarray, k: int) -> np.ndarray:
    out = np.zeros(len(codes) - k + 1, KMER)
    for j in range(k):
        out["lo" if j < 32 else "hi"] |= codes[j:len(codes) - k + 1 + j].astype(np.uint64) << np.uint64(2 * (j % 32))
    return out


def _codes(seq: str) -> np.ndarray:
    lut = np.zeros(256, np.uint8)
    lut[ord("C")], lut[ord("G")], lut[ord("T")] = 1, 2, 3
    return lut[np.frombuffer(seq.encode(), np.uint8)]


def _unpack(words: np.ndarray, start: int, length: int) -> np.ndarray:
    pos = np.arange(start, start + length, dtype=np.int64)
    return ((words[pos >> 5] >> ((pos & 31) * 2).astype(np.uint64)) & np.uint64(3)).astype(np.uint8)


@pytest.mark.parametrize("k", [20, 31, 64])   # the reference runs validate_dbg at K = 20 and K = 64 (src/build_index.rs:394-409)
def test_validate_dbg_kmer_colours(small_index, k):
    """validate_dbg part 1 (src/build_index.rs:263-298): for EVERY k-mer of every transcript the graph's colour list
    equals the naive list of transcripts containing it, and the graph holds no other k-mers (vectorised)."""
    _, seqs = helpers.read_fasta()
    km, tx = [], []
    for i, s in enumerate(seqs):
        if len(s) >= k:
            v = _kmers_of(_codes(s), k)
            km.append(v)
            tx.append(np.full(len(v), i, np.uint32))
    km, tx = np.concatenate(km), np.concatenate(tx)
    order = np.lexsort((tx, km["lo"], km["hi"]))
    km, tx = km[order], tx[order]
    keep = np.ones(len(km), bool)
    keep[1:] = (km[1:] != km[:-1]) | (tx[1:] != tx[:-1])        # test_eqclass.dedup() (:275)
    km, tx = km[keep], tx[keep]
    starts = np.flatnonzero(np.r_[True, km[1:] != km[:-1]])
    naive_kmers = km[starts]
    weights = np.random.RandomState(1).randint(1, 2**62, 4096, dtype=np.int64).astype(np.uint64)
    w = weights[tx % 4096] * (tx.astype(np.uint64) + np.uint64(0x9E3779B97F4A7C15))
    naive_sig = np.add.reduceat(w, starts)
    naive_len = np.diff(np.r_[starts, len(km)])

    a = small_index(k).arrays()
    ec_off, ec_ids = a["ec_offset"].astype(np.int64), a["ec_ids"]
    wc = weights[ec_ids % 4096] * (ec_ids.astype(np.uint64) + np.uint64(0x9E3779B97F4A7C15))
    class_sig = np.array([wc[ec_off[c]:ec_off[c + 1]].sum(dtype=np.uint64) for c in range(a["num_classes"])], np.uint64)
    class_len = np.diff(ec_off)
    for c in range(a["num_classes"]):                           # dbg eq classes are sorted + dedup'd (:286-295)
        lst = ec_ids[ec_off[c]:ec_off[c + 1]]
        assert len(lst) > 0 and np.all(lst[1:] > lst[:-1])
    gk, gs, gl = [], [], []
    for n in range(a["num_nodes"]):
        codes = _unpack(a["node_seq"], int(a["node_start"][n]), int(a["node_len"][n]))
        v = _kmers_of(codes, k)
        gk.append(v)
        gs.append(np.full(len(v), class_sig[a["node_colour"][n]], np.uint64))
        gl.append(np.full(len(v), class_len[a["node_colour"][n]], np.int64))
    gk, gs, gl = np.concatenate(gk), np.concatenate(gs), np.concatenate(gl)
    order = np.lexsort((gk["lo"], gk["hi"]))
    gk, gs, gl = gk[order], gs[order], gl[order]
    assert len(gk) == len(naive_kmers) and np.array_equal(gk, naive_kmers)     # same k-mer set, each exactly once
    assert np.array_equal(gl, naive_len) and np.array_equal(gs, naive_sig)     # same transcript list per k-mer


@pytest.mark.parametrize("k", [20, 31, 64])
def test_validate_dbg_self_mapping(small_index, k):
    """validate_dbg part 2 (src/build_index.rs:300-367): every transcript with len >= k maps with bases_aligned == len;
    the class is [i] or contains i; identical sequences share a class."""
    _, seqs = helpers.read_fasta()
    o = helpers.Oracle(small_index(k))
    idx = [i for i, s in enumerate(seqs) if len(s) >= k]
    res, coff, cids, _ = o.map_reads([seqs[i] for i in idx], 2, 8)
    for j, i in enumerate(idx):
        assert res["mapped"][j] == 1 and res["coverage"][j] == len(seqs[i]), (i, res[j])      # :309-310
        cls = cids[int(coff[j]):int(coff[j + 1])].tolist()
        if len(cls) > 1:
            assert i in cls                                                                     # :313
        else:
            assert cls == [i]                                                                   # :365
    # node-subset property (:329-356) for a sample of multi-member classes
    checked = 0
    for j, i in enumerate(idx):
        cls = cids[int(coff[j]):int(coff[j + 1])].tolist()
        if len(cls) < 2 or checked >= 40:
            continue
        if len(cls) == 2 and seqs[cls[0]] == seqs[cls[1]]:
            continue
        if len(seqs[i]) == min(len(seqs[x]) for x in cls):
            continue
        mine = set(o.map_read(seqs[i])[4])
        for x in cls:
            assert mine <= set(o.map_read(seqs[x])[4])
        checked += 1


def test_error_free_reads_equal_naive_intersection(small_index):
    """P5 of SURVEY.md §4: for an error-free read the class is the intersection over ALL its k-mers of the naive
    transcript lists and coverage == L — independent of how the graph was compacted."""
    k = 24
    host = small_index(k)
    _, seqs = helpers.read_fasta()
    tx = helpers.pa.Txome.from_host_index(host)
    tiles, lens = tx.simulate_host(100, 1, 1500)
    reads = helpers.pa.unpack_tiles(tiles, lens, 4)
    table = {}
    for i, s in enumerate(seqs):
        if len(s) >= k:
            for v in np.unique(_kmers_of(_codes(s), k)).tolist():
                table.setdefault(v, []).append(i)
    res, coff, cids, _ = helpers.Oracle(host).map_tiles(tiles, lens, 4, 2, 4)
    for j, r in enumerate(reads):
        want = None
        for v in _kmers_of(_codes(r), k).tolist():
            s = set(table[v])
            want = s if want is None else want & s
        assert res["mapped"][j] == 1 and res["coverage"][j] == 100 and res["mismatches"][j] == 0
        assert cids[int(coff[j]):int(coff[j + 1])].tolist() == sorted(want)
