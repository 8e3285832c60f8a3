"""CPU tier of the parity tests: the product's per-lane state machine (lane_steps.hpp) and GPU index layout
(device_flatten.cpp), executed on the host by tests/emu, against the oracle — bit exact."""
import numpy as np
import pytest

import helpers

pa = helpers.pa


def check(host, tiles, lens, wpr, allowed=2, col_cap=8, nodes=False):
    o_res, o_coff, o_ids, ctr = helpers.Oracle(host).map_tiles(tiles, lens, wpr, allowed, 4)
    r = helpers.Emu(host).map_tiles(tiles, lens, wpr, allowed, col_cap, want_nodes=nodes)
    helpers.assert_same_as_oracle(r["results"], r["coff"], r["ids"], o_res, o_coff, o_ids, "emu")
    return r, (o_res, o_coff, o_ids, ctr)


@pytest.mark.parametrize("k", [20, 24, 31])
def test_small_fq(small_index, k):
    _, seqs = helpers.read_fastq()
    tiles, lens, wpr = pa.encode_reads_host(seqs)
    check(small_index(k), tiles, lens, wpr)


@pytest.mark.parametrize("k,read_len,ppm,allowed", [(24, 100, 0, 2), (24, 150, 10000, 2), (31, 150, 10000, 2), (20, 75, 50000, 2),
                                                    (31, 150, 30000, 0), (24, 150, 30000, 1), (24, 150, 60000, 3), (32, 150, 10000, 2),
                                                    (8, 40, 20000, 2), (33, 150, 10000, 2), (48, 150, 5000, 2), (64, 150, 5000, 2),
                                                    (64, 100, 0, 2)])
def test_simulated_reads(small_index, built, k, read_len, ppm, allowed):
    host = small_index(k) if k in (20, 24, 31) else pa.build_index(str(helpers.FASTA), k, 8)
    tx = pa.Txome.from_host_index(host)
    tiles, lens = tx.simulate_host(read_len, 4, 30000, ppm)
    r, (o_res, _, _, ctr) = check(host, tiles, lens, pa.lib().pa_words_per_read(read_len), allowed)
    if ppm:
        assert ctr["reseeks"] > 0 and (k < 20 or ctr["left_extensions"] > 0)   # the error paths were exercised
    else:
        assert np.all(o_res["coverage"] == read_len)


# 12..308: seeds on which a lookup that only tried the first fingerprint match of a bucket missed k-mers (two keys of one
# bucket sharing their low 31 bits — low-complexity sequence); found by tools/gpu_soak.py
@pytest.mark.parametrize("seed", list(range(10)) + [12, 19, 31, 39, 48, 55, 96, 242, 278, 308])
def test_random_transcriptomes(tmp_path, seed):
    """differential fuzz (helpers.random_txome_case): the kernel's steps and GPU index layout on the host vs the oracle"""
    host, k, reads, clean, allowed = helpers.random_txome_case(seed, tmp_path)
    if host is None:
        pytest.skip("every transcript is shorter than k")
    tiles, lens, wpr = pa.encode_reads_host(reads)
    ctiles, clens, cwpr = pa.encode_reads_host(clean)
    assert np.array_equal(tiles, ctiles) and np.array_equal(lens, clens)      # N and lower case encode like the reference says
    check(host, tiles, lens, wpr, allowed)


@pytest.mark.parametrize("seed", range(6))
def test_random_transcriptomes_list_mode(tmp_path, seed):
    """the same fuzz on hundreds of transcripts over few shared segments: classes of many ids spread over more than two
    windows (list mode, many classes per read, bases of more than 8 ids)"""
    host, k, reads, clean, allowed = helpers.random_txome_case(seed, tmp_path, big=True)
    if host is None:
        pytest.skip("every transcript is shorter than k")
    tiles, lens, wpr = pa.encode_reads_host(reads)
    check(host, tiles, lens, wpr, allowed)


@pytest.mark.parametrize("ordered", [False, True])
def test_many_classes_per_read(tmp_path, ordered):
    """list mode with tens to hundreds of DIFFERENT classes per read (chains of short shared segments, K = 11, reads of up to
    1500 bases): class rows far into the spill area, bases of one or two ids"""
    host, reads = helpers.many_classes_case(0, tmp_path, nreads=150, ordered=ordered)
    tiles, lens, wpr = pa.encode_reads_host(reads)
    check(host, tiles, lens, wpr, 2)


def test_thousands_of_classes_per_read(tmp_path):
    """a 16 383-base read that passes ~5 400 unitigs with pairwise different classes (the class counter of the lane
    state had 12 bits and turned such a read into PA_ERR_INTERNAL for its whole batch; it has 14 now = the read length limit)"""
    host, reads = helpers.thousands_of_classes_case(tmp_path)
    assert host.arrays()["num_classes"] > 5000
    tiles, lens, wpr = pa.encode_reads_host(reads)
    r, (o_res, o_coff, o_ids, ctr) = check(host, tiles, lens, wpr, 2)
    assert ctr["node_visits"] > 4 * 4096 // 2 and o_res["mapped"].all()
    assert o_ids[o_coff[0]:o_coff[1]].tolist() == [0]          # only the whole transcript holds every k-mer of itself


def test_ragged_short_and_unmappable_reads(small_index):
    host = small_index(24)
    _, seqs = helpers.read_fastq()
    rng = np.random.RandomState(3)
    reads = [s[: rng.randint(0, 61)] for s in seqs[:600]]                       # 0..60 bases, many shorter than k
    reads += ["".join(rng.choice(list("ACGT"), rng.randint(24, 200))) for _ in range(300)]   # random: unmappable
    reads += ["A" * 80, "ACGT" * 30, "N" * 50, seqs[0][:30] + "NNNN" + seqs[0][34:], seqs[0].lower()]
    tiles, lens, wpr = pa.encode_reads_host(reads)
    r, (o_res, _, _, _) = check(host, tiles, lens, wpr)
    assert np.all(o_res["mapped"][np.asarray(lens) < 24] == 0)                  # L < K -> None (:82-84)
    assert o_res["mapped"][-1] == 1                                             # lower case maps like upper case


def test_colour_spill_and_node_traces(small_index):
    """col_cap = 1 forces the distinct-colour list through the HBM spill path; node traces equal map_read_to_nodes."""
    host = small_index(20)
    ids, seqs = helpers.read_fastq()
    tiles, lens, wpr = pa.encode_reads_host(seqs)
    r, _ = check(host, tiles, lens, wpr, 2, col_cap=1, nodes=True)
    assert r["steps"][3] > 0
    o = helpers.Oracle(host)
    a = host.arrays()
    for i in list(range(0, 400)) + [885, 1229, 153, 377]:
        rc, _, _, _, nodes = o.map_read(seqs[i])
        got = r["nodes"][i][: r["nodes_len"][i]].tolist()
        assert got == (nodes if rc else []), i


def test_long_reads_cross_many_nodes(small_index):
    host = small_index(31)
    _, seqs = helpers.read_fasta()
    reads = [s[:2000] for s in seqs if len(s) >= 300][:200]
    tiles, lens, wpr = pa.encode_reads_host(reads)
    check(host, tiles, lens, wpr, 2, col_cap=4)


@pytest.mark.parametrize("k", [31, 64])
def test_golden_synthetic_error_reads(small_index, k):
    lines = (helpers.GOLDEN / ("synth_err_k%d.tsv" % k)).read_text().splitlines()
    reads = [l.split("\t")[0] for l in lines]
    tiles, lens, wpr = pa.encode_reads_host(reads)
    o_res, o_coff, o_ids, _ = helpers.Oracle(small_index(k)).map_tiles(tiles, lens, wpr, 2, 2)
    assert [g.rstrip("\n") for g in helpers.result_lines(reads, o_res["mapped"], o_res["coverage"], o_res["mismatches"], o_coff, o_ids)] == lines
    r = helpers.Emu(small_index(k)).map_tiles(tiles, lens, wpr)
    res = r["results"]
    got = helpers.result_lines(reads, res["mismatches"] >> 31, res["coverage"], res["mismatches"] & 0x7FFFFFFF, r["coff"], r["ids"])
    assert [g.rstrip("\n") for g in got] == lines


@pytest.mark.parametrize("seed", [8, 10, 11, 12, 14, 27, 31])   # k = 21, 8, 32, 32, 16, 11, 21
def test_dense_dictionary_probe_paths(seed, tmp_path, monkeypatch):
    """the probe's rare paths — a key in another slot of its bucket than the first one named, in the next bucket, several buckets on —
    hardly occur in the table the library ships (load 0.25); a dictionary built at load 0.9 (the emulator's flattener honours
    PA_DICT_LOAD) makes them common: same results"""
    monkeypatch.setenv("PA_DICT_LOAD", "0.9")
    host, k, reads, clean, allowed = helpers.random_txome_case(seed, tmp_path)
    if host is None or k > 32:
        pytest.skip("no k <= 32 dictionary in this case")
    info = helpers.Emu(host).info()
    assert info["nbuckets"] * 4 < 2 * info["num_kmers"]            # denser than load 0.5
    tiles, lens, wpr = pa.encode_reads_host(reads)
    check(host, tiles, lens, wpr, allowed)


def test_device_dictionary_layout(small_index):
    info = helpers.Emu(small_index(24)).info()
    assert info["num_kmers"] == 1165762
    assert info["nbuckets"] * 4 >= 4 * info["num_kmers"]   # load factor <= 0.25 (device_layout.hpp, DICT_LOAD)


@pytest.mark.parametrize("k", [20, 31])
def test_chain_block_layout_is_well_formed(small_index, k):
    """chains merge nodes (fewer chains than nodes), tails add copies (more records than nodes), and every 128-byte block obeys the
    slot grammar of device_layout.hpp (checked block by block inside the emulator library)"""
    host = small_index(k)
    info = helpers.Emu(host).info()
    n = host.arrays()["num_nodes"]
    assert 0 < info["num_chains"] < n < info["num_segs"]
    assert info["blob_bytes"] % 128 == 64 and info["bad_blocks"] == 0   # (whole blocks + the 64-byte tail pad)
    assert info["branch_records"] > 1000                                # nodes with several right extensions carry a copy of the favoured one


def test_flatten_rejects_inconsistent_graphs(small_index):
    """a k-mer present in two nodes / a dangling extension must be refused at index creation."""
    host = small_index(20)
    a = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in host.arrays().items()}
    flat = host.flat()
    exts = a["node_exts"].copy()
    # set a right-extension bit that has no neighbour: pick a node and a base not in its exts
    n = int(np.flatnonzero((exts & 15) == 0)[0]) if np.any((exts & 15) == 0) else 0
    exts[n] |= 1 << int(np.flatnonzero([(exts[n] >> b) & 1 == 0 for b in range(4)])[0])
    flat.node_exts = exts.ctypes.data
    import ctypes as C
    h = C.c_void_p()
    assert helpers.emu_lib().emu_index_new(C.byref(flat), 2, C.byref(h)) != 0
    assert b"missing link" in helpers.emu_lib().emu_last_error()


def _index_of(tmp_path, txs, k, name="lt.fa"):
    fa = tmp_path / name
    fa.write_text("".join(">t%d|g%d\n%s\n" % (i, i, s) for i, s in enumerate(txs)))
    return pa.HostIndex.build_fasta(str(fa), k, 3)


def test_left_extension_into_a_node_that_starts_on_a_block_seam(tmp_path):
    """T2 = T1[64:] cuts T1's unitig by colour exactly at chain position 64, i.e. at window position 0 of chain
    block 1, where the node before it still has its record in the same block. A left extension that walks more than 192 bases
    back has to take the in-chain hop there (has_ext(Left), src/pseudoaligner.rs:183-199), not look for a left EDGE of the chain."""
    rng = np.random.RandomState(5)
    T1 = "".join(rng.choice(list("ACGT"), 700))
    host = _index_of(tmp_path, [T1, T1[64:]], 24)
    read = list(T1[29:329])                       # 300 bases, chain positions 29..328
    for j in range(0, 231, 23):                   # errors every 23 bases: no 24-mer hits before kmer_pos 231 (chain position 260)
        read[j] = "ACGT"[("ACGT".index(read[j]) + 1) % 4]
    reads = ["".join(read)] * 3
    tiles, lens, wpr = pa.encode_reads_host(reads)
    r, (o_res, o_coff, o_ids, ctr) = check(host, tiles, lens, wpr, allowed=12)
    assert ctr["left_extensions"] == 3 and int(o_res["coverage"][0]) == 300
    assert o_ids[o_coff[0]:o_coff[1]].tolist() == [0]   # the read starts before T2 does


@pytest.mark.parametrize("seed", range(12))
def test_long_chain_left_extensions(tmp_path, seed):
    host, reads, allowed = helpers.long_chain_case(seed, tmp_path)
    tiles, lens, wpr = pa.encode_reads_host(reads)
    r, (o_res, _, _, ctr) = check(host, tiles, lens, wpr, allowed)
    assert ctr["left_extensions"] > 50


def test_repeat_families_pending_classes_and_bitmaps():
    """a slice of bench.py's "config3r" transcriptome (repeat families in the last exons, low-complexity tracts): reads that cross a
    repeat element meet classes WITHOUT windows — tens to hundreds of ids spread over unrelated genes — which window mode keeps pending and
    applies after the walk (mask_pending). The kernel answers "which ids of the window are in the class" from the class's membership
    bitmap (device_layout.hpp, class_bitmap); the emulator computes the mask from the id list AND from the flattener's bitmap and
    poisons the result when they differ — so this test is the bitmaps' parity check on the CPU tier"""
    tx = pa.Txome.synthesize_repeats(1500, 5200, 7, gene_fraction_ppm=400000, young_div_lo_ppm=10000, young_div_hi_ppm=30000)
    host = pa.HostIndex.from_txome(tx, 24, 4)
    a = host.arrays()
    clen = (a["ec_offset"][1:] - a["ec_offset"][:-1]).astype(np.int64)
    assert int(clen.max()) > 60
    emu = helpers.Emu(host)
    info = emu.info()
    assert info["bitmap_min"] == 16 and info["num_bitmaps"] > 50 and info["bad_blocks"] == 0
    o = helpers.Oracle(host)
    for ppm, allowed in ((0, 2), (10000, 2), (30000, 1)):
        tiles, lens = tx.simulate_host(150, 9, 40000, ppm)
        wpr = pa.lib().pa_words_per_read(150)
        want = o.map_tiles(tiles, lens, wpr, allowed, 4)
        r = emu.map_tiles(tiles, lens, wpr, allowed, 8)
        helpers.assert_same_as_oracle(r["results"], r["coff"], r["ids"], want[0], want[1], want[2], "repeat families ppm=%d" % ppm)
        assert r["steps"][4] > 500                     # reads with pending classes were met (mask_pending ran)
        assert want[3]["class_sizes"] / want[3]["reads"] > 15


@pytest.mark.parametrize("seed", range(10))
def test_branch_points_on_block_seams(tmp_path, seed):
    """helpers.branch_case: branch records, favoured-branch tails and bubbles on / beside multiples of 64, clustered and dense-head errors,
    reads to 700 bases, allowed to 12 — the lane steps on the host vs the oracle"""
    host, reads, allowed = helpers.branch_case(seed, tmp_path)
    if host is None:
        pytest.skip("no k-mer")
    tiles, lens, wpr = pa.encode_reads_host(reads)
    r, _ = check(host, tiles, lens, wpr, allowed)
    assert helpers.Emu(host).info()["bad_blocks"] == 0


@pytest.mark.parametrize("seed", range(10))
def test_tandem_repeats_and_reads_of_exactly_k(tmp_path, seed):
    """helpers.tandem_case: k-mer cycles and self-loops over 1..4-letter alphabets, reads of exactly K, K + 1, K + 2 ... 1 200 bases"""
    host, reads, allowed = helpers.tandem_case(seed, tmp_path)
    if host is None:
        pytest.skip("no k-mer")
    tiles, lens, wpr = pa.encode_reads_host(reads)
    check(host, tiles, lens, wpr, allowed)


def test_transcripts_of_a_hundred_kilobases_map_onto_themselves(tmp_path):
    """validate_dbg (src/build_index.rs:300-367: every transcript mapped onto its own graph has coverage == its length and a class that
    contains it) for transcripts of 120 kb and 86 kb — reads far beyond the 2^14 bases the narrow lane state can count: the WIDE packing of
    lane_steps.hpp (28-bit positions, what the kernel's GREAD instantiations run), on the host, against the oracle; with substitutions too"""
    host, seqs = helpers.long_transcript_case(tmp_path)
    assert max(len(s) for s in seqs) == 120000
    tiles, lens, wpr = pa.encode_reads_host(seqs)
    r, (o_res, o_coff, o_ids, ctr) = check(host, tiles, lens, wpr, 2)
    for t, s in enumerate(seqs):
        if len(s) >= host.k:
            assert o_res["mapped"][t] and o_res["coverage"][t] == len(s) and t in o_ids[int(o_coff[t]):int(o_coff[t + 1])].tolist(), t
    rng = np.random.RandomState(3)
    noisy = []
    for s in seqs[-2:] + seqs[:6]:
        r_ = list(s)
        for j in rng.randint(0, len(r_), max(1, len(r_) // 400)):
            r_[j] = "ACGT"[("ACGT".index(r_[j]) + 1 + rng.randint(3)) % 4]
        noisy.append("".join(r_))
    tiles, lens, wpr = pa.encode_reads_host(noisy)
    _, (_, _, _, ctr) = check(host, tiles, lens, wpr, 2)
    assert ctr["reseeks"] > 50


@pytest.mark.parametrize("k,read_len,ppm,allowed", [(24, 150, 10000, 2), (31, 150, 30000, 0), (64, 100, 5000, 2), (20, 75, 50000, 2)])
def test_wide_lane_packing_on_short_reads(small_index, monkeypatch, k, read_len, ppm, allowed):
    """the WIDE packing of the lane state (lane_steps.hpp: what reads of more than 512 bases take on the GPU) forced onto short reads
    (PA_EMU_WIDE=1): the same text with 28-bit positions and counters in words of their own gives the same results as the oracle — and as the
    narrow packing, step for step"""
    host = small_index(k) if k in (20, 24, 31) else pa.build_index(str(helpers.FASTA), k, 8)
    tx = pa.Txome.from_host_index(host)
    tiles, lens = tx.simulate_host(read_len, 4, 20000, ppm)
    wpr = pa.lib().pa_words_per_read(read_len)
    narrow, _ = check(host, tiles, lens, wpr, allowed)
    monkeypatch.setenv("PA_EMU_WIDE", "1")
    wide, _ = check(host, tiles, lens, wpr, allowed)
    assert np.array_equal(narrow["steps"][1:3], wide["steps"][1:3])          # the same number of forward / left steps (probe steps depend on which keys the
                                                                             # parallel dictionary build happened to give their home slots)
    _, seqs = helpers.read_fastq()
    t2, l2, w2 = pa.encode_reads_host(seqs[:3000])
    check(host, t2, l2, w2, 2)
