"""`-m gpu` tier: the HIP path (through the C ABI) against the oracle on the same inputs — bit exact — plus the
committed golden fixtures and size-independent properties at BASELINE.json's batch sizes."""
import ctypes as C
import os

import numpy as np
import pytest

import helpers

pa = helpers.pa
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def aligners(small_index):
    cache = {}

    def get(k):
        if k not in cache:
            if pa.lib().pa_device_count() < 1:
                raise RuntimeError("the gpu tier needs a GPU and the HIP library: %s" % pa.lib().pa_last_error().decode())
            cache[k] = pa.Pseudoaligner(small_index(k), 0)
        return cache[k]
    return get


def gpu_vs_oracle(aligner, reads, allowed=2, what=""):
    res, coff, cids = aligner.map_batch(reads, allowed)
    o_res, o_coff, o_ids, ctr = helpers.Oracle(aligner.host).map_reads(reads, allowed, 8)
    helpers.assert_same_as_oracle(res, coff, cids, o_res, o_coff, o_ids, what)
    return res, coff, cids, ctr


@pytest.mark.parametrize("k", [20, 24, 31])
def test_small_fq_vs_oracle_and_fixture(aligners, k):
    ids, seqs = helpers.read_fastq()
    res, coff, cids, _ = gpu_vs_oracle(aligners(k), seqs, 2, "small.fq K=%d" % k)
    lines = helpers.result_lines(ids, res["mismatches"] >> 31, res["coverage"], res["mismatches"] & 0x7FFFFFFF, coff, cids)
    assert "".join(lines) == (helpers.GOLDEN / ("small_fq_k%d.tsv" % k)).read_text()


def test_tuning_knobs_in_the_environment_change_nothing(aligners, monkeypatch):
    """PA_MAP_ABLATE=3 (no result stores, no counts) / PA_POOL_SLOTS / PA_MAP_STATS are read by -DPA_DEBUG_KNOBS builds only: the
    shipped library ignores them and stays bit-exact, fused count table included"""
    import torch
    for k, v in (("PA_MAP_ABLATE", "3"), ("PA_POOL_SLOTS", "64"), ("PA_MAP_STATS", "1"), ("PA_MAP_BLOCKS_PER_CU", "1")):
        monkeypatch.setenv(k, v)
    a = aligners(24)
    _, seqs = helpers.read_fastq()
    res, coff, cids, _ = gpu_vs_oracle(a, seqs, 2, "small.fq K=24 with knobs exported")
    tiles, lens, wpr = pa.encode_reads_host(seqs)
    dev = torch.device("cuda", 0)
    n = len(seqs)
    d_tiles = torch.from_numpy(tiles.view(np.int64)).to(dev)
    d_lens = torch.from_numpy(lens.view(np.int32)).to(dev)
    cap = a.arena_hint(n)
    d_res = torch.zeros(n * 4, dtype=torch.int32, device=dev)
    d_arena = torch.zeros(cap, dtype=torch.int32, device=dev)
    d_counts = torch.zeros(a.counts_len(), dtype=torch.int64, device=dev)
    a.map_count_batch_device(d_tiles.data_ptr(), d_lens.data_ptr(), n, wpr, d_res.data_ptr(), d_arena.data_ptr(), cap, d_counts.data_ptr(), 2)
    a.map_finish()
    want = helpers.counts_reference(*helpers.Oracle(a.host).map_reads(seqs, 2, 8)[:3], a.host)
    assert np.array_equal(d_counts.cpu().numpy().astype(np.uint64), want.astype(np.uint64))


def test_thousands_of_classes_per_read(tmp_path):
    """16 383-base reads over ~5 400 unitigs with pairwise different classes: list mode far into the spill rows, the one-list-
    per-lane scan 64 lists at a time, more classes than the 12-bit counter of round 2 could count"""
    host, reads = helpers.thousands_of_classes_case(tmp_path)
    a = pa.Pseudoaligner(host, 0)
    res, coff, cids, ctr = gpu_vs_oracle(a, reads, 2, "suffix transcripts")
    assert ctr["node_visits"] > 8000 and cids[coff[0]:coff[1]].tolist() == [0]


def test_reference_literals_through_map_read(aligners):
    a = aligners(20)
    ex1 = "GGCTGTCAACCAGTCCATAGGCAGGGCCATCAGGCACCAAAGGGATTCTGCCAGCATAGT"          # src/build_index.rs:429-434
    snp = "GGCTGTCAACCAGTCCATAGGCGGGGCCATCAGGCACCAAAGGGATTCTGCCAGCATAGT"          # :436-441
    assert a.map_read(ex1) == ([1, 30], len(ex1))
    assert a.map_read(snp) == ([1, 30], len(snp))
    assert a.map_read_with_mismatch(snp, 2) == ([1, 30], 60, 1)
    assert a.map_read("ACGT") is None                                           # shorter than k (:82-84)
    assert a.map_read("ACGTTGCA" * 10) is None                                  # no k-mer in the graph
    nodes, cov = a.map_read_to_nodes(ex1)
    rc, _, ocov, _, onodes = helpers.Oracle(a.host).map_read(ex1)
    assert (nodes, cov) == (onodes, ocov)


@pytest.mark.parametrize("k,read_len,ppm,allowed,n", [(24, 100, 0, 2, 300000), (24, 150, 10000, 2, 300000), (31, 150, 10000, 2, 300000),
                                                      (20, 75, 50000, 2, 100000), (31, 150, 30000, 0, 100000), (24, 150, 30000, 1, 100000),
                                                      (24, 150, 60000, 3, 100000), (64, 150, 5000, 2, 100000), (40, 100, 10000, 2, 100000)])
def test_simulated_reads_device_batches(aligners, k, read_len, ppm, allowed, n):
    """device-resident API: reads simulated on the GPU, mapped, compared with the oracle on the host-simulated twins"""
    import torch
    a = aligners(k)
    tx = pa.Txome.from_host_index(a.host)
    wpr = pa.lib().pa_words_per_read(read_len)
    dev = torch.device("cuda", 0)
    d_tiles = torch.zeros(pa.lib().pa_tiles_words(n, wpr), dtype=torch.int64, device=dev)
    d_lens = torch.zeros(n, dtype=torch.int32, device=dev)
    tx.simulate_device(read_len, 4, n, d_tiles.data_ptr(), d_lens.data_ptr(), ppm, 0, wpr)
    h_tiles, h_lens = tx.simulate_host(read_len, 4, n, ppm, 0, wpr)
    torch.cuda.synchronize()
    assert np.array_equal(d_tiles.cpu().numpy().view(np.uint64), h_tiles) and np.array_equal(d_lens.cpu().numpy().view(np.uint32), h_lens)
    cap = a.arena_hint(n)
    d_res = torch.zeros(n * 4, dtype=torch.int32, device=dev)
    d_arena = torch.zeros(cap, dtype=torch.int32, device=dev)
    d_col = torch.zeros(n, dtype=torch.int32, device=dev)
    a.map_batch_device(d_tiles.data_ptr(), d_lens.data_ptr(), n, wpr, d_res.data_ptr(), d_arena.data_ptr(), cap, allowed, d_col.data_ptr())
    used, _ = a.map_finish()
    res = d_res.cpu().numpy().view(pa.RESULT_DTYPE)
    coff, cids = pa.gather_classes(res, d_arena[: max(used, 1)].cpu().numpy().view(np.uint32), a.host)
    o_res, o_coff, o_ids, ctr = helpers.Oracle(a.host).map_tiles(h_tiles, h_lens, wpr, allowed, 8)
    helpers.assert_same_as_oracle(res, coff, cids, o_res, o_coff, o_ids, "simulated k=%d" % k)
    # the count kernel against its numpy definition
    d_counts = torch.zeros(a.counts_len(), dtype=torch.int64, device=dev)
    a.counts_accumulate_device(d_res.data_ptr(), d_arena.data_ptr(), d_col.data_ptr(), n, d_counts.data_ptr())
    torch.cuda.synchronize()
    sub = slice(0, 20000)
    want_sub = helpers.counts_reference(o_res[sub], o_coff[: 20001], o_ids, a.host)
    d_counts2 = torch.zeros(a.counts_len(), dtype=torch.int64, device=dev)
    a.counts_accumulate_device(d_res.data_ptr(), d_arena.data_ptr(), d_col.data_ptr(), 20000, d_counts2.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(d_counts2.cpu().numpy(), want_sub)
    assert int(d_counts.sum().item()) == n
    # fused variant: map + count in one launch
    d_counts4 = torch.zeros(a.counts_len(), dtype=torch.int64, device=dev)
    a.map_count_batch_device(d_tiles.data_ptr(), d_lens.data_ptr(), 20000, wpr, d_res.data_ptr(), d_arena.data_ptr(), cap, d_counts4.data_ptr(), allowed)
    a.map_finish()
    assert np.array_equal(d_counts4.cpu().numpy(), want_sub)
    a.map_count_batch_device(d_tiles.data_ptr(), d_lens.data_ptr(), n, wpr, d_res.data_ptr(), d_arena.data_ptr(), cap, d_counts4.data_ptr(), allowed)
    a.map_finish()
    assert np.array_equal((d_counts4 - d_counts).cpu().numpy(), want_sub)
    # without the colour hint every class goes through the content lookup: same table
    d_counts3 = torch.zeros(a.counts_len(), dtype=torch.int64, device=dev)
    a.counts_accumulate_device(d_res.data_ptr(), d_arena.data_ptr(), 0, 20000, d_counts3.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(d_counts3.cpu().numpy(), want_sub)


@pytest.mark.parametrize("seed", range(6))
def test_random_transcriptomes_list_mode(tmp_path, seed):
    """the same fuzz on hundreds of transcripts over few shared segments: classes of many ids spread over more than two
    windows (list mode, many classes per read, bases of more than 8 ids)"""
    host, k, reads, clean, allowed = helpers.random_txome_case(seed, tmp_path, big=True)
    if host is None:
        pytest.skip("every transcript is shorter than k")
    res, coff, cids = pa.Pseudoaligner(host).map_batch(reads, allowed)
    o_res, o_coff, o_ids, _ = helpers.Oracle(host).map_reads(clean, allowed, 4)
    helpers.assert_same_as_oracle(res, coff, cids, o_res, o_coff, o_ids, "big random txome seed %d k=%d" % (seed, k))


@pytest.mark.parametrize("seed,ordered", [(0, False), (1, False), (0, True), (1, True)])
def test_many_classes_per_read(tmp_path, seed, ordered):
    """list mode with tens to hundreds of DIFFERENT classes per read (chains of short shared segments, K = 11, reads of up to
    1500 bases): the SCAN step's packed passes (<= 64 classes) and its whole-wave passes (more), class rows far into the
    spill area; ordered: every class has ~90 ids, so the base of the intersection has more than 8 (the cooperative step)"""
    host, reads = helpers.many_classes_case(seed, tmp_path, ordered=ordered)
    a = host.arrays()
    orc = helpers.Oracle(host)
    ncls = [len({int(a["node_colour"][n]) for n in orc.map_read(r, 2)[4]}) for r in reads[:40]]
    assert max(ncls) > 64 and min(ncls) < 64                                     # both kinds of pass are exercised
    res, coff, cids = pa.Pseudoaligner(host).map_batch(reads, 2)
    o_res, o_coff, o_ids, _ = orc.map_reads(reads, 2, 4)
    helpers.assert_same_as_oracle(res, coff, cids, o_res, o_coff, o_ids, "many classes per read, seed %d ordered %d" % (seed, ordered))


def test_ragged_empty_short_and_odd_reads(aligners):
    a = aligners(24)
    _, seqs = helpers.read_fastq()
    rng = np.random.RandomState(3)
    reads = [s[: rng.randint(0, 61)] for s in seqs[:600]]
    reads += ["".join(rng.choice(list("ACGT"), rng.randint(24, 200))) for _ in range(300)]
    reads += ["A" * 80, "ACGT" * 30, "N" * 50, seqs[0][:30] + "NNNN" + seqs[0][34:], seqs[0].lower(), ""]
    res, coff, cids, _ = gpu_vs_oracle(a, reads, 2, "ragged")
    lens = np.array([len(r) for r in reads])
    assert np.all((res["mismatches"] >> 31)[lens < 24] == 0)
    r0, c0, i0 = a.map_batch([])                                                 # empty batch
    assert len(r0) == 0 and c0.tolist() == [0] and len(i0) == 0
    one = a.map_batch([seqs[0]])                                                 # batch of one
    assert one[0]["coverage"][0] == 60


def test_long_reads_and_max_length(aligners):
    a = aligners(31)
    _, seqs = helpers.read_fasta()
    reads = [s[:2048] for s in seqs if len(s) >= 300][:300] + [s[:600] for s in seqs if len(s) >= 600][:300]
    gpu_vs_oracle(a, reads, 2, "long reads")
    # reads beyond 512 bases stay in their HBM tile while they are mapped (the other kernel variant): with errors, ragged
    rng = np.random.RandomState(11)
    longer = []
    for s in [s for s in seqs if len(s) >= 1200][:120]:
        r = list(s[: rng.randint(513, min(len(s), 9000) + 1)])
        for j in rng.randint(0, len(r), len(r) // 150):
            r[j] = "ACGT"[rng.randint(4)]
        longer.append("".join(r))
    gpu_vs_oracle(a, longer + [seqs[0][:40], "", seqs[1][:700]], 2, "reads kept in HBM")
    top = max(seqs, key=len)
    assert len(top) > 16000
    gpu_vs_oracle(a, [(top * 2)[:16383]], 2, "16 383 bases (the narrow lane state's limit; such reads take the wide one)")
    gpu_vs_oracle(a, [(top * 2)[:16384], (top * 3)[:40000]], 2, "beyond 2^14 bases")
    with pytest.raises(pa.PaError):
        a.map_batch(["A" * (pa._ffi.PA_MAX_READ_LEN + 1)])                       # beyond PA_MAX_READ_LEN


def test_self_mapping_of_transcripts(aligners):
    """validate_dbg part 2 (src/build_index.rs:300-367) through the GPU for EVERY transcript (up to 16 355 bases: reads that
    long stay in HBM while they are mapped), incl. the node traces of the longest ones"""
    a = aligners(20)
    _, seqs = helpers.read_fasta()
    idx = [i for i, s in enumerate(seqs) if len(s) >= 20]
    assert max(len(seqs[i]) for i in idx) > 16000 and len(idx) == sum(len(s) >= 20 for s in seqs)
    res, coff, cids = a.map_batch([seqs[i] for i in idx])
    for j, i in enumerate(idx):
        assert res["mismatches"][j] >> 31 == 1 and res["coverage"][j] == len(seqs[i])
        cls = cids[int(coff[j]):int(coff[j + 1])].tolist()
        assert (i in cls) if len(cls) > 1 else cls == [i]
    o = helpers.Oracle(a.host)
    longest = sorted(idx, key=lambda i: -len(seqs[i]))[:3]
    r, nodes, nlen = a.map_batch_nodes([seqs[i] for i in longest])
    for j, i in enumerate(longest):
        rc, _, cov, _, onodes = o.map_read(seqs[i])
        assert rc == 1 and nodes[j][: nlen[j]].tolist() == onodes and r["coverage"][j] == cov


def test_node_traces_match_map_read_to_nodes(aligners):
    a = aligners(20)
    _, seqs = helpers.read_fastq()
    sel = seqs[:1500]
    res, nodes, nlen = a.map_batch_nodes(sel)
    o = helpers.Oracle(a.host)
    for i, s in enumerate(sel):
        rc, _, cov, _, onodes = o.map_read(s)
        assert nodes[i][: nlen[i]].tolist() == (onodes if rc else []) and res["coverage"][i] == cov


def test_arena_overflow_is_reported_not_silent(aligners):
    import torch
    a = aligners(24)
    tx = pa.Txome.from_host_index(a.host)
    n, wpr = 50000, 4
    h_tiles, h_lens = tx.simulate_host(100, 9, n)
    dev = torch.device("cuda", 0)
    d_tiles = torch.from_numpy(h_tiles.view(np.int64)).to(dev)
    d_lens = torch.from_numpy(h_lens.view(np.int32)).to(dev)
    d_res = torch.zeros(n * 4, dtype=torch.int32, device=dev)
    d_arena = torch.zeros(1000, dtype=torch.int32, device=dev)
    a.map_batch_device(d_tiles.data_ptr(), d_lens.data_ptr(), n, wpr, d_res.data_ptr(), d_arena.data_ptr(), 1000)
    with pytest.raises(pa.PaError) as e:
        a.map_finish()
    assert e.value.code == pa._ffi.PA_ERR_ARENA_FULL


def test_encode_kernel_equals_host_packing(aligners):
    import torch
    a = aligners(24)
    _, seqs = helpers.read_fastq()
    reads = seqs[:1000] + ["acgtnNRY" * 9, "", "T"]
    data, offsets = pa.concat_reads(reads)
    h_tiles, h_lens, wpr = pa.encode_reads_host((data, offsets))
    dev = torch.device("cuda", 0)
    d_ascii = torch.from_numpy(data.copy()).to(dev)
    d_off = torch.from_numpy(offsets.view(np.int64)).to(dev)
    d_tiles = torch.zeros(len(h_tiles), dtype=torch.int64, device=dev)
    d_lens = torch.zeros(len(reads), dtype=torch.int32, device=dev)
    a.encode_reads_device(d_ascii.data_ptr(), d_off.data_ptr(), len(reads), wpr, d_tiles.data_ptr(), d_lens.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(d_tiles.cpu().numpy().view(np.uint64), h_tiles) and np.array_equal(d_lens.cpu().numpy().view(np.uint32), h_lens)


def test_process_reads_output_format(aligners, tmp_path):
    """process_reads (src/pseudoaligner.rs:420-514): Rust Debug tuples, flag rule of :455, input order"""
    a = aligners(20)
    out = tmp_path / "out.txt"
    n, flagged = pa.process_reads(str(helpers.FASTQ), a, str(out), 3)
    ids, seqs = helpers.read_fastq()
    res, coff, cids, _ = helpers.Oracle(a.host).map_reads(seqs, 2, 4)
    want = []
    for i, rid in enumerate(ids):
        cl = cids[int(coff[i]):int(coff[i + 1])].tolist()
        flag = bool(res["mapped"][i]) and res["coverage"][i] >= 32 and not cl
        want.append('(%s, "%s", [%s], %d)' % ("true" if flag else "false", rid, ", ".join(map(str, cl)), res["coverage"][i] if res["mapped"][i] else 0))
    got = out.read_text().splitlines()
    assert n == len(ids) == len(got) and got == want
    assert got[0] == '(false, "gencode_small_line15", [0, 1, 30], 60)'
    assert flagged == sum(1 for w in want if w.startswith("(true")) == 12        # 12 flagged at K=20
    with pytest.raises(pa.PaError):
        pa.process_reads(str(tmp_path / "missing.fq"), a, str(out))
    bad = tmp_path / "bad.fq"
    bad.write_text("@r1\nACGT\n+\nIIII\nnot a record\n")
    with pytest.raises(pa.PaError):
        pa.process_reads(str(bad), a, str(out))


def _expected_lines(a, ids, seqs):
    res, coff, cids, _ = helpers.Oracle(a.host).map_reads(seqs, 2, 4)
    want = []
    for i, rid in enumerate(ids):
        cl = cids[int(coff[i]):int(coff[i + 1])].tolist()
        flag = bool(res["mapped"][i]) and res["coverage"][i] >= 32 and not cl
        want.append('(%s, "%s", [%s], %d)' % ("true" if flag else "false", rid, ", ".join(map(str, cl)), res["coverage"][i] if res["mapped"][i] else 0))
    return want


def _pack_words(seqs, msb_first=False):
    """reads as a DnaString holds them: 2-bit words, every read on a word boundary (checker-side packer: helpers.pack_read)"""
    words, offs, lens = [], [0], []
    for s in seqs:
        w = helpers.pack_read(s)[: (len(s) + 31) // 32].copy()
        if msb_first:   # base j in bits 62 - 2 (j % 32): the 32 two-bit fields of every word in the opposite order
            out = np.zeros_like(w)
            for j in range(32):
                out |= ((w >> np.uint64(2 * j)) & np.uint64(3)) << np.uint64(62 - 2 * j)
            w = out
        words.append(w)
        offs.append(offs[-1] + len(w))
        lens.append(len(s))
    return (np.concatenate(words) if words else np.zeros(0, np.uint64)), np.array(offs, np.uint64), np.array(lens, np.uint32)


@pytest.mark.parametrize("k", [20, 64])
def test_packed_reads_map_like_their_ascii(aligners, k):
    """pa_map_batch_packed / pa_map_read_packed (map_read(&DnaString), src/pseudoaligner.rs:381): reads handed over as 2-bit
    words, in both word orders, against the oracle; ragged lengths, empty and shorter-than-k reads, garbage beyond a read's length"""
    a = aligners(k) if k == 20 else pa.Pseudoaligner(pa.build_index(str(helpers.FASTA), k, 8), 0)
    _, seqs = helpers.read_fastq()
    rng = np.random.default_rng(11)
    reads = [s[: int(rng.integers(0, 61))] for s in seqs[:500]] + seqs[500:3000] + ["", "ACGT", seqs[0] * 3]
    o_res, o_coff, o_ids, _ = helpers.Oracle(a.host).map_reads(reads, 2, 4)
    for msb in (False, True):
        words, offs, lens = _pack_words(reads, msb)
        junk = words.copy()
        for i, L in enumerate(lens):   # bits beyond the last base must not matter
            if L % 32 and L:
                last = int(offs[i]) + (int(L) - 1) // 32
                keep = np.uint64((1 << (2 * (int(L) % 32))) - 1)
                junk[last] |= ~(keep if not msb else ~np.uint64((1 << (64 - 2 * (int(L) % 32))) - 1))
        res, coff, cids = a.map_batch_packed(junk, offs, lens, 1 if msb else 0, 2)
        helpers.assert_same_as_oracle(res, coff, cids, o_res, o_coff, o_ids, "packed reads, msb_first=%s" % msb)
    words, offs, lens = _pack_words(reads)
    for i in (0, 600, 2999, len(reads) - 1, len(reads) - 3):
        got = a.map_read_packed(words[int(offs[i]):int(offs[i + 1])], int(lens[i]))
        want = (o_ids[int(o_coff[i]):int(o_coff[i + 1])].tolist(), int(o_res["coverage"][i]), int(o_res["mismatches"][i])) if o_res["mapped"][i] else None
        assert got == want, i
    with pytest.raises(pa.PaError):
        a.map_batch_packed(words, offs, lens, 7)                       # unknown layout
    with pytest.raises(pa.PaError):
        a.map_batch_packed(words[:1], np.array([0, 1], np.uint64), np.array([40], np.uint32))   # 40 bases do not fit one word


def test_record_stream_is_process_reads_for_a_caller_that_holds_the_reader(aligners):
    """pa_record_stream_*: records pushed in uneven chunks through small batches (seams, words per read changing between batches),
    tuples pulled in push order == the oracle's tuples; nothing is rendered before a batch is full or flushed"""
    a = aligners(24)
    ids, seqs = helpers.read_fastq()
    rng = np.random.default_rng(9)
    ids, seqs = list(ids[:5000]), list(seqs[:5000])
    for i in range(0, 5000, 7):
        seqs[i] = seqs[i][: int(rng.integers(0, 61))]
    seqs[100] = seqs[101] * 4                                            # one long read: its batch needs more words per read
    weird = 'we"ird\\id\ttab'                                            # a quote, a backslash, a tab: Rust's Debug escapes them
    ids[5] = weird
    seqs[6] = seqs[6].lower().replace("a", "n", 1)
    want = [w.replace(weird, 'we\\"ird\\\\id\\ttab') for w in _expected_lines(a, ids, seqs)]
    for batch, threads in ((640, 3), (64, 1), (0, 0)):
        rs = pa.RecordStream(a, threads, batch)
        got = b""
        i = 0
        while i < len(ids):
            n = int(rng.integers(1, 900))
            rs.push(ids[i:i + n], seqs[i:i + n])
            got += rs.drain(1 << 12)                                     # small pulls: whole lines only
            i += n
        if batch == 0:
            assert got == b""                                            # 5000 records never fill the default batch: nothing yet
        rs.flush()
        if batch == 0:
            with pytest.raises(pa.PaError) as err:
                rs.pull(8, grow=False)                                   # smaller than one tuple: refused, nothing lost
            assert err.value.code == pa._ffi.PA_ERR_BUFFER_TOO_SMALL     # ... with a status of its own and the size the tuple needs
            n = C.c_size_t()
            buf = C.create_string_buffer(8)
            assert pa.lib().pa_records_pull(rs._h, buf, 8, C.byref(n)) == pa._ffi.PA_ERR_BUFFER_TOO_SMALL and n.value == len(want[0]) + 1
            assert rs.pull(8) == (want[0] + "\n").encode()                    # ... and the Python wrapper grows its buffer to that and pulls again
            want_rest = want[1:]
        got += rs.drain()
        assert rs.stats() == (len(ids), sum(1 for w in want if w.startswith("(true")))
        assert got.decode().splitlines() == (want_rest if batch == 0 else want), (batch, threads)
        rs.flush()                                                       # nothing pending: a no-op
        assert rs.drain() == b""
        rs.close()


def test_record_streams_and_process_reads_share_their_parked_buffers(aligners, tmp_path):
    """the pinned / device buffers of the two batches in flight are parked on the index between calls and taken over by whichever
    form of process_reads runs next: file -> stream -> two streams at once -> file, every output equal to the oracle's tuples"""
    a = aligners(24)
    ids, seqs = helpers.read_fastq()
    ids, seqs = list(ids[:4000]), list(seqs[:4000])
    want = _expected_lines(a, ids, seqs)
    fq = tmp_path / "in.fq"
    fq.write_text("".join("@%s\n%s\n+\n%s\n" % (i, s, "I" * len(s)) for i, s in zip(ids, seqs)))

    def by_file():
        out = tmp_path / "out.txt"
        assert pa.process_reads(str(fq), a, str(out), 3)[0] == len(ids)
        return out.read_text().splitlines()

    def by_stream(rs, lo, hi):
        for i in range(lo, hi, 700):
            rs.push(ids[i:min(i + 700, hi)], seqs[i:min(i + 700, hi)])
        rs.flush()
        return rs.drain().decode().splitlines()

    assert by_file() == want
    rs = pa.RecordStream(a, 2, 1024)
    assert by_stream(rs, 0, 4000) == want
    rs.close()
    r1, r2 = pa.RecordStream(a, 2, 512), pa.RecordStream(a, 1, 2048)          # the second finds nothing parked and makes its own
    got2 = by_stream(r2, 1000, 4000)
    got1 = by_stream(r1, 0, 1000)
    assert got1 + got2 == want
    r1.close(); r2.close()
    assert by_file() == want


def test_process_reads_pipeline_seams(aligners, tmp_path, monkeypatch):
    """the ingest pipeline of process_reads: many small batches, every thread count, CRLF, no final newline, trailing
    blank lines, ragged read lengths (words per read change between batches), lower case and N, empty input"""
    a = aligners(24)
    ids, seqs = helpers.read_fastq()
    rng = np.random.default_rng(5)
    ids, seqs = list(ids[:3000]), list(seqs[:3000])
    for i in range(0, 3000, 7):                       # ragged: truncate, extend with a second read, lower-case, N
        s = seqs[i]
        kind = i % 4
        seqs[i] = s[: int(rng.integers(1, len(s)))] if kind == 0 else (s + seqs[(i + 1) % 3000] + s)[: int(rng.integers(61, 181))] if kind == 1 else \
            s.lower() if kind == 2 else s[:20] + "N" + s[21:]
    ids[5], ids[6] = 'qu"ote', "back\\slash\x01ctl"          # Debug formatting of the id (:490): \" \\\\ \\u{1}
    want = _expected_lines(a, ids, [s.upper().replace("N", "A") for s in seqs])
    want[5] = want[5].replace('qu"ote', 'qu\\"ote')
    want[6] = want[6].replace("back\\slash\x01ctl", "back\\\\slash\\u{1}ctl")
    out = tmp_path / "o.txt"
    variants = {
        "plain": "".join("@%s extra words\n%s\n+\n%s\n" % (i, s, "I" * len(s)) for i, s in zip(ids, seqs)),
        "crlf": "".join("@%s\r\n%s\r\n+\r\n%s\r\n" % (i, s, "I" * len(s)) for i, s in zip(ids, seqs)),
    }
    # sequence and qualities wrapped over several lines (bio's reader accepts that; the scan falls back to a sequential rewrite)
    wrap = lambda t, w: "\n".join(t[j:j + w] for j in range(0, len(t), w)) if t else ""
    variants["wrapped"] = "".join("@%s extra\n%s\n+\n%s\n" % (i, wrap(s, 25), wrap("I" * len(s), 25)) for i, s in zip(ids, seqs))
    variants["wrapped_crlf_tail"] = variants["wrapped"].replace("\n", "\r\n") + "\r\n\r\n"
    variants["no_final_newline"] = variants["plain"][:-1]
    variants["trailing_blank_lines"] = variants["plain"] + "\n\n\n"
    for name, text in variants.items():
        fq = tmp_path / (name + ".fq")
        fq.write_text(text, newline="")
        for batch, threads in ((64, 1), (192, 3), (1024, 8), (1 << 22, 5)):
            monkeypatch.setenv("PA_INGEST_BATCH", str(batch))
            n, flagged = pa.process_reads(str(fq), a, str(out), threads)
            got = out.read_text().splitlines()
            assert n == len(ids) and got == want, (name, batch, threads)
            assert flagged == sum(1 for w in want if w.startswith("(true"))
    # a LAST record with an empty sequence: its empty quality line looks like a trailing blank line, or is missing altogether
    # (bio's reader reads nothing at the end of the file and hands the record out)
    monkeypatch.setenv("PA_INGEST_BATCH", "1024")
    for name, ending in (("empty_last", "@last\n\n+\n\n"), ("empty_last_crlf", "@last\r\n\r\n+\r\n\r\n"), ("empty_last_no_qual", "@last\n\n+"),
                         ("empty_last_blank_lines", "@last\n\n+\n\n\n\n")):
        fq = tmp_path / (name + ".fq")
        fq.write_text(variants["plain"] + ending, newline="")
        n, flagged = pa.process_reads(str(fq), a, str(out), 3)
        assert n == len(ids) + 1 and out.read_text().splitlines() == want + ['(false, "last", [], 0)'], name
    # gzip input (utils::open_with_gz, src/utils.rs:45-57), also as two concatenated members
    import gzip
    gzp = tmp_path / "plain.fq.gz"
    text = variants["plain"].encode()
    gzp.write_bytes(gzip.compress(text[: len(text) // 2]) + gzip.compress(text[len(text) // 2:]))
    monkeypatch.setenv("PA_INGEST_BATCH", "1024")
    n, flagged = pa.process_reads(str(gzp), a, str(out), 4)
    assert n == len(ids) and out.read_text().splitlines() == want
    broken = tmp_path / "broken.fq.gz"
    broken.write_bytes(gzp.read_bytes()[:2000])
    with pytest.raises(pa.PaError):
        pa.process_reads(str(broken), a, str(out), 2)
    empty = tmp_path / "empty.fq"
    empty.write_text("")
    assert pa.process_reads(str(empty), a, str(out), 4) == (0, 0) and out.read_text() == ""
    for name, text in (("noat", "r1\nACGT\n+\nIIII\n"), ("blank_inside", "@r1\nACGT\n+\nIIII\n\n@r2\nACGT\n+\nIIII\n"),
                       ("wrapped_trunc", "@r1\nACGT\nACGT\n+\nIIII\nIIII\n@r2\nAC\nGT\n")):
        bad = tmp_path / (name + ".fq")
        bad.write_text(text)
        with pytest.raises(pa.PaError):
            pa.process_reads(str(bad), a, str(out), 2)
    trunc = tmp_path / "trunc.fq"
    trunc.write_text("@r1\nACGT\n+\nIIII\n@r2\nACGT\n")
    with pytest.raises(pa.PaError):
        pa.process_reads(str(trunc), a, str(out), 2)


@pytest.mark.parametrize("window", [1, 300, 5000, 100000])
def test_process_reads_scan_windows(aligners, tmp_path, monkeypatch, window):
    """pa_process_reads scans a mapped file a window at a time (PA_INGEST_WINDOW bytes; a window ends behind its last whole record, the
    next one starts there): whatever the window — one record per window, a few, windows that end inside every kind of line — the
    output is the output of the one-window scan, for four-line text, CRLF, wrapped records (a window that is not in four-line shape
    hands the rest of the file to the rewriting scan), a missing final line break, trailing blank lines and an empty last record"""
    a = aligners(24)
    ids, seqs = helpers.read_fastq()
    ids, seqs = list(ids[:1500]), list(seqs[:1500])
    for i in range(0, 1500, 5):
        seqs[i] = seqs[i][: 1 + (7 * i) % len(seqs[i])]
    want = _expected_lines(a, ids, [s.upper().replace("N", "A") for s in seqs])
    wrap = lambda t, w: "\n".join(t[j:j + w] for j in range(0, len(t), w)) if t else ""
    plain = "".join("@%s extra words\n%s\n+\n%s\n" % (i, s, "I" * len(s)) for i, s in zip(ids, seqs))
    half = len(ids) // 2
    variants = {
        "plain": (plain, want),
        "crlf": (plain.replace("\n", "\r\n"), want),
        "no_final_newline": (plain[:-1], want),
        "trailing_blank_lines": (plain + "\n\n\n", want),
        "empty_last": (plain + "@last\n\n+\n\n\n", want + ['(false, "last", [], 0)']),
        # four-line records first, wrapped ones behind them: the windows before the first wrapped record are taken as they are
        "wrapped_second_half": ("".join("@%s x\n%s\n+\n%s\n" % (i, s if k < half else wrap(s, 25), "I" * len(s) if k < half else wrap("I" * len(s), 25))
                                        for k, (i, s) in enumerate(zip(ids, seqs))), want),
    }
    out = tmp_path / "o.txt"
    monkeypatch.setenv("PA_INGEST_WINDOW", str(window))
    for name, (text, lines) in variants.items():
        fq = tmp_path / (name + ".fq")
        fq.write_text(text, newline="")
        for batch, threads in ((64, 1), (1 << 22, 5)):
            monkeypatch.setenv("PA_INGEST_BATCH", str(batch))
            n, _ = pa.process_reads(str(fq), a, str(out), threads)
            assert n == len(lines) and out.read_text().splitlines() == lines, (name, window, batch, threads)
    bad = tmp_path / "bad.fq"
    bad.write_text(plain + "not a record\n")
    with pytest.raises(pa.PaError):
        pa.process_reads(str(bad), a, str(out), 2)


def test_process_reads_concurrent_calls_one_index(aligners, tmp_path, monkeypatch):
    """pa_process_reads parks its batch buffers on the index between calls (include/pseudoaligner_amd.h): calls that overlap
    on one index each get their own set, and a later call that needs bigger batches grows the parked one"""
    import threading
    a = aligners(24)
    ids, seqs = helpers.read_fastq()
    ids, seqs = list(ids[:4000]), list(seqs[:4000])
    want = _expected_lines(a, ids, seqs)
    fq = tmp_path / "in.fq"
    fq.write_text("".join("@%s\n%s\n+\n%s\n" % (i, s, "I" * len(s)) for i, s in zip(ids, seqs)))
    monkeypatch.setenv("PA_INGEST_BATCH", "256")
    outs = [tmp_path / ("o%d.txt" % t) for t in range(4)]
    got = [None] * 4
    def run(t):
        for _ in range(3):
            got[t] = pa.process_reads(str(fq), a, str(outs[t]), 2)
    th = [threading.Thread(target=run, args=(t,)) for t in range(4)]
    for t in th: t.start()
    for t in th: t.join()
    for t in range(4):
        assert got[t] is not None and got[t][0] == len(ids) and outs[t].read_text().splitlines() == want, t
    monkeypatch.setenv("PA_INGEST_BATCH", "2048")   # bigger batches than the parked buffers were sized for
    assert pa.process_reads(str(fq), a, str(outs[0]), 3)[0] == len(ids) and outs[0].read_text().splitlines() == want


def test_process_reads_fuzz(monkeypatch):
    """tools/gpu_fastq_fuzz.py, a short run: random FASTQ files (ids with quotes / control bytes / tabs, read lengths 0..300, IUPAC
    letters, LF / CRLF, wrapped records, gzip members, trailing blank lines) through pa_process_reads against the oracle's tuples"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gpu_fastq_fuzz", str(helpers.ROOT / "tools" / "gpu_fastq_fuzz.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    monkeypatch.setenv("PA_INGEST_BATCH", "64")   # (the fuzz sets it per file; restored afterwards)
    assert fuzz.run(36, 20000) == 0


# 12..308: seeds on which a lookup that only tried the first fingerprint match of a bucket missed k-mers (two keys of one
# bucket sharing their low 31 bits — low-complexity sequence); found by tools/gpu_soak.py
@pytest.mark.parametrize("seed", list(range(10)) + [12, 19, 31, 39, 48, 55, 96, 242, 278, 308])
def test_random_transcriptomes(tmp_path, seed):
    """differential fuzz (helpers.random_txome_case): GPU vs oracle, bit exact, incl. the fused count table"""
    import torch
    host, k, reads, clean, allowed = helpers.random_txome_case(seed, tmp_path)
    if host is None:
        pytest.skip("every transcript is shorter than k")
    a = pa.Pseudoaligner(host)
    res, coff, cids = a.map_batch(reads, allowed)
    o_res, o_coff, o_ids, _ = helpers.Oracle(host).map_reads(clean, allowed, 4)
    helpers.assert_same_as_oracle(res, coff, cids, o_res, o_coff, o_ids, "random txome seed %d k=%d" % (seed, k))
    # fused count table through the device API
    tiles, lens, wpr = pa.encode_reads_host(reads)
    dev = torch.device("cuda", 0)
    d_tiles, d_lens = torch.from_numpy(tiles.view(np.int64)).to(dev), torch.from_numpy(np.asarray(lens, np.uint32).view(np.int32)).to(dev)
    n = len(reads)
    cap = a.arena_hint(n)
    d_res = torch.zeros(n * 4, dtype=torch.int32, device=dev)
    d_arena = torch.zeros(cap, dtype=torch.int32, device=dev)
    d_counts = torch.zeros(a.counts_len(), dtype=torch.int64, device=dev)
    a.map_count_batch_device(d_tiles.data_ptr(), d_lens.data_ptr(), n, wpr, d_res.data_ptr(), d_arena.data_ptr(), cap, d_counts.data_ptr(), allowed)
    a.map_finish()
    assert np.array_equal(d_counts.cpu().numpy(), helpers.counts_reference(o_res, o_coff, o_ids, host))


def test_hot_classes_count_table(tmp_path):
    """a handful of classes take every read (a highly expressed gene): the counting sort of the key streams (count_sort.hip: LDS atomics on few addresses)
    must still add up to exactly the histogram of the per-read results, also across repeated launches into one table"""
    import torch
    _, seqs = helpers.read_fasta()
    fa = tmp_path / "hot.fa"
    fa.write_text("".join(">t%d\n%s\n" % (i, s) for i, s in enumerate([s for s in seqs if len(s) >= 300][:6])))
    host = pa.HostIndex.build_fasta(str(fa), 24, 4)
    a = pa.Pseudoaligner(host)
    tx = pa.Txome.from_host_index(host)
    n, L, wpr = 400_000, 100, 4
    dev = torch.device("cuda", 0)
    h_tiles, h_lens = tx.simulate_host(L, 9, n, 20000, 0, wpr)                   # 2 % substitutions: some novel / empty / unmapped too
    d_tiles = torch.from_numpy(h_tiles.view(np.int64)).to(dev)
    d_lens = torch.from_numpy(h_lens.view(np.int32)).to(dev)
    cap = a.arena_hint(n)
    d_res = torch.zeros(n * 4, dtype=torch.int32, device=dev)
    d_arena = torch.zeros(cap, dtype=torch.int32, device=dev)
    d_counts = torch.zeros(a.counts_len(), dtype=torch.int64, device=dev)
    for _ in range(3):
        a.map_count_batch_device(d_tiles.data_ptr(), d_lens.data_ptr(), n, wpr, d_res.data_ptr(), d_arena.data_ptr(), cap, d_counts.data_ptr(), 2)
        a.map_finish()
    o_res, o_coff, o_ids, _ = helpers.Oracle(host).map_tiles(h_tiles, h_lens, wpr, 2, 8)
    want = helpers.counts_reference(o_res, o_coff, o_ids, host)
    assert np.array_equal(d_counts.cpu().numpy(), 3 * want) and int(want.sum()) == n


def test_full_size_batch_properties(aligners):
    """BASELINE.json configs[1] size (10 M x 100 bp on gencode_small, K=24) through size-independent properties:
    every error-free read maps over its full length with 0 mismatches and a non-empty class; the count table adds up;
    a re-run is idempotent; ALL 10 M reads are bit-exact against the oracle."""
    import torch
    a = aligners(24)
    tx = pa.Txome.from_host_index(a.host)
    n, wpr = 10_000_000, 4
    dev = torch.device("cuda", 0)
    d_tiles = torch.zeros(pa.lib().pa_tiles_words(n, wpr), dtype=torch.int64, device=dev)
    d_lens = torch.zeros(n, dtype=torch.int32, device=dev)
    tx.simulate_device(100, 1, n, d_tiles.data_ptr(), d_lens.data_ptr(), 0, 0, wpr)
    cap = a.arena_hint(n)
    d_res = torch.zeros(n * 4, dtype=torch.int32, device=dev)
    d_arena = torch.zeros(cap, dtype=torch.int32, device=dev)
    d_col = torch.zeros(n, dtype=torch.int32, device=dev)
    a.map_batch_device(d_tiles.data_ptr(), d_lens.data_ptr(), n, wpr, d_res.data_ptr(), d_arena.data_ptr(), cap, 2, d_col.data_ptr())
    used, _ = a.map_finish()
    res = d_res.view(n, 4)
    assert bool((res[:, 0] == 100).all()) and bool((res[:, 1] == -2**31).all()) and bool((res[:, 3] > 0).all())
    d_counts = torch.zeros(a.counts_len(), dtype=torch.int64, device=dev)
    a.counts_accumulate_device(d_res.data_ptr(), d_arena.data_ptr(), d_col.data_ptr(), n, d_counts.data_ptr())
    torch.cuda.synchronize()
    assert int(d_counts.sum().item()) == n and int(d_counts[-3:].sum().item()) == int(d_counts[-3].item())
    checksum1 = (int(res[:, 3].sum().item()), int(res[:, 0].sum().item()))
    # checksum of the class ids, independent of arena placement
    # ALL 10 M reads bit-exact against the oracle, 2.5 M at a time (src/pseudoaligner.rs:361-376)
    arena_h = d_arena[: max(used, 1)].cpu().numpy().view(np.uint32)
    oracle = helpers.Oracle(a.host)
    chunk = 2_500_000 // 64 * 64
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        part = d_res[lo * 4: (lo + m) * 4].cpu().numpy().view(pa.RESULT_DTYPE)
        coff, cids = pa.gather_classes(part, arena_h, a.host)
        h_tiles, h_lens = tx.simulate_host(100, 1, m, 0, lo, wpr)
        t_lo = lo // 64 * wpr * 64
        assert np.array_equal(d_tiles[t_lo: t_lo + len(h_tiles)].cpu().numpy().view(np.uint64), h_tiles)
        o_res, o_coff, o_ids, _ = oracle.map_tiles(h_tiles, h_lens, wpr, 2, min(16, os.cpu_count() or 1))
        helpers.assert_same_as_oracle(part, coff, cids, o_res, o_coff, o_ids, "10M batch reads [%d, %d)" % (lo, lo + m))
    a.map_batch_device(d_tiles.data_ptr(), d_lens.data_ptr(), n, wpr, d_res.data_ptr(), d_arena.data_ptr(), cap, 2, d_col.data_ptr())
    a.map_finish()
    assert (int(res[:, 3].sum().item()), int(res[:, 0].sum().item())) == checksum1


def _map_with_overflow(a, ovf, reads, allowed, repeats=1):
    """fused count launches with an overflow table attached -> (dense table, overflow dict)"""
    import torch
    tiles, lens, wpr = pa.encode_reads_host(reads)
    dev = torch.device("cuda", 0)
    d_tiles, d_lens = torch.from_numpy(tiles.view(np.int64)).to(dev), torch.from_numpy(np.asarray(lens, np.uint32).view(np.int32)).to(dev)
    n = len(reads)
    cap = a.arena_hint(n)
    d_res = torch.zeros(n * 4, dtype=torch.int32, device=dev)
    d_arena = torch.zeros(cap, dtype=torch.int32, device=dev)
    d_counts = torch.zeros(a.counts_len(), dtype=torch.int64, device=dev)
    a.set_overflow(ovf)
    for _ in range(repeats):
        a.map_count_batch_device(d_tiles.data_ptr(), d_lens.data_ptr(), n, wpr, d_res.data_ptr(), d_arena.data_ptr(), cap, d_counts.data_ptr(), allowed)
        a.map_finish()
    return d_counts


@pytest.mark.parametrize("seed,big", [(1, False), (4, False), (7, False), (0, True), (2, True)])
def test_overflow_table_holds_the_novel_classes(tmp_path, seed, big):
    """the dense table lumps every result that is no index class into ONE slot; the overflow table attached to the
    index must hold exactly those id sets with their read counts (oracle histogram), also across repeated launches"""
    host, k, reads, clean, allowed = helpers.random_txome_case(seed, tmp_path, big=big)
    if host is None:
        pytest.skip("every transcript is shorter than k")
    a = pa.Pseudoaligner(host)
    ovf = pa.Overflow(0, 1 << 14, 1 << 20)
    d_counts = _map_with_overflow(a, ovf, reads, allowed, repeats=2)
    o_res, o_coff, o_ids, _ = helpers.Oracle(host).map_reads(clean, allowed, 4)
    want = helpers.novel_reference(o_res, o_coff, o_ids, host)
    got = pa.parse_overflow(ovf.fetch())
    assert got == {ids: 2 * c for ids, c in want.items()}
    counts = d_counts.cpu().numpy()
    assert int(counts[-3]) == sum(got.values()) and np.array_equal(counts, 2 * helpers.counts_reference(o_res, o_coff, o_ids, host))
    assert np.array_equal(ovf.fetch(), pa.serialise_overflow(got))               # canonical order
    ovf.reset()
    assert pa.parse_overflow(ovf.fetch()) == {}
    a.set_overflow(None)


def test_overflow_capacity_is_reported(tmp_path):
    host, k, reads, clean, allowed = helpers.random_txome_case(0, tmp_path, big=True)
    a = pa.Pseudoaligner(host)
    want = helpers.novel_reference(*helpers.Oracle(host).map_reads(clean, allowed, 4)[:3], host)
    assert len(want) > 8
    tiny = pa.Overflow(0, 4, 16)                                                 # room for far fewer classes / ids than occur
    _map_with_overflow(a, tiny, reads, allowed)
    with pytest.raises(pa.PaError) as e:
        tiny.fetch()
    assert e.value.code == pa._ffi.PA_ERR_ARENA_FULL
    a.set_overflow(None)


def test_rccl_world_of_one_reduces_counts_and_overflow(aligners):
    """the product's own collective entry points (pa_comm_*, pa_counts_allreduce, pa_overflow_allgather) over RCCL with a
    communicator of ONE rank (all a 1-GPU box offers): the reduce must leave the table as it is, the gather must return this
    GPU's canonical overflow table"""
    import torch
    a = aligners(24)
    tx = pa.Txome.from_host_index(a.host)
    n, wpr = 200_000, 4
    dev = torch.device("cuda", 0)
    h_tiles, h_lens = tx.simulate_host(100, 11, n, 30000, 0, wpr)                # 3 % substitutions: plenty of novel classes
    d_tiles = torch.from_numpy(h_tiles.view(np.int64)).to(dev)
    d_lens = torch.from_numpy(h_lens.view(np.int32)).to(dev)
    cap = a.arena_hint(n)
    d_res = torch.zeros(n * 4, dtype=torch.int32, device=dev)
    d_arena = torch.zeros(cap, dtype=torch.int32, device=dev)
    d_counts = torch.zeros(a.counts_len(), dtype=torch.int64, device=dev)
    ovf = pa.Overflow(0, 1 << 16, 1 << 22)
    a.set_overflow(ovf)
    a.map_count_batch_device(d_tiles.data_ptr(), d_lens.data_ptr(), n, wpr, d_res.data_ptr(), d_arena.data_ptr(), cap, d_counts.data_ptr(), 2)
    a.map_finish()
    before = d_counts.cpu().numpy().copy()
    comm = pa.Comm(0, 1, 0, pa.Comm.unique_id())
    assert (comm.rank, comm.size) == (0, 1)
    a.counts_allreduce(d_counts.data_ptr(), comm)
    torch.cuda.synchronize()
    assert np.array_equal(d_counts.cpu().numpy(), before)
    gathered = ovf.allgather(comm)
    assert np.array_equal(gathered, ovf.fetch()) and np.array_equal(gathered, ovf.allgather(None))
    o_res, o_coff, o_ids, _ = helpers.Oracle(a.host).map_tiles(h_tiles, h_lens, wpr, 2, 8)
    want = helpers.novel_reference(o_res, o_coff, o_ids, a.host)
    assert pa.parse_overflow(gathered) == want and sum(want.values()) == int(before[-3]) > 0
    a.set_overflow(None)


def test_c_client_of_the_header_runs_its_device_half(tmp_path):
    """integration/c/abi_check.c (plain C, every entry point of the header once) on the GPU: maps, counts, overflow table, RCCL
    world of one, encode kernel — the call sequence a Rust host would make, without Python in between"""
    import subprocess
    exe = helpers._build.build_abi_check()
    out = subprocess.run([str(exe), str(helpers.FASTA), str(helpers.FASTQ), str(tmp_path)], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "device halves ok" in out.stdout and "0 failures" in out.stdout, out.stdout + out.stderr


def test_concurrent_launches_on_two_streams_from_two_threads(aligners):
    """"pa_map_batch callable concurrently from multiple host threads (one stream each)": two threads, two streams,
    ONE index handle, ONE count table; every launch's records and the summed table equal the sequential results"""
    import threading
    import torch
    a = aligners(24)
    tx = pa.Txome.from_host_index(a.host)
    n, wpr, rounds = 400_000, 4, 4
    dev = torch.device("cuda", 0)
    d_counts = torch.zeros(a.counts_len(), dtype=torch.int64, device=dev)
    inputs, want_res = [], []
    for t in range(2):
        h_tiles, h_lens = tx.simulate_host(100, 21 + t, n, 15000, 0, wpr)
        inputs.append((torch.from_numpy(h_tiles.view(np.int64)).to(dev), torch.from_numpy(h_lens.view(np.int32)).to(dev)))
        want_res.append(helpers.Oracle(a.host).map_tiles(h_tiles, h_lens, wpr, 2, 8))
    torch.cuda.synchronize()
    out, errors = [None, None], []

    def worker(t):
        try:
            s = torch.cuda.Stream(device=dev)
            cap = a.arena_hint(n)
            d_res = torch.zeros(n * 4, dtype=torch.int32, device=dev)
            d_arena = torch.zeros(cap, dtype=torch.int32, device=dev)
            for _ in range(rounds):
                a.map_count_batch_device(inputs[t][0].data_ptr(), inputs[t][1].data_ptr(), n, wpr, d_res.data_ptr(), d_arena.data_ptr(), cap,
                                         d_counts.data_ptr(), 2, s.cuda_stream)
                used, _ = a.map_finish(s.cuda_stream)
            out[t] = (d_res.cpu().numpy().view(pa.RESULT_DTYPE), d_arena[: max(used, 1)].cpu().numpy().view(np.uint32))
        except Exception as e:   # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    torch.cuda.synchronize()
    want_counts = np.zeros(a.counts_len(), np.int64)
    for t in range(2):
        o_res, o_coff, o_ids, _ = want_res[t]
        coff, cids = pa.gather_classes(out[t][0], out[t][1], a.host)
        helpers.assert_same_as_oracle(out[t][0], coff, cids, o_res, o_coff, o_ids, "thread %d" % t)
        want_counts += rounds * helpers.counts_reference_fast(o_res, o_coff, o_ids, a.host)
    assert np.array_equal(d_counts.cpu().numpy(), want_counts)


def test_per_barcode_counts_equal_the_histogram(aligners):
    """SURVEY §8f.3 (single-cell use, README.md:3): the sparse (barcode, class) -> reads matrix from the GPU (keys sorted,
    barcode << 32 | column) equals the histogram of (barcode, column) built from the oracle's per-read results"""
    import torch
    a = aligners(24)
    tx = pa.Txome.from_host_index(a.host)
    n, wpr, ncells = 300_000, 4, 700
    dev = torch.device("cuda", 0)
    h_tiles, h_lens = tx.simulate_host(100, 31, n, 20000, 0, wpr)
    rng = np.random.RandomState(8)
    barcode = rng.randint(0, ncells, n).astype(np.uint32)
    d_tiles = torch.from_numpy(h_tiles.view(np.int64)).to(dev)
    d_lens = torch.from_numpy(h_lens.view(np.int32)).to(dev)
    d_bc = torch.from_numpy(barcode.view(np.int32)).to(dev)
    cap = a.arena_hint(n)
    d_res = torch.zeros(n * 4, dtype=torch.int32, device=dev)
    d_arena = torch.zeros(cap, dtype=torch.int32, device=dev)
    a.map_batch_device(d_tiles.data_ptr(), d_lens.data_ptr(), n, wpr, d_res.data_ptr(), d_arena.data_ptr(), cap, 2, 0)
    a.map_finish()
    d_keys = torch.zeros(n, dtype=torch.int64, device=dev)
    d_vals = torch.zeros(n, dtype=torch.int32, device=dev)
    cells = a.counts_by_barcode_device(d_res.data_ptr(), d_arena.data_ptr(), d_bc.data_ptr(), n, d_keys.data_ptr(), d_vals.data_ptr(), 10)
    keys = d_keys[:cells].cpu().numpy().view(np.uint64)
    vals = d_vals[:cells].cpu().numpy().view(np.uint32)
    # the checker: column of every read from the oracle's classes
    o_res, o_coff, o_ids, _ = helpers.Oracle(a.host).map_tiles(h_tiles, h_lens, wpr, 2, 8)
    arr = a.host.arrays()
    off = arr["ec_offset"].astype(np.int64)
    nc = arr["num_classes"]
    table = {tuple(arr["ec_ids"][off[c]:off[c + 1]].tolist()): c for c in range(nc)}
    col = np.zeros(n, np.uint64)
    for i in range(n):
        if not o_res["mapped"][i]:
            col[i] = nc + 2
        elif o_res["class_len"][i] == 0:
            col[i] = nc + 1
        else:
            col[i] = table.get(tuple(o_ids[int(o_coff[i]):int(o_coff[i + 1])].tolist()), nc)
    want_keys, want_vals = np.unique((barcode.astype(np.uint64) << np.uint64(32)) | col, return_counts=True)
    assert np.array_equal(keys, want_keys) and np.array_equal(vals, want_vals.astype(np.uint32)) and int(vals.sum()) == n
    assert a.counts_by_barcode_device(d_res.data_ptr(), d_arena.data_ptr(), d_bc.data_ptr(), 0, d_keys.data_ptr(), d_vals.data_ptr()) == 0


@pytest.mark.parametrize("seed", range(8))
def test_long_chain_left_extensions(tmp_path, seed):
    """helpers.long_chain_case: reads of 300..500 bases whose first hit lies far into the read, nested transcripts cut on and around
    multiples of 64, allowed 6 / 12 — the LEFT path across chain blocks (a node that starts exactly on a block seam)"""
    host, reads, allowed = helpers.long_chain_case(seed, tmp_path)
    a = pa.Pseudoaligner(host)
    _, _, _, ctr = gpu_vs_oracle(a, reads, allowed, "long chains seed %d" % seed)
    assert ctr["left_extensions"] > 50


@pytest.mark.parametrize("long_len", [120000, 500000])
def test_transcripts_of_hundreds_of_kilobases_map_onto_themselves(tmp_path, long_len):
    """validate_dbg's property (src/build_index.rs:300-367: a transcript mapped onto its own graph has coverage == its length and a class
    that contains it) for transcripts of 120 kb / 500 kb — the reference has no length limit and its own test maps whole transcripts (:309).
    Reads of more than 512 bases stay in HBM and take the WIDE lane state (lane_steps.hpp); bit-exact against the oracle, with substitutions too;
    pa_map_read on the longest one"""
    host, seqs = helpers.long_transcript_case(tmp_path, long_len=long_len)
    a = pa.Pseudoaligner(host, 0)
    res, coff, cids, ctr = gpu_vs_oracle(a, seqs, 2, "transcripts as reads, longest %d" % long_len)
    for t, s in enumerate(seqs):
        if len(s) >= host.k:
            assert res["mismatches"][t] >> 31 and res["coverage"][t] == len(s) and t in cids[int(coff[t]):int(coff[t + 1])].tolist(), t
    rng = np.random.RandomState(3)
    noisy = []
    for s in seqs[-2:] + seqs[:6]:
        r_ = list(s)
        for j in rng.randint(0, len(r_), max(1, len(r_) // 400)):
            r_[j] = "ACGT"[("ACGT".index(r_[j]) + 1 + rng.randint(3)) % 4]
        noisy.append("".join(r_))
    _, _, _, ctr = gpu_vs_oracle(a, noisy, 2, "long reads with substitutions")
    assert ctr["reseeks"] > 50
    got = a.map_read(seqs[-2])
    assert got is not None and got[1] == long_len and (len(seqs) - 2) in got[0]
