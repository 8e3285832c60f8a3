"""`-m gpu` tier of the window pipeline behind pa_process_reads / pa_process_reads_multi (src/pseudoaligner.rs:420-514): windows of raw
FASTQ text go to HBM as the file holds them, the GPU finds the records (csrc/fastq_scan.hip), sequences and ids are read in place.
Whatever the window size, the number of lanes, and whether the GPU or the host finds the records: the same bytes come out, and they
are the oracle's tuples."""
import os

import numpy as np
import pytest

import helpers

pa = helpers.pa
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def aligner(small_index):
    if pa.lib().pa_device_count() < 1:
        raise RuntimeError("the gpu tier needs a GPU and the HIP library: %s" % pa.lib().pa_last_error().decode())
    return pa.Pseudoaligner(small_index(24), 0)


def expected_lines(a, ids, seqs):
    res, coff, cids, _ = helpers.Oracle(a.host).map_reads(seqs, 2, 8)
    want = []
    for i, rid in enumerate(ids):
        cl = cids[int(coff[i]):int(coff[i + 1])].tolist()
        flag = bool(res["mapped"][i]) and res["coverage"][i] >= 32 and not cl
        want.append('(%s, "%s", [%s], %d)' % ("true" if flag else "false", rid, ", ".join(map(str, cl)), res["coverage"][i] if res["mapped"][i] else 0))
    return want


def make_reads(n, seed, lo=1, hi=181):
    ids, seqs = helpers.read_fastq()
    rng = np.random.default_rng(seed)
    pick = rng.integers(0, len(seqs), n)
    out_ids, out_seqs = [], []
    for j, i in enumerate(pick):
        s = seqs[i] + seqs[(i + 1) % len(seqs)] + seqs[(i + 2) % len(seqs)]
        out_seqs.append(s[: int(rng.integers(lo, hi))])
        out_ids.append("%s/%d" % (ids[i], j))
    return out_ids, out_seqs


def write_fastq(path, ids, seqs, eol="\n", tail=""):
    path.write_text("".join("@%s some description%s%s%s+%s%s%s" % (i, eol, s, eol, eol, "I" * len(s), eol) for i, s in zip(ids, seqs)) + tail, newline="")


@pytest.mark.parametrize("window", [700, 4096, 65536, 1 << 20])
def test_lanes_and_windows_give_the_same_bytes(aligner, tmp_path, monkeypatch, window):
    """pa_process_reads_multi over 1, 2 and 3 lanes (the same handle listed several times: streams of one GPU) and pa_process_reads:
    byte-identical output == the oracle's tuples, windows that end inside every kind of line"""
    ids, seqs = make_reads(20000, 3)
    want = expected_lines(aligner, ids, seqs)
    fq = tmp_path / "r.fq"
    write_fastq(fq, ids, seqs)
    monkeypatch.setenv("PA_INGEST_WINDOW", str(window))
    outs = []
    for lanes in (0, 1, 2, 3):
        out = tmp_path / ("o%d.txt" % lanes)
        if lanes == 0:
            n, flagged = pa.process_reads(str(fq), aligner, str(out), 4)
        else:
            n, flagged = pa.process_reads_multi(str(fq), [aligner] * lanes, str(out), 4)
        assert n == len(ids) and flagged == sum(1 for w in want if w.startswith("(true")), lanes
        outs.append(out.read_bytes())
    assert outs[0].decode().splitlines() == want
    assert all(o == outs[0] for o in outs[1:])
    st = pa.process_reads_stage_seconds()
    assert st["reads"] == len(ids) and st["scan_s"] < st["total_s"]


def test_gpu_scan_and_host_scan_agree(aligner, tmp_path, monkeypatch):
    """the records the GPU finds are the records the host's scan finds: CRLF, ids with spaces / tabs / trailing blanks / quotes /
    control bytes, empty sequences, reads shorter than k, a window far smaller than the line table's first guess (short reads: the
    scan regrows it and fills it again)"""
    ids, seqs = make_reads(30000, 5, 0, 40)           # short reads: more lines per byte than the first guess allows for
    ids[7], ids[8], ids[9] = 'qu"ote', "back\\slash\x01ctl", "tab\tinside"
    seqs[11] = ""
    for eol in ("\n", "\r\n"):
        fq = tmp_path / ("s%d.fq" % len(eol))
        fq.write_text("".join("@%s  two spaces and a tab\t %s%s%s+%s%s%s" % (i, eol, s, eol, eol, "#" * len(s), eol) for i, s in zip(ids, seqs)), newline="")
        got = {}
        for mode in ("gpu", "host"):
            if mode == "host":
                monkeypatch.setenv("PA_INGEST_HOST_SCAN", "1")
            else:
                monkeypatch.delenv("PA_INGEST_HOST_SCAN", raising=False)
            monkeypatch.setenv("PA_INGEST_WINDOW", "300000")
            out = tmp_path / "o.txt"
            n, _ = pa.process_reads(str(fq), aligner, str(out), 3)
            assert n == len(ids)
            got[mode] = out.read_bytes()
        monkeypatch.delenv("PA_INGEST_HOST_SCAN", raising=False)
        assert got["gpu"] == got["host"]
        lines = got["gpu"].decode().splitlines()
        want = expected_lines(aligner, ids, seqs)
        want[7] = want[7].replace('qu"ote', 'qu\\"ote')
        want[8] = want[8].replace("back\\slash\x01ctl", "back\\\\slash\\u{1}ctl")
        want[9] = want[9].replace("tab\tinside", "tab\\tinside")
        assert lines == want, eol


def test_default_window_and_long_headers(aligner, tmp_path, monkeypatch):
    """the production window (128 MiB: the whole of a small file minus its end, which is the host's), a record longer than a small
    window's head room would be, and gzip text through the same windows"""
    import gzip
    monkeypatch.delenv("PA_INGEST_WINDOW", raising=False)
    monkeypatch.delenv("PA_INGEST_BATCH", raising=False)
    ids, seqs = make_reads(60000, 8, 20, 181)
    ids[100] = "long" + "x" * 5000                       # longer than a window of the small-window run below
    want = expected_lines(aligner, ids, seqs)
    fq = tmp_path / "d.fq"
    write_fastq(fq, ids, seqs, tail="\n\n")
    out = tmp_path / "o.txt"
    assert pa.process_reads(str(fq), aligner, str(out), 8)[0] == len(ids)
    assert out.read_text().splitlines() == want
    monkeypatch.setenv("PA_INGEST_WINDOW", "1500")
    assert pa.process_reads_multi(str(fq), [aligner, aligner], str(out), 8)[0] == len(ids)
    assert out.read_text().splitlines() == want
    gz = tmp_path / "d.fq.gz"
    gz.write_bytes(gzip.compress(fq.read_bytes(), 1))
    monkeypatch.setenv("PA_INGEST_WINDOW", "50000")
    assert pa.process_reads(str(gz), aligner, str(out), 8)[0] == len(ids)
    assert out.read_text().splitlines() == want


@pytest.mark.parametrize("lanes", [1, 2])
def test_tuples_that_grow_from_window_to_window(aligner, tmp_path, monkeypatch, lanes):
    """a window's tuples are rendered and sent back ahead of knowing their length, sized by the window before (on the lane's copy-back
    stream); text that turns out longer is rendered again. Ids that grow in steps — 8, 60, 400, 30, 1500 characters — make every kind
    of window: longer than guessed, much shorter, first of its call"""
    ids, seqs = make_reads(30000, 21, 30, 181)
    steps = [8, 60, 400, 30, 1500, 12]
    ids = [("%07d" % j) + "q" * (steps[j * len(steps) // len(ids)] - 7) for j in range(len(ids))]
    want = expected_lines(aligner, ids, seqs)
    fq = tmp_path / "g.fq"
    fq.write_text("".join("@%s\n%s\n+\n%s\n" % (i, s, "I" * len(s)) for i, s in zip(ids, seqs)))
    out = tmp_path / "o.txt"
    for window in (40000, 300000):
        monkeypatch.setenv("PA_INGEST_WINDOW", str(window))
        for _ in range(2):   # (the second call starts with the first one's parked buffers and guesses)
            assert pa.process_reads_multi(str(fq), [aligner] * lanes, str(out), 8)[0] == len(ids)
            assert out.read_text().splitlines() == want


def test_replicas_are_checked(aligner, small_index, tmp_path):
    other = pa.Pseudoaligner(small_index(20), 0)
    fq = tmp_path / "x.fq"
    fq.write_text("@r\nACGT\n+\nIIII\n")
    with pytest.raises(pa.PaError):
        pa.process_reads_multi(str(fq), [aligner, other], str(tmp_path / "o.txt"), 2)


def test_progress_line_is_the_references(small_index, tmp_path, monkeypatch, capfd):
    """`Done Mapping {} reads w/ Rate: {}` (src/pseudoaligner.rs:497-503) at exactly every 10^6-th read with the flagged count of exactly
    the first 10^6 m reads, the rate printed as Rust prints an f32 (shortest digits that round-trip, no exponent); 2.3 M reads of
    small.fq (12 in 9 309 flagged at K = 20), windows that do not end on a million"""
    monkeypatch.setenv("PA_INGEST_WINDOW", str(24 << 20))
    a = pa.Pseudoaligner(small_index(20), 0)
    _, seqs = helpers.read_fastq()
    rng = np.random.default_rng(2)
    n = 2300000
    pick = rng.integers(0, len(seqs), n)
    fq = tmp_path / "big.fq"
    with open(fq, "w") as f:
        for lo in range(0, n, 100000):
            f.write("".join("@r%d\n%s\n+\n%s\n" % (j, seqs[pick[j]], "I" * len(seqs[pick[j]])) for j in range(lo, min(n, lo + 100000))))
    out = tmp_path / "o.txt"
    capfd.readouterr()
    got_n, flagged = pa.process_reads(str(fq), a, str(out), 8)
    err = capfd.readouterr().err
    assert got_n == n
    flags = np.fromiter((l.startswith("(true") for l in open(out)), bool, n)
    assert flagged == int(flags.sum()) and flagged > 1000
    marks = [m for m in err.split("\r") if m.startswith("Done Mapping")]
    assert len(marks) == 2, err
    for m, at in zip(marks, (1000000, 2000000)):
        rate = np.float32(np.float32(flags[:at].sum()) * np.float32(100.0) / np.float32(at))
        assert m.rstrip("\n") == "Done Mapping %d reads w/ Rate: %s" % (at, np.format_float_positional(rate, unique=True, trim="-")), (m, rate)
    assert err.endswith("\n")                                                  # `eprintln!()` behind the progress line (:508)


@pytest.mark.parametrize("k,ppm", [(24, 0), (24, 20000), (31, 40000)])
def test_compact_records_are_the_records(small_index, k, ppm):
    """pa_results_compact_device: the 8-byte records + the packed stream of the classes that are no index classes, unpacked on the host,
    are what map_read_with_mismatch returns (src/pseudoaligner.rs:361-376) — against the oracle, with novel classes present"""
    import torch
    host = small_index(k)
    a = pa.Pseudoaligner(host, 0)
    tx = pa.Txome.from_host_index(host)
    n, read_len = 300000, 150
    wpr = pa.lib().pa_words_per_read(read_len)
    tiles, lens = tx.simulate_host(read_len, 6, n, ppm, 0, wpr)
    dev = torch.device("cuda", 0)
    d_tiles = torch.from_numpy(tiles.view(np.int64)).to(dev)
    d_lens = torch.from_numpy(lens.view(np.int32)).to(dev)
    cap = a.arena_hint(n)
    d_res = torch.zeros(n * 4, dtype=torch.int32, device=dev)
    d_arena = torch.zeros(cap, dtype=torch.int32, device=dev)
    a.map_batch_device(d_tiles.data_ptr(), d_lens.data_ptr(), n, wpr, d_res.data_ptr(), d_arena.data_ptr(), cap, 2)
    used, need = a.map_finish()
    scr = pa.lib().pa_compact_scratch_bytes(n)
    d_scr = torch.zeros(scr, dtype=torch.uint8, device=dev)
    d_compact = torch.zeros(n, dtype=torch.int64, device=dev)
    d_packed = torch.zeros(cap, dtype=torch.int32, device=dev)
    d_pw = torch.zeros(1, dtype=torch.int64, device=dev)
    pa.check(pa.lib().pa_results_compact_device(a._h, d_res.data_ptr(), d_arena.data_ptr(), cap, n, d_compact.data_ptr(), d_packed.data_ptr(), cap, d_pw.data_ptr(),
                                                d_scr.data_ptr(), scr, None))
    torch.cuda.synchronize()
    pw = int(d_pw.item())
    assert pw <= used + n                                    # no padding travels: ids + one length word per packed class
    res, coff, ids = pa.unpack_compact(d_compact.cpu().numpy().view(np.uint64), d_packed[:pw].cpu().numpy().view(np.uint32), host)
    want = helpers.Oracle(host).map_tiles(tiles, lens, wpr, 2, 8)
    helpers.assert_same_as_oracle(res, coff, ids, want[0], want[1], want[2], "compact records K=%d ppm=%d" % (k, ppm))
    npacked = int(((d_compact.cpu().numpy().view(np.uint64) & np.uint64(pa.PA_COMPACT_PACKED)) != 0).sum())
    assert (npacked > 100) == (ppm > 0) or npacked > 0      # error reads produce intersections that are no index class


@pytest.mark.parametrize("uniform", [False, True])
def test_host_to_host_batches(small_index, uniform):
    """pa_map_tiles_host (host tiles -> per-read outputs + count table on the host): chunks of 4 096 reads over three streams,
    ragged and uniform batches, compact records + packed classes unpacked == the oracle's results, the count table == their histogram;
    a packed buffer that is too small is reported, not overrun"""
    import torch
    host = small_index(24)
    a = pa.Pseudoaligner(host, 0)
    tx = pa.Txome.from_host_index(host)
    n, read_len = 150000, 100
    wpr = pa.lib().pa_words_per_read(read_len)
    tiles, lens = tx.simulate_host(read_len, 12, n, 20000, 0, wpr)
    if not uniform:                                  # ragged: shorten every third read (the tile words beyond a read's length are ignored)
        lens = lens.copy()
        lens[::3] = (np.arange(len(lens[::3])) % 101).astype(np.uint32)
    h_tiles = torch.from_numpy(tiles.view(np.int64)).pin_memory()
    h_lens = torch.from_numpy(lens.view(np.int32)).pin_memory()
    h_compact = torch.zeros(n, dtype=torch.int64).pin_memory()
    cap = a.arena_hint(n)
    h_packed = torch.zeros(cap, dtype=torch.int32).pin_memory()
    h_counts = torch.zeros(a.counts_len(), dtype=torch.int64).pin_memory()
    for _ in range(2):                               # (the second call finds streams and staging buffers parked on the handle)
        words = a.map_tiles_host(h_tiles.data_ptr(), n, wpr, h_compact.data_ptr(), h_packed.data_ptr(), cap, h_lens=0 if uniform else h_lens.data_ptr(),
                                 uniform_len=read_len if uniform else 0, h_counts=h_counts.data_ptr(), chunk_reads=4096, n_streams=3)
    res, coff, ids = pa.unpack_compact(h_compact.numpy().view(np.uint64), h_packed[:words].numpy().view(np.uint32), host)
    want = helpers.Oracle(host).map_tiles(tiles, lens, wpr, 2, 8)
    helpers.assert_same_as_oracle(res, coff, ids, want[0], want[1], want[2], "host to host, uniform=%s" % uniform)
    assert np.array_equal(h_counts.numpy().astype(np.int64), helpers.counts_reference_fast(want[0], want[1], want[2], host))
    assert words > 100
    with pytest.raises(pa.PaError) as err:
        a.map_tiles_host(h_tiles.data_ptr(), n, wpr, h_compact.data_ptr(), h_packed.data_ptr(), 50, h_lens=0 if uniform else h_lens.data_ptr(),
                         uniform_len=read_len if uniform else 0, chunk_reads=4096, n_streams=3)
    assert err.value.code == pa._ffi.PA_ERR_ARENA_FULL


def test_record_stream_over_several_lanes(aligner):
    """pa_record_stream_create_multi: full batches go round-robin to the lanes (the same handle listed one, two and three times: streams of one GPU), the
    tuples come back in push order — byte for byte what a single lane gives, and what the oracle says"""
    ids, seqs = make_reads(9000, 21, 0, 181)
    want = "\n".join(expected_lines(aligner, ids, seqs)) + "\n"
    rng = np.random.default_rng(4)
    for lanes, batch in ((1, 512), (2, 512), (3, 320), (2, 64)):
        rs = pa.RecordStream([aligner] * lanes, 3, batch)
        got, i = b"", 0
        while i < len(ids):
            n = int(rng.integers(1, 1500))
            rs.push(ids[i:i + n], seqs[i:i + n])
            got += rs.drain(1 << 14)
            i += n
        rs.flush()
        got += rs.drain()
        assert rs.stats()[0] == len(ids)
        assert got.decode() == want, (lanes, batch)
        rs.flush()
        assert rs.drain() == b""
        rs.close()
