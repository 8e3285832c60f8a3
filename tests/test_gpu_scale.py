"""`-m gpu` tier at BASELINE.json's scale: the ~202 k-transcript synthetic index of configs 3/4/5 (bench.py builds the same
one: Txome.synthesize(58000, 203000, 7)) through the HIP path, against the oracle.

  config 3  K = 24, error-free 150 bp reads (seed 2): 2 M reads bit-exact incl. the fused class-count table; the full
            100 M-read batch bit-exact against the oracle (every read, 10 M at a time) + size-independent properties
  config 5  K = 31, 1 % substitutions (seed 4): the re-seek (src/pseudoaligner.rs:293-299) and left-extension (:124-205)
            paths at scale: 2 M reads bit-exact incl. the count table; the full 100 M-read batch bit-exact (every read)
  config 4  is config 3 sharded over ranks: the shard arithmetic of bench.py (rank r maps reads [r*N, (r+1)*N) of ONE global
            stream) is checked here on one GPU — per-shard tables add up to the table of the whole range
and the committed golden error-read fixtures (tests/golden/synth_err_k31.tsv, synth_err_k64.tsv) against the HIP path.
"""
import os

import numpy as np
import pytest

import helpers

pa = helpers.pa
pytestmark = pytest.mark.gpu

GENES, TRANSCRIPTS, TX_SEED = 58000, 203000, 7
FULL_BATCH = int(os.environ.get("PA_TEST_FULL_BATCH", 100_000_000))
FULL_CHUNK = int(os.environ.get("PA_TEST_FULL_CHUNK", 10_000_000))   # reads per oracle comparison (multiple of 64)


@pytest.fixture(scope="module")
def txome():
    if pa.lib().pa_device_count() < 1:
        raise RuntimeError("the gpu tier needs a GPU and the HIP library: %s" % pa.lib().pa_last_error().decode())
    return pa.Txome.synthesize(GENES, TRANSCRIPTS, TX_SEED)


@pytest.fixture(scope="module")
def big(txome):
    """(host index, device index, oracle) of the GENCODE-scale transcriptome per k, one at a time (the K=24 objects are dropped
    before K=31 is built: host index + oracle index are a few GB each)."""
    state = {}

    def get(k):
        if state.get("k") != k:
            state.clear()
            host = pa.HostIndex.from_txome(txome, k, 0)
            state.update(k=k, host=host, aligner=pa.Pseudoaligner(host, 0), oracle=helpers.Oracle(host))
        return state["host"], state["aligner"], state["oracle"]
    return get


def _map_device(aligner, d_tiles, d_lens, n, wpr, allowed, with_counts):
    import torch
    dev = d_tiles.device
    cap = aligner.arena_hint(n)
    d_res = torch.empty(n * 4, dtype=torch.int32, device=dev)
    d_arena = torch.empty(cap, dtype=torch.int32, device=dev)
    d_counts = torch.zeros(aligner.counts_len(), dtype=torch.int64, device=dev) if with_counts else None
    if with_counts:
        aligner.map_count_batch_device(d_tiles.data_ptr(), d_lens.data_ptr(), n, wpr, d_res.data_ptr(), d_arena.data_ptr(), cap,
                                       d_counts.data_ptr(), allowed)
    else:
        aligner.map_batch_device(d_tiles.data_ptr(), d_lens.data_ptr(), n, wpr, d_res.data_ptr(), d_arena.data_ptr(), cap, allowed, 0)
    used, _ = aligner.map_finish()
    return d_res, d_arena, used, d_counts


def _bit_exact_with_counts(txome, host, aligner, oracle, k, read_len, seed, ppm, n, first, what):
    import torch
    dev = torch.device("cuda", 0)
    wpr = pa.lib().pa_words_per_read(read_len)
    d_tiles = torch.empty(pa.lib().pa_tiles_words(n, wpr), dtype=torch.int64, device=dev)
    d_lens = torch.empty(n, dtype=torch.int32, device=dev)
    txome.simulate_device(read_len, seed, n, d_tiles.data_ptr(), d_lens.data_ptr(), ppm, first, wpr)
    h_tiles, h_lens = txome.simulate_host(read_len, seed, n, ppm, first, wpr)
    torch.cuda.synchronize()
    assert np.array_equal(d_tiles.cpu().numpy().view(np.uint64), h_tiles), what + ": device and host simulators disagree"
    d_res, d_arena, used, d_counts = _map_device(aligner, d_tiles, d_lens, n, wpr, 2, True)
    res = d_res.cpu().numpy().view(pa.RESULT_DTYPE)
    coff, cids = pa.gather_classes(res, d_arena[: max(used, 1)].cpu().numpy().view(np.uint32), host)
    o_res, o_coff, o_ids, ctr = oracle.map_tiles(h_tiles, h_lens, wpr, 2, min(16, os.cpu_count() or 1))
    helpers.assert_same_as_oracle(res, coff, cids, o_res, o_coff, o_ids, what)
    want = helpers.counts_reference_fast(o_res, o_coff, o_ids, host)
    assert int(want.sum()) == n
    assert np.array_equal(d_counts.cpu().numpy(), want), what + ": fused count table differs from the histogram of the oracle's results"
    return ctr, want


def test_config3_two_million_reads_bit_exact(txome, big):
    """BASELINE.json configs[2]: K=24, error-free 150 bp reads, seed 2; the first 2 M reads of the stream bench.py times"""
    host, aligner, oracle = big(24)
    st = aligner.stats()
    assert st.num_kmers > 90_000_000 and host.arrays()["num_classes"] > 300_000      # GENCODE scale, not the small fixture
    ctr, want = _bit_exact_with_counts(txome, host, aligner, oracle, 24, 150, 2, 0, 2_000_000, 0, "config 3, 2 M reads")
    assert ctr["mapped"] == 2_000_000 and ctr["reseeks"] == 0 and ctr["left_extensions"] == 0
    assert int(want[-3:].sum()) == int(want[-3])                                      # error-free: nothing empty, nothing unmapped


def test_config4_shards_of_one_stream_add_up(txome, big):
    """BASELINE.json configs[3] (8 GPUs, reads sharded by rank, RCCL-reduced counts) on ONE GPU: the tables of the shards
    [r*N, (r+1)*N) of the seed-3 stream, summed (what the all-reduce computes), equal the table of the whole range mapped
    in one launch, and both equal the oracle's histogram"""
    import torch
    host, aligner, oracle = big(24)
    dev = torch.device("cuda", 0)
    shards, per, wpr = 4, 250_000, pa.lib().pa_words_per_read(150)
    total = torch.zeros(aligner.counts_len(), dtype=torch.int64, device=dev)
    for r in range(shards):
        d_tiles = torch.empty(pa.lib().pa_tiles_words(per, wpr), dtype=torch.int64, device=dev)
        d_lens = torch.empty(per, dtype=torch.int32, device=dev)
        txome.simulate_device(150, 3, per, d_tiles.data_ptr(), d_lens.data_ptr(), 0, r * per, wpr)
        total += _map_device(aligner, d_tiles, d_lens, per, wpr, 2, True)[3]
    n = shards * per
    _, want = _bit_exact_with_counts(txome, host, aligner, oracle, 24, 150, 3, 0, n, 0, "config 4, whole range")
    assert np.array_equal(total.cpu().numpy(), want)


def _full_batch_properties(txome, host, aligner, oracle, read_len, seed, ppm, n, what):
    """one full batch: size-independent properties on the device, then ALL of its reads bit-exact against the oracle in chunks"""
    import torch
    dev = torch.device("cuda", 0)
    wpr = pa.lib().pa_words_per_read(read_len)
    k = aligner.stats().k
    d_tiles = torch.empty(pa.lib().pa_tiles_words(n, wpr), dtype=torch.int64, device=dev)
    d_lens = torch.empty(n, dtype=torch.int32, device=dev)
    txome.simulate_device(read_len, seed, n, d_tiles.data_ptr(), d_lens.data_ptr(), ppm, 0, wpr)
    d_res, d_arena, used, d_counts = _map_device(aligner, d_tiles, d_lens, n, wpr, 2, True)
    res = d_res.view(n, 4)
    cov, mm_raw, coff_col, clen = res[:, 0], res[:, 1], res[:, 2], res[:, 3]
    mapped = mm_raw < 0                                                               # bit 31 = mapped
    mm = mm_raw & 0x7FFFFFFF
    counts = d_counts.cpu().numpy()
    nc = len(counts) - 3
    assert int(counts.sum()) == n                                                     # every read counted exactly once
    assert int(counts[nc + 2]) == n - int(mapped.sum().item())                        # unmapped bucket
    assert int(counts[nc + 1]) == int((mapped & (clen == 0)).sum().item())            # mapped, empty class
    by_ref = mapped & (coff_col < 0) & (clen > 0)                                     # class returned by reference (bit 31 of class_off)
    assert int(counts[:nc].sum()) >= int(by_ref.sum().item())
    assert bool((cov[mapped] <= read_len).all()) and bool((cov[~mapped] == 0).all()) and bool((clen[~mapped] == 0).all())
    assert bool((cov[mapped] >= k).all())                                             # a mapped read matched at least one k-mer (:216)
    if ppm == 0:
        assert bool(mapped.all()) and bool((cov == read_len).all()) and bool((mm == 0).all()) and bool((clen > 0).all())
    else:
        assert int(mapped.sum().item()) > 0.99 * n and bool((mm[mapped] <= cov[mapped]).all())
    # by-reference classes: the length stored with the result is the index's length of that class
    ref_ids = (coff_col[by_ref] & 0x7FFFFFFF).long()
    class_len = torch.from_numpy((host.arrays()["ec_offset"][1:] - host.arrays()["ec_offset"][:-1]).astype(np.int64)).to(dev)
    assert bool((class_len[ref_ids] == clen[by_ref].long()).all())
    # a by-reference result is counted under its class; the other non-empty results (ids in the arena) are counted either as
    # "novel" or — when the content lookup finds that the id set equals an index class after all — under that class
    hist = torch.bincount(ref_ids, minlength=nc)
    extra = torch.from_numpy(counts[:nc]).to(dev) - hist
    in_arena = mapped & (coff_col >= 0) & (clen > 0)
    assert bool((extra >= 0).all()) and int(extra.sum().item()) + int(counts[nc]) == int(in_arena.sum().item())
    checksum = (int(cov.sum().item()), int(mm.sum().item()), int(clen.sum().item()), int(mapped.sum().item()))
    # EVERY read of the batch bit-exact against the oracle (src/pseudoaligner.rs:361-376), in chunks: the host simulator regenerates
    # the chunk's tiles (checked against the device's), the oracle maps them on the box's threads, records and class ids are compared
    arena_h = d_arena[: max(used, 1)].cpu().numpy().view(np.uint32)
    threads = min(16, os.cpu_count() or 1)
    chunk = min(n, FULL_CHUNK)
    assert chunk % 64 == 0 or chunk == n
    oracle_s = 0.0
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        part = d_res[lo * 4: (lo + m) * 4].cpu().numpy().view(pa.RESULT_DTYPE)
        coff, cids = pa.gather_classes(part, arena_h, host)
        h_tiles, h_lens = txome.simulate_host(read_len, seed, m, ppm, lo, wpr)
        t_lo = lo // 64 * wpr * 64
        assert np.array_equal(d_tiles[t_lo: t_lo + len(h_tiles)].cpu().numpy().view(np.uint64), h_tiles), what + ": device and host simulators disagree"
        o_res, o_coff, o_ids, _ = oracle.map_tiles(h_tiles, h_lens, wpr, 2, threads)
        oracle_s += oracle.last_seconds
        helpers.assert_same_as_oracle(part, coff, cids, o_res, o_coff, o_ids, "%s reads [%d, %d)" % (what, lo, lo + m))
        del part, coff, cids, h_tiles, h_lens, o_res, o_coff, o_ids
    print("%s: all %d reads bit-exact vs the oracle (%.1f s of oracle time on %d threads)" % (what, n, oracle_s, threads))
    # idempotence: a second launch over the same tiles gives the same records (arena placement aside) and doubles the table
    del d_res, d_arena
    d_res2, d_arena2, used2, _ = _map_device(aligner, d_tiles, d_lens, n, wpr, 2, False)
    res2 = d_res2.view(n, 4)
    mm2 = res2[:, 1] & 0x7FFFFFFF
    assert (int(res2[:, 0].sum().item()), int(mm2.sum().item()), int(res2[:, 3].sum().item()), int((res2[:, 1] < 0).sum().item())) == checksum
    return counts


def test_config3_full_batch_properties(txome, big):
    """the 100 M x 150 bp batch BASELINE.json quotes for config 3 (one launch)"""
    host, aligner, oracle = big(24)
    _full_batch_properties(txome, host, aligner, oracle, 150, 2, 0, FULL_BATCH, "config 3 full batch")


def test_config5_two_million_reads_bit_exact(txome, big):
    """BASELINE.json configs[4]: K=31, 1 % substitutions, seed 4: re-seek after a dead end / a third mismatch
    (src/pseudoaligner.rs:287-299), left extension incl. the offset-0 quirk (:124-205, :129), empty and novel classes"""
    host, aligner, oracle = big(31)
    ctr, want = _bit_exact_with_counts(txome, host, aligner, oracle, 31, 150, 4, 10000, 2_000_000, 0, "config 5, 2 M reads")
    assert ctr["reseeks"] > 50_000 and ctr["left_extensions"] > 50_000                # the paths this config exists for
    assert want[-3] > 0 and want[-2] > 0                                              # novel and empty classes do occur


def test_config5_full_batch_properties(txome, big):
    host, aligner, oracle = big(31)
    _full_batch_properties(txome, host, aligner, oracle, 150, 4, 10000, FULL_BATCH, "config 5 full batch")


@pytest.mark.parametrize("k", [31, 64])
def test_golden_synthetic_error_reads_through_hip(small_index, k):
    """tests/golden/synth_err_k{31,64}.tsv (error reads on gencode_small, generated by tests/golden/make_golden.py from the
    pinned oracle) against the HIP path"""
    lines = (helpers.GOLDEN / ("synth_err_k%d.tsv" % k)).read_text().splitlines()
    reads = [l.split("\t")[0] for l in lines]
    a = pa.Pseudoaligner(small_index(k), 0)
    res, coff, cids = a.map_batch(reads, 2)
    got = helpers.result_lines(reads, res["mismatches"] >> 31, res["coverage"], res["mismatches"] & 0x7FFFFFFF, coff, cids)
    assert [g.rstrip("\n") for g in got] == lines


def test_config3r_repeat_families_two_million_reads_bit_exact():
    """bench.py's "config3r" (VERDICT r5 item 4): the config-3 transcriptome with 50 repeat families in the last exons of 13 % of the genes
    and 200 low-complexity tracts — k-mers shared by tens to hundreds of transcripts of unrelated genes (classes far beyond two 32-id
    windows: the list tiers of map_pool.hip at scale), branch-dense unitigs, self-loops. 2 M error-free reads and 1 M reads with 1 %
    substitutions bit-exact incl. the count table (src/pseudoaligner.rs:323-356, :389-418)"""
    if pa.lib().pa_device_count() < 1:
        raise RuntimeError("the gpu tier needs a GPU and the HIP library: %s" % pa.lib().pa_last_error().decode())
    tx = pa.Txome.synthesize_repeats(GENES, TRANSCRIPTS, TX_SEED)
    plain_bases = int(pa.Txome.synthesize(GENES, TRANSCRIPTS, TX_SEED).arrays()[1][-1])
    assert tx.num_transcripts > 190_000 and int(tx.arrays()[1][-1]) > plain_bases + 3_000_000       # the same genes, longer last exons
    host = pa.HostIndex.from_txome_device(tx, 24, 0)
    a = host.arrays()
    clen = (a["ec_offset"][1:] - a["ec_offset"][:-1]).astype(np.int64)
    assert int(clen.max()) > 150 and int((clen > 32).sum()) > 500                                   # hub classes exist
    aligner, oracle = pa.Pseudoaligner(host, 0), helpers.Oracle(host)
    ctr, want = _bit_exact_with_counts(tx, host, aligner, oracle, 24, 150, 2, 0, 2_000_000, 0, "config3r, 2 M reads")
    assert ctr["mapped"] == 2_000_000 and ctr["class_sizes"] / ctr["reads"] > 12                   # (config 3: 10.7 ids per read over the visited classes)
    ctr, _ = _bit_exact_with_counts(tx, host, aligner, oracle, 24, 150, 5, 10000, 1_000_000, 0, "config3r, 1 M reads with 1 % substitutions")
    assert ctr["reseeks"] > 20_000


def test_count_table_beyond_max_bins_full_batch(tmp_path):
    """count_sort.hip's plain-atomics path (tables beyond MAX_BINS bins) on a LARGE batch: in the product only a table of more than 8.4 M classes
    gets there with more than 65 536 reads; the test build _build.build_maxbins_variant() (count_sort.hip with MAX_BINS = 2, everything else the
    product's objects) sends a 100 k-class table down that path. A subprocess loads it through PA_PRODUCT_SO: 1.5 M reads with 1 % errors, the
    table against the histogram of the oracle's results."""
    import subprocess, sys
    _b = helpers._build
    so = _b.MAXBINS_SO
    if not so.exists():
        so = _b.build_maxbins_variant()
    code = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch, numpy as np
import helpers
pa = helpers.pa
tx = pa.Txome.synthesize(12000, 42000, 7)
host = pa.HostIndex.from_txome_device(tx, 24, 0)
a = pa.Pseudoaligner(host, 0)
assert a.counts_len() > 3 * 32768, a.counts_len()
n, wpr = 1500000, 5
dev = torch.device("cuda", 0)
d_tiles = torch.empty(pa.lib().pa_tiles_words(n, wpr), dtype=torch.int64, device=dev)
d_lens = torch.empty(n, dtype=torch.int32, device=dev)
tx.simulate_device(150, 3, n, d_tiles.data_ptr(), d_lens.data_ptr(), 10000, 0, wpr)
h_tiles, h_lens = tx.simulate_host(150, 3, n, 10000, 0, wpr)
cap = a.arena_hint(n)
d_res = torch.empty(n * 4, dtype=torch.int32, device=dev); d_arena = torch.empty(cap, dtype=torch.int32, device=dev)
d_counts = torch.zeros(a.counts_len(), dtype=torch.int64, device=dev)
a.map_count_batch_device(d_tiles.data_ptr(), d_lens.data_ptr(), n, wpr, d_res.data_ptr(), d_arena.data_ptr(), cap, d_counts.data_ptr(), 2)
a.map_finish()
o = helpers.Oracle(host).map_tiles(h_tiles, h_lens, wpr, 2, 16)
want = helpers.counts_reference_fast(o[0], o[1], o[2], host)
got = d_counts.cpu().numpy()
assert int(got.sum()) == n and np.array_equal(got, want), "count table differs"
print("OK", a.counts_len())
''' % (str(helpers.ROOT), str(helpers.ROOT / "tests"))
    env = dict(os.environ, PA_PRODUCT_SO=str(so))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
