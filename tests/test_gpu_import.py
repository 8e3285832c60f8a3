"""`-m gpu` tier of the index hand-over: the HIP path maps through FOREIGN flat indexes —
arrays the product's builders did not lay out: nodes permuted, classes renumbered, about half of the unitigs cut at random
k-mer boundaries into same-colour pieces, and the node sets of the reference's two-pass build order
(/root/reference/src/build_index.rs:127-179) — pa_flat_index -> pa_host_index_from_flat -> pa_index_create -> mapping, bit-exact
against the oracle built on the SAME arrays (/root/reference/src/pseudoaligner.rs:26-33 is what the exporter reads)."""
import numpy as np
import pytest

import helpers
import import_cases as ic

pa = helpers.pa
pytestmark = pytest.mark.gpu


def gpu_mapper():
    import torch
    cache = {}

    def run(foreign, tiles, lens, wpr, allowed):
        if pa.lib().pa_device_count() < 1:
            raise RuntimeError("the gpu tier needs a GPU and the HIP library: %s" % pa.lib().pa_last_error().decode())
        if id(foreign) not in cache:
            cache[id(foreign)] = pa.Pseudoaligner(foreign, 0)
        a = cache[id(foreign)]
        n = len(lens)
        dev = torch.device("cuda", 0)
        d_tiles = torch.from_numpy(tiles.view(np.int64)).to(dev)
        d_lens = torch.from_numpy(np.ascontiguousarray(lens).view(np.int32)).to(dev)
        cap = a.arena_hint(n)
        d_res = torch.zeros(n * 4, dtype=torch.int32, device=dev)
        d_arena = torch.zeros(cap, dtype=torch.int32, device=dev)
        a.map_batch_device(d_tiles.data_ptr(), d_lens.data_ptr(), n, wpr, d_res.data_ptr(), d_arena.data_ptr(), cap, allowed)
        used, need = a.map_finish()
        assert need <= cap
        res = d_res.cpu().numpy().view(pa.RESULT_DTYPE)
        coff, ids = pa.gather_classes(res, d_arena.cpu().numpy().view(np.uint32)[:used], foreign)
        return res, coff, ids
    return run


@pytest.mark.parametrize("k,seed", [(20, 1), (31, 2), (64, 3)])
def test_permuted_renumbered_cut_index(small_index, k, seed):
    own = small_index(k) if k != 64 else pa.build_index(str(helpers.FASTA), 64, 8)
    foreign, ncut = helpers.foreign_index(own, seed)
    assert ncut > 1000 and foreign.arrays()["num_nodes"] > own.arrays()["num_nodes"] + 1000
    mapper = gpu_mapper()
    differ = ic.check_foreign(own, foreign, mapper, what="gpu foreign K=%d" % k)
    assert differ > 0, "break points changed no result: the test is vacuous"
    tiles, lens, wpr = ic.small_fq_tiles()
    want = helpers.Oracle(foreign).map_tiles(tiles, lens, wpr, 2, 4)
    helpers.assert_same_as_oracle(*mapper(foreign, tiles, lens, wpr, 2), want[0], want[1], want[2], "small.fq through a foreign index")


def test_foreign_index_through_the_ascii_batch_entry_and_map_read(small_index):
    """the boundary a day-one user calls: pa_map_batch / pa_map_read on an imported index"""
    own = small_index(20)
    foreign, _ = helpers.foreign_index(own, 9)
    a = pa.Pseudoaligner(foreign, 0)
    _, seqs = helpers.read_fastq()
    res, coff, cids = a.map_batch(seqs, 2)
    want = helpers.Oracle(foreign).map_reads(seqs, 2, 8)
    helpers.assert_same_as_oracle(res, coff, cids, want[0], want[1], want[2], "pa_map_batch foreign")
    ex1 = "GGCTGTCAACCAGTCCATAGGCAGGGCCATCAGGCACCAAAGGGATTCTGCCAGCATAGT"          # src/build_index.rs:429-434
    assert a.map_read(ex1) == ([1, 30], len(ex1))
    for i in (0, 5, 77, 885):
        rc, ids, cov, mm, nodes = helpers.Oracle(foreign).map_read(seqs[i])
        assert a.map_read_with_mismatch(seqs[i], 2) == ((ids, cov, mm) if rc else None)
        got = a.map_read_to_nodes(seqs[i])
        assert got == ((nodes, cov) if rc else None)                            # node ids are the FOREIGN index's


def test_pass_one_and_two_pass_node_sets_of_the_reference_build(small_index):
    _, seqs = helpers.read_fasta()
    seqs = [s.upper() for s in seqs]
    own = small_index(24)
    mapper = gpu_mapper()
    p1 = ic.pass_one_index(seqs, 24, len(seqs))
    assert p1.arrays()["num_nodes"] > own.arrays()["num_nodes"]
    assert ic.check_foreign(own, p1, mapper, alloweds=(0, 2), what="gpu pass-one K=24") > 0
    p2 = ic.two_pass_index(seqs, 24, len(seqs))
    assert own.compare(p2)[1].startswith("identical")
    assert ic.check_foreign(own, p2, mapper, alloweds=(2,), what="gpu two-pass K=24") == 0


def test_synthetic_slice_foreign(tmp_path):
    """a 3000-gene slice of the bench's synthetic transcriptome at K = 24 (the index family of configs 3-5), handed over cut and permuted;
    count table of the fused launch included (class numbering is the foreign one)"""
    import torch
    tx = pa.Txome.synthesize(3000, 10500, 7)
    own = pa.HostIndex.from_txome(tx, 24, 8)
    foreign, ncut = helpers.foreign_index(own, 4)
    assert ncut > 5000
    mapper = gpu_mapper()
    assert ic.check_foreign(own, foreign, mapper, what="gpu foreign synth K=24") > 0
    a = pa.Pseudoaligner(foreign, 0)
    tiles, lens, wpr = helpers.error_reads(own, 150, 200000, 10000, 5)
    dev = torch.device("cuda", 0)
    n = len(lens)
    d_tiles = torch.from_numpy(tiles.view(np.int64)).to(dev)
    d_lens = torch.from_numpy(lens.view(np.int32)).to(dev)
    cap = a.arena_hint(n)
    d_res = torch.zeros(n * 4, dtype=torch.int32, device=dev)
    d_arena = torch.zeros(cap, dtype=torch.int32, device=dev)
    d_counts = torch.zeros(a.counts_len(), dtype=torch.int64, device=dev)
    a.map_count_batch_device(d_tiles.data_ptr(), d_lens.data_ptr(), n, wpr, d_res.data_ptr(), d_arena.data_ptr(), cap, d_counts.data_ptr(), 2)
    a.map_finish()
    want = helpers.Oracle(foreign).map_tiles(tiles, lens, wpr, 2, 8)
    assert np.array_equal(d_counts.cpu().numpy().astype(np.uint64), helpers.counts_reference_fast(want[0], want[1], want[2], foreign).astype(np.uint64))


@pytest.mark.parametrize("seed", range(12))
def test_random_transcriptomes_through_foreign_indexes(tmp_path, seed):
    host, k, reads, clean, allowed = helpers.random_txome_case(seed, tmp_path)
    if host is None:
        pytest.skip("every transcript is shorter than k")
    foreign, _ = helpers.foreign_index(host, 50 + seed, cut_frac=0.7)
    a = pa.Pseudoaligner(foreign, 0)
    res, coff, cids = a.map_batch(reads, allowed)
    want = helpers.Oracle(foreign).map_reads(reads, allowed, 4)
    helpers.assert_same_as_oracle(res, coff, cids, want[0], want[1], want[2], "fuzz seed %d foreign" % seed)
