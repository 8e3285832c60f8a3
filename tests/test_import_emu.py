"""CPU tier of the index hand-over: the product's flattener + lane steps (tests/emu) over FOREIGN flat indexes —
nodes permuted, classes renumbered, unitigs cut into same-colour pieces, the reference's two-pass / first-pass node sets —
against the oracle built on the SAME arrays. The GPU tier's twin is tests/test_gpu_import.py."""
import numpy as np
import pytest

import helpers
import import_cases as ic

pa = helpers.pa


def emu_mapper():
    cache = {}

    def run(foreign, tiles, lens, wpr, allowed):
        if id(foreign) not in cache:
            cache[id(foreign)] = helpers.Emu(foreign)
        r = cache[id(foreign)].map_tiles(tiles, lens, wpr, allowed, 8)
        return r["results"], r["coff"], r["ids"]
    return run


@pytest.mark.parametrize("k,seed", [(20, 1), (31, 2), (64, 3)])
def test_permuted_renumbered_cut_index(small_index, k, seed):
    own = small_index(k) if k != 64 else pa.build_index(str(helpers.FASTA), 64, 8)
    foreign, ncut = helpers.foreign_index(own, seed)
    assert ncut > 1000 and foreign.arrays()["num_nodes"] > own.arrays()["num_nodes"] + 1000
    differ = ic.check_foreign(own, foreign, emu_mapper(), what="emu foreign K=%d" % k)
    assert differ > 0, "break points changed no result: the test is vacuous"
    tiles, lens, wpr = ic.small_fq_tiles()
    want = helpers.Oracle(foreign).map_tiles(tiles, lens, wpr, 2, 4)
    got = emu_mapper()(foreign, tiles, lens, wpr, 2)
    helpers.assert_same_as_oracle(*got, want[0], want[1], want[2], "small.fq through a foreign index")


def test_pass_one_node_set_of_the_reference_build(small_index):
    """the unitigs as the reference's assemble_shard leaves them (paths cut at MSP shard seams, build_index.rs:153-172)"""
    _, seqs = helpers.read_fasta()
    seqs = [s.upper() for s in seqs]
    own = small_index(24)
    foreign = ic.pass_one_index(seqs, 24, len(seqs))
    assert foreign.arrays()["num_nodes"] > own.arrays()["num_nodes"]
    differ = ic.check_foreign(own, foreign, emu_mapper(), alloweds=(0, 2), what="emu pass-one K=24")
    assert differ > 0


def test_two_pass_node_set_is_the_product_builders(small_index):
    _, seqs = helpers.read_fasta()
    seqs = [s.upper() for s in seqs]
    own = small_index(24)
    foreign = ic.two_pass_index(seqs, 24, len(seqs))
    rc, why = own.compare(foreign)
    assert rc == 0 and why.startswith("identical"), why
    assert ic.check_foreign(own, foreign, emu_mapper(), alloweds=(2,), what="emu two-pass K=24") == 0


@pytest.mark.parametrize("seed", range(12))
def test_random_transcriptomes_through_foreign_indexes(tmp_path, seed):
    """the differential-fuzz transcriptomes (repeats, two-letter alphabets, k 8..64) handed over in a foreign layout"""
    host, k, reads, clean, allowed = helpers.random_txome_case(seed, tmp_path)
    if host is None:
        pytest.skip("every transcript is shorter than k")
    foreign, _ = helpers.foreign_index(host, 50 + seed, cut_frac=0.7)
    tiles, lens, wpr = pa.encode_reads_host(reads)
    want = helpers.Oracle(foreign).map_tiles(tiles, lens, wpr, allowed, 4)
    got = emu_mapper()(foreign, tiles, lens, wpr, allowed)
    helpers.assert_same_as_oracle(*got, want[0], want[1], want[2], "fuzz seed %d foreign" % seed)
