"""`-m gpu` tier, VOLUME: what the builder's soak tools run by the thousand (tools/gpu_soak.py, gpu_long_chain_soak.py,
gpu_many_classes.py), a slice of it in the driver-run tier — the pool scheduler and the list-mode tiers of map_pool.hip exist on the GPU
only, so the emulator tier cannot stand in for them. Families: helpers.random_txome_case (400 seeds), long_chain_case (60), many_classes_case
(24), branch_case (branch points on block seams, 96), tandem_case (k-mer cycles, reads of exactly K bases, 96), and the same transcriptomes
handed over as foreign flat indexes. Every read bit-exact against the oracle."""
import numpy as np
import pytest

import helpers

pa = helpers.pa
pytestmark = pytest.mark.gpu


def gpu_vs_oracle(host, reads, allowed, what):
    if pa.lib().pa_device_count() < 1:
        raise RuntimeError("the gpu tier needs a GPU and the HIP library: %s" % pa.lib().pa_last_error().decode())
    a = pa.Pseudoaligner(host, 0)
    res, coff, cids = a.map_batch(reads, allowed)
    want = helpers.Oracle(host).map_reads(reads, allowed, 8)
    helpers.assert_same_as_oracle(res, coff, cids, want[0], want[1], want[2], what)
    return want[3]


@pytest.mark.parametrize("block", range(10))
def test_random_transcriptomes_four_hundred_seeds(tmp_path, block):
    for seed in range(1000 + 40 * block, 1040 + 40 * block):
        host, k, reads, clean, allowed = helpers.random_txome_case(seed, tmp_path, big=(seed % 5 == 0))
        if host is None:
            continue
        gpu_vs_oracle(host, reads, allowed, "random txome seed %d k=%d" % (seed, k))


@pytest.mark.parametrize("block", range(4))
def test_long_chains_sixty_seeds(tmp_path, block):
    for seed in range(500 + 15 * block, 515 + 15 * block):
        host, reads, allowed = helpers.long_chain_case(seed, tmp_path)
        ctr = gpu_vs_oracle(host, reads, allowed, "long chain seed %d" % seed)
        assert ctr["left_extensions"] > 0


@pytest.mark.parametrize("seed", range(24))
def test_many_classes_twenty_four_seeds(tmp_path, seed):
    host, reads = helpers.many_classes_case(100 + seed, tmp_path, nreads=120, ordered=bool(seed % 2))
    gpu_vs_oracle(host, reads, 2, "many classes seed %d" % seed)


@pytest.mark.parametrize("block", range(4))
def test_branch_points_on_block_seams(tmp_path, block):
    """branch records / favoured-branch tails / bubbles on and beside multiples of 64, clustered and dense-head errors, allowed to 12"""
    for seed in range(24 * block, 24 * block + 24):
        host, reads, allowed = helpers.branch_case(seed, tmp_path)
        if host is not None:
            gpu_vs_oracle(host, reads, allowed, "branch case seed %d" % seed)
            if seed % 3 == 0:   # ... and handed over in a foreign layout (cut unitigs: more branch-free nodes ending on seams)
                foreign, _ = helpers.foreign_index(host, 900 + seed, cut_frac=0.6)
                gpu_vs_oracle(foreign, reads, allowed, "branch case seed %d, foreign layout" % seed)


@pytest.mark.parametrize("block", range(4))
def test_tandem_repeats_and_reads_of_exactly_k(tmp_path, block):
    for seed in range(24 * block, 24 * block + 24):
        host, reads, allowed = helpers.tandem_case(seed, tmp_path)
        if host is not None:
            gpu_vs_oracle(host, reads, allowed, "tandem case seed %d" % seed)
