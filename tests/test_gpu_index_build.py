"""`-m gpu`: index construction on the GPU (csrc/index_build.hip) against the CPU builder (csrc/dbg_build.cpp):
the two must give the SAME flat index — node sequences, order, lengths, extension bits, colours and class lists, array for
array — on the reference's own transcriptome (test/gencode_small.fa) at one- and two-word k, on random small transcriptomes
with repeats and pure cycles, and at BASELINE.json's scale (the ~202 k-transcript synthetic transcriptome of configs 3-5).
The CPU builder itself is pinned by tests/test_index_build.py (naive Python graph) and tests/test_oracle_golden.py
(validate_dbg, src/build_index.rs:262-368)."""
import os
import time

import numpy as np
import pytest

import helpers
from test_index_build import naive_graph, graph_of, pack, random_txome

pa = helpers.pa
pytestmark = pytest.mark.gpu

KEYS = ("node_seq", "node_start", "node_len", "node_exts", "node_colour", "ec_offset", "ec_ids")


def assert_same_index(gpu, cpu, what):
    a, b = gpu.arrays(), cpu.arrays()
    for key in ("k", "num_nodes", "num_classes", "num_transcripts"):
        assert a[key] == b[key], "%s: %s %r != %r" % (what, key, a[key], b[key])
    for key in KEYS:
        assert np.array_equal(a[key], b[key]), "%s: %s differs" % (what, key)


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if pa.lib().pa_device_count() < 1:
        raise RuntimeError("the gpu tier needs a GPU and the HIP library: %s" % pa.lib().pa_last_error().decode())


@pytest.mark.parametrize("k", [20, 24, 31, 32, 33, 48, 64])
def test_gencode_small_same_index_as_cpu_builder(k):
    gpu = pa.HostIndex.build_fasta_device(str(helpers.FASTA), k, 0)
    cpu = pa.HostIndex.build_fasta(str(helpers.FASTA), k, 0)
    assert_same_index(gpu, cpu, "gencode_small K=%d" % k)
    assert gpu.tx_names() == cpu.tx_names() and gpu.tx_genes() == cpu.tx_genes()
    if k == 24:   # 1 165 762 distinct 24-mers
        a = gpu.arrays()
        assert int((a["node_len"] - 23).sum()) == 1165762


@pytest.mark.parametrize("seed", range(24))
def test_random_transcriptomes_same_index_and_naive_graph(seed):
    rng = np.random.RandomState(seed)
    k = int(rng.choice([8, 9, 12, 16, 31, 32, 33, 48, 64]))
    seqs = random_txome(rng, rng.randint(2, 25), "ACGT" if seed % 3 else "AC")   # two-letter alphabet: repeats and pure cycles
    if seed % 5 == 0:
        seqs += ["AC" * 40, "CA" * 37, "A" * 70]                                   # a 2-cycle of joinable k-mers, a self-loop
    words, tx_start = pack(seqs)
    gpu = pa.HostIndex.build_packed_device(words, tx_start, k, 0)
    cpu = pa.HostIndex.build_packed(words, tx_start, k, 1 + seed % 4)
    assert_same_index(gpu, cpu, "seed %d K=%d" % (seed, k))
    _, got = graph_of(gpu)
    want, cyc = naive_graph(seqs, k)
    got_set = set(got)
    cyc_nodes = {g for g in got_set if g[0][:k] in cyc}
    assert got_set - cyc_nodes == want
    assert {g[0][p:p + k] for g in cyc_nodes for p in range(len(g[0]) - k + 1)} == cyc


def test_degenerate_inputs():
    # nothing but transcripts shorter than k (src/build_index.rs:134,148-150), and an empty transcript in the middle
    words, tx_start = pack(["ACGTACG", "", "ACG"])
    gpu = pa.HostIndex.build_packed_device(words, tx_start, 8, 0)
    cpu = pa.HostIndex.build_packed(words, tx_start, 8, 1)
    assert_same_index(gpu, cpu, "no k-mers")
    words, tx_start = pack(["ACGTACG", "", "ACGTACGTTGCAAGGCT", "", "TTGCAAGGCTA"])
    gpu = pa.HostIndex.build_packed_device(words, tx_start, 8, 0)
    cpu = pa.HostIndex.build_packed(words, tx_start, 8, 2)
    assert_same_index(gpu, cpu, "short + empty transcripts")
    with pytest.raises(pa.PaError):
        pa.HostIndex.build_packed_device(words, tx_start, 7, 0)        # k below PA_MIN_K
    with pytest.raises(pa.PaError):
        pa.HostIndex.build_packed_device(words, tx_start, 8, 99)       # no such device


@pytest.mark.parametrize("k", [24, 31])
def test_gencode_scale_same_index_as_cpu_builder(k):
    """configs 3 (K=24) and 5 (K=31): 201 914 transcripts, ~3.1e8 k-mer occurrences, ~1.04e8 distinct k-mers"""
    txome = pa.Txome.synthesize(58000, 203000, 7)
    t0 = time.time()
    gpu = pa.HostIndex.from_txome_device(txome, k, 0)
    t_gpu = time.time() - t0
    t0 = time.time()
    cpu = pa.HostIndex.from_txome(txome, k, 0)
    t_cpu = time.time() - t0
    print("index build K=%d: GPU %.2f s, CPU (%d threads) %.2f s" % (k, t_gpu, os.cpu_count() or 1, t_cpu))
    assert_same_index(gpu, cpu, "GENCODE scale K=%d" % k)


def test_mapping_through_a_gpu_built_index():
    """small.fq through an index that never saw the CPU builder: same results as the oracle on the CPU-built index"""
    gpu = pa.HostIndex.build_fasta_device(str(helpers.FASTA), 20, 0)
    aligner = pa.Pseudoaligner(gpu, 0)
    ids, seqs = helpers.read_fastq()
    res, coff, cids = aligner.map_batch(seqs)
    cpu = pa.HostIndex.build_fasta(str(helpers.FASTA), 20, 0)
    o_res, o_coff, o_ids, _ = helpers.Oracle(cpu).map_reads(seqs, 2, 4)
    helpers.assert_same_as_oracle(res, coff, cids, o_res, o_coff, o_ids, "small.fq on a GPU-built index")
