"""The C oracle against an independent Python model of the same spec (tests/pymodel.py, written from SURVEY.md §3.2): thousands
of config-5-type reads (150 bp, substitutions) at K = 31 and K = 64 on a slice of the bench's synthetic transcriptome, and reads
built to hit the left extension with kmer_offset == 0 (the quirk of src/pseudoaligner.rs:129), re-seeks and premature breaks.
Both sides get the same flat index; class ids, coverage, mismatches and the node list in visit order must agree read for read.
The event counters of the model make sure the comparison is not vacuous."""
import numpy as np
import pytest

import helpers
from pymodel import Model

pa = helpers.pa


def _strings(tiles, lens, wpr):
    n = len(lens)
    out = []
    lut = np.frombuffer(b"ACGT", np.uint8)
    t = tiles.reshape(-1, wpr, 64)
    for i in range(n):
        words = t[i >> 6, :, i & 63]
        pos = np.arange(int(lens[i]), dtype=np.int64)
        codes = ((words[pos >> 5] >> ((pos & 31) * 2).astype(np.uint64)) & np.uint64(3)).astype(np.uint8)
        out.append(lut[codes].tobytes().decode())
    return out


@pytest.fixture(scope="module")
def slice_txome():
    return pa.Txome.synthesize(260, 900, 11)


def _compare(host, reads, allowed, model=None):
    model = model or Model(host.arrays())
    oracle = helpers.Oracle(host)
    for i, r in enumerate(reads):
        m = model.map_read(r, allowed)
        rc, cls, cov, mm, nodes = oracle.map_read(r, allowed)
        if m is None:
            assert rc == 0, (i, r)
        else:
            assert rc == 1 and (cls, cov, mm, nodes) == (m[0], m[1], m[2], m[3]), (i, r, (cls, cov, mm, nodes), m)
    return model


@pytest.mark.parametrize("k", [31, 64])
def test_oracle_equals_the_independent_model_on_error_reads(slice_txome, k):
    host = pa.HostIndex.from_txome(slice_txome, k, 4)
    wpr = 5
    tiles, lens = slice_txome.simulate_host(150, 4, 5120, 10000, 0, wpr)              # 1 % substitutions: config 5's reads
    reads = _strings(tiles, lens, wpr)
    model = _compare(host, reads, 2)
    tiles, lens = slice_txome.simulate_host(150, 9, 1536, 40000, 0, wpr)              # 4 %: many seeds past the first fifth of the read
    model = _compare(host, _strings(tiles, lens, wpr), 2, model)
    ev = model.events
    # (at K = 64 a 150-base read has few k-mers left to re-seek with after a premature break)
    assert ev["left_ext"] > 150 and ev["left_hops"] > 20 and ev["reseek"] > (300 if k == 31 else 10) and ev["premature_fwd"] > 200 and ev["hops"] > 1500, ev


@pytest.mark.parametrize("k,allowed", [(20, 0), (20, 2), (31, 1), (31, 3)])
def test_constructed_left_extension_quirk_reseek_and_premature_break(slice_txome, k, allowed):
    """Reads assembled around node boundaries: the seed is the FIRST k-mer of a node (kmer_offset == 0: :129 clamps
    prev_kmer_offset to 0 and compares the k-mer's own first base) preceded by 0..3 damaged copies of what lies to the left of it,
    long enough for the seed to sit past a fifth of the read; and reads with a burst of errors in the middle (premature break,
    then a re-seek that finds the rest)."""
    host = pa.HostIndex.from_txome(slice_txome, k, 4)
    model = Model(host.arrays())
    rng = np.random.RandomState(17 + k + allowed)
    seqs = model.seq
    reads = []
    starts = [n for n in range(len(seqs)) if (model.exts[n] >> 4) and len(seqs[n]) >= k]
    for n in rng.choice(starts, 400):
        # walk left from node n over random left edges to collect 30-60 bases of true context
        left, cur = "", int(n)
        while len(left) < 60 and (model.exts[cur] >> 4):
            b = [c for c in "ACGT" if model.has_left(cur, c)][rng.randint(0, bin(model.exts[cur] >> 4).count("1"))]
            cur = model.l_edge(cur, b)
            left = seqs[cur][:len(seqs[cur]) - (k - 1)][-(60 - len(left)):] + left
        # the scan probes every third position (:110): the seed has to start at a multiple of 3 to be met at kmer_offset 0
        left = list(left[-3 * rng.randint(9, 20):])
        if len(left) % 3:
            left = left[len(left) % 3:]
        for p in range(len(left) - 1, -1, -k):                   # one damaged base in every window of k bases: no k-mer left of the seed hits
            left[p] = "ACGT"[("ACGT".index(left[p]) + rng.randint(1, 4)) % 4]
        body = seqs[n][:rng.randint(k, max(k + 1, min(len(seqs[n]), 90)))]
        reads.append("".join(left) + body)
    long_nodes = [n for n in range(len(seqs)) if len(seqs[n]) >= 3 * k + 20]
    for n in rng.choice(long_nodes, 400):
        s = list(seqs[n][:min(len(seqs[n]), 150)])
        mid = rng.randint(k + 2, len(s) - k - 8)
        for p in range(mid, mid + allowed + 1 + rng.randint(0, 2)):   # more consecutive errors than a node visit tolerates
            s[p] = "ACGT"[("ACGT".index(s[p]) + 1 + rng.randint(0, 3)) % 4]
        reads.append("".join(s))
    model = _compare(host, reads, allowed, model)
    ev = model.events
    assert ev["q1"] > 100 and ev["left_ext"] > 150 and ev["premature_fwd"] > 150 and ev["reseek"] > 150, ev
