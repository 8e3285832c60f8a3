"""Unitig break points: an emulation of the REFERENCE'S build order against the product's builder.

The reference does not compress the De Bruijn graph in one pass. `build_index` (src/build_index.rs:27-91) cuts every transcript
into minimum-substring-partition pieces (`partition_contigs`, :127-151, p = 6, permutation `PERM` :95-113), sorts the pieces by
bucket, groups whole buckets into shards of more than 2000 pieces (`group_by_slices`, :227-244), assembles every shard on its
own (`assemble_shard`, :153-172: `filter_kmers` + `compress_kmers_with_hash` with `ScmapCompress`), concatenates the shard
graphs and compresses AGAIN (`merge_shard_dbgs`, :174-179: `compress_graph`). The product's builders (csrc/dbg_build.cpp on
the CPU, csrc/index_build.hip on the GPU) compute the maximal unique-extension same-colour paths of the whole k-mer set in one
pass. Results of reads WITH errors depend on where unitigs end (a node visit resets the mismatch budget and pushes a class), so
the two orders must give the same node set. This test performs the reference's order step by step — test infrastructure only,
numpy, no product code — and compares the nodes (sequence, extension bits, transcript-id list) with the product's index on
test/gencode_small.fa at K = 20, 24, 31.

Assumptions about the `debruijn` crate (0.3.4 @ 8d9a5c52; its source is NOT on disk — every line below that leans on it says so):
  A1  `msp::simple_scan(k, seq, perm, rc = false)`: every k-mer is assigned its minimum p-mer under `perm` (leftmost on ties);
      consecutive k-mers with the same minimum p-mer POSITION form one piece; `bucket()` is the p-mer's value. The property under
      test only needs "all occurrences of a k-mer land in one bucket", which holds for any function of the k-mer alone.
  A2  `PERM` (:95-113) sorts the 4096 6-mers by `count_a_t_bases`, which compares `kmer.get(i)` with b'A' / b'T'. `Mer::get`
      returns the 2-bit code (0..3), so the count is always 0 and the stable sort leaves the identity. Both the identity and the
      ordering the authors meant (by A/T count) are emulated: the node set must not depend on it.
  A3  `Exts::from_dna_string(contig, start, len)`: the bases before / after the piece inside the contig. Together with the
      neighbours inside a piece this gives every k-mer occurrence its true neighbours in its transcript; `CountFilterEqClass`
      (src/equiv_classes.rs:62-91, MIN_KMERS = 1) ORs them over all occurrences and interns the sorted transcript-id list.
  A4  `compress_kmers_with_hash` extends a path from x to y = x[1:] + b iff b is x's ONLY right extension, y is a k-mer of the
      SAME shard, y has exactly ONE left extension (which is then x[0]), and `ScmapCompress::join_test` holds (equal data = equal
      class id); a path ends where any of the four fails. A k-mer cycle is cut where the walk started.
  A5  `compress_graph` applies the same test to NODES: A -> B is joined iff A's last k-mer has one right extension, it leads to
      the FIRST k-mer of B, B's first k-mer has one left extension, and the classes are equal.
Under A4/A5 the second pass can only join what the first pass was not allowed to see (a neighbour in another shard), so the final
nodes are the maximal paths of the one-pass rule — except on pure k-mer cycles, whose cut is arbitrary in both (listed, not compared).
"""
import numpy as np
import pytest

import helpers

pa = helpers.pa
P = 6   # PmerType = Kmer6 (:93)


def _codes(seq):
    lut = np.full(256, 255, np.uint8)
    for ch, v in (("A", 0), ("C", 1), ("G", 2), ("T", 3), ("a", 0), ("c", 1), ("g", 2), ("t", 3)):
        lut[ord(ch)] = v
    c = lut[np.frombuffer(seq.encode(), np.uint8)]
    assert (c < 4).all(), "gencode_small.fa holds no N: from_acgt_bytes_hashn (src/utils.rs:76) is not exercised"
    return c.astype(np.uint64)


def _mers(codes, k):
    """value of every k-mer of a transcript, first base most significant (the debruijn crate's integer k-mers)"""
    n = len(codes) - k + 1
    v = np.zeros(n, np.uint64)
    for j in range(k):
        v = (v << np.uint64(2)) | codes[j:j + n]
    return v


def perm_table(mode):
    if mode == "identity":   # what the code does (A2)
        return np.arange(4 ** P, dtype=np.int64)
    vals = np.arange(4 ** P)
    at = np.zeros(4 ** P, np.int64)
    for j in range(P):
        b = (vals >> (2 * j)) & 3
        at += (b == 0) | (b == 3)
    order = np.argsort(at, kind="stable")            # sort_by_key is stable
    perm = np.zeros(4 ** P, np.int64)
    perm[order] = np.arange(4 ** P)
    return perm


def reference_order_nodes(seqs, k, perm_mode, min_shard=2000, pass_one_only=False):
    """-> (set of (sequence, class tuple, left exts, right exts), number of shards, k-mers on pure cycles, nodes after pass 1);
    pass_one_only: the nodes as assemble_shard leaves them (before merge_shard_dbgs), with the extension bits of their end k-mers"""
    perm = perm_table(perm_mode)
    km_all, ext_all, tx_all, bucket_all, piece_start = [], [], [], [], []
    for t, s in enumerate(seqs):
        if len(s) < k:
            continue                                                                  # :134
        c = _codes(s)
        km = _mers(c, k)
        n = len(km)
        pm = _mers(c, P)                                                              # the p-mers of the contig
        win = np.lib.stride_tricks.sliding_window_view(perm[pm.astype(np.int64)], k - P + 1)
        mpos = np.arange(n) + win.argmin(axis=1)                                      # A1: leftmost minimum p-mer of every k-mer
        bucket = pm[mpos]
        new_piece = np.ones(n, bool)
        new_piece[1:] = mpos[1:] != mpos[:-1]
        ext = np.zeros(n, np.uint8)
        ext[:-1] |= (np.uint8(1) << c[k:].astype(np.uint8))                           # right neighbour inside the transcript (A3)
        ext[1:] |= (np.uint8(16) << c[:n - 1].astype(np.uint8))                       # left neighbour
        km_all.append(km); ext_all.append(ext); tx_all.append(np.full(n, t, np.uint32)); bucket_all.append(bucket); piece_start.append(new_piece)
    km, ext, tx = np.concatenate(km_all), np.concatenate(ext_all), np.concatenate(tx_all)
    bucket, new_piece = np.concatenate(bucket_all), np.concatenate(piece_start)

    # ---- shards: pieces sorted by bucket, whole buckets grouped while a slice holds at most min_shard pieces (:227-244) ----
    pieces_per_bucket = np.bincount(bucket[new_piece].astype(np.int64), minlength=4 ** P)
    shard_of_bucket = np.zeros(4 ** P, np.int64)
    shard, in_slice = 0, 0
    for b in np.flatnonzero(pieces_per_bucket):                                       # buckets in sorted order
        if in_slice > min_shard:                                                      # (i - slice_start) > min_size and the key changes
            shard += 1
            in_slice = 0
        shard_of_bucket[b] = shard
        in_slice += int(pieces_per_bucket[b])
    n_shards = shard + 1

    # ---- filter_kmers + CountFilterEqClass: one entry per distinct k-mer (A3) ----
    order = np.argsort(km, kind="stable")
    km, ext, tx, bucket = km[order], ext[order], tx[order], bucket[order]
    first = np.ones(len(km), bool)
    first[1:] = km[1:] != km[:-1]
    starts = np.flatnonzero(first)
    ukm = km[starts]
    uext = np.bitwise_or.reduceat(ext, starts)
    ushard = shard_of_bucket[bucket[starts].astype(np.int64)]
    assert (np.maximum.reduceat(bucket, starts) == np.minimum.reduceat(bucket, starts)).all()   # a k-mer has ONE bucket (A1)
    grp = np.cumsum(first) - 1
    pair = np.unique(np.stack([grp, tx.astype(np.int64)], axis=1), axis=0)            # distinct (k-mer, transcript)
    pstart = np.flatnonzero(np.r_[True, pair[1:, 0] != pair[:-1, 0]])
    classes, ucol = {}, np.zeros(len(ukm), np.int64)
    bounds = np.r_[pstart, len(pair)]
    for g in range(len(ukm)):                                                         # intern the sorted id lists (:84-86)
        key = pair[bounds[g]:bounds[g + 1], 1].tobytes()
        ucol[g] = classes.setdefault(key, len(classes))
    class_list = {v: tuple(np.frombuffer(key, np.int64).tolist()) for key, v in classes.items()}

    # ---- the join test of A4 / A5 on the k-mer level ----
    mask = np.uint64((1 << (2 * k)) - 1)
    r = uext & 15
    one_r = (r & (r - 1) == 0) & (r != 0)
    rb = np.where(one_r, np.log2(np.maximum(r, 1)).astype(np.uint64), np.uint64(0))
    nxt = ((ukm << np.uint64(2)) | rb) & mask
    j = np.searchsorted(ukm, nxt)
    j = np.minimum(j, len(ukm) - 1)
    exists = ukm[j] == nxt
    l = uext[j] >> 4
    one_l = (l & (l - 1) == 0) & (l != 0)
    joinable = one_r & exists & one_l & (ucol[j] == ucol) & (j != np.arange(len(ukm)))   # the global rule
    succ_global = np.where(joinable, j, -1)
    succ_shard = np.where(joinable & (ushard[j] == ushard), j, -1)                    # first pass: the neighbour must be in the shard

    def chains(succ):
        """maximal chains of the functional graph `succ`: (list of index lists, indices on pure cycles)"""
        has_pred = np.zeros(len(succ), bool)
        has_pred[succ[succ >= 0]] = True
        out, seen = [], np.zeros(len(succ), bool)
        for h in np.flatnonzero(~has_pred):
            path, cur = [], int(h)
            while cur >= 0 and not seen[cur]:
                seen[cur] = True
                path.append(cur)
                cur = int(succ[cur])
            out.append(path)
        return out, np.flatnonzero(~seen)

    # ---- pass 1: every shard by itself (assemble_shard); pass 2: compress_graph over the shard nodes (A5) ----
    nodes1, cyc1 = chains(succ_shard)
    head_of = {p[0]: i for i, p in enumerate(nodes1)}
    node_succ = np.full(len(nodes1), -1, np.int64)
    for i, p in enumerate(nodes1):
        y = int(succ_global[p[-1]])                                                   # the node's last k-mer under the node-level test
        if y >= 0 and y in head_of:
            node_succ[i] = head_of[y]
    final, cyc2 = chains(node_succ)                                                   # (a loop that pass 1 cut at shard seams closes again here)
    if pass_one_only:
        final, cyc2 = [[i] for i in range(len(nodes1))], np.zeros(0, np.int64)
    bases = "ACGT"
    out = set()
    for chain in final:
        idx = [x for n in chain for x in nodes1[n]]
        first_km = int(ukm[idx[0]])
        seq = "".join(bases[(first_km >> (2 * (k - 1 - t))) & 3] for t in range(k)) + "".join(bases[int(ukm[x]) & 3] for x in idx[1:])
        le, re = int(uext[idx[0]]) >> 4, int(uext[idx[-1]]) & 15
        out.add((seq, class_list[int(ucol[idx[0]])], "".join(bases[b] for b in range(4) if le >> b & 1), "".join(bases[b] for b in range(4) if re >> b & 1)))
    cycle_kmers = len(cyc1) + sum(len(nodes1[n]) for n in cyc2)                       # k-mers on pure cycles: in no node of `out`
    return out, n_shards, cycle_kmers, len(nodes1)


def product_nodes(host):
    a = host.arrays()
    out = set()
    bases = "ACGT"
    for n in range(a["num_nodes"]):
        s, l = int(a["node_start"][n]), int(a["node_len"][n])
        pos = np.arange(s, s + l)
        codes = (a["node_seq"][pos >> 5] >> ((pos & 31) * 2).astype(np.uint64)) & np.uint64(3)
        c = int(a["node_colour"][n])
        e = int(a["node_exts"][n])
        out.add(("".join(bases[int(x)] for x in codes), tuple(a["ec_ids"][int(a["ec_offset"][c]):int(a["ec_offset"][c + 1])].tolist()),
                 "".join(bases[b] for b in range(4) if e >> (4 + b) & 1), "".join(bases[b] for b in range(4) if e >> b & 1)))
    return out


@pytest.mark.parametrize("k,perm_mode", [(20, "identity"), (24, "identity"), (31, "identity"), (24, "at_count")])
def test_two_pass_sharded_compression_gives_the_product_builders_nodes(small_index, k, perm_mode):
    _, seqs = helpers.read_fasta()
    seqs = [s.upper() for s in seqs]
    want, n_shards, cyc, n_pass1 = reference_order_nodes(seqs, k, perm_mode)
    got = product_nodes(small_index(k))
    assert n_shards > 20 and n_pass1 > len(want)                                      # the sharding really cuts paths: pass 2 has work to do
    assert cyc == 0                                                                   # no pure k-mer cycle in this transcriptome: nothing excepted
    assert got == want, "%d nodes only in the product, %d only in the reference's order" % (len(got - want), len(want - got))


def test_shard_seams_do_cut_paths_in_the_first_pass():
    """the emulation is not vacuous: with shards, pass 1 alone leaves MORE nodes than the final graph (paths end at shard seams
    and are only healed by the second compression)"""
    _, seqs = helpers.read_fasta()
    seqs = [s.upper() for s in seqs[:300]]
    full, n_shards, _, pass1 = reference_order_nodes(seqs, 24, "identity", min_shard=200)
    single, one, _, pass1_single = reference_order_nodes(seqs, 24, "identity", min_shard=10 ** 9)
    assert n_shards > 5 and one == 1 and full == single
    assert pass1 > len(full) and pass1_single == len(single)


def _txome_strings(txome):
    packed, tx_start = txome.arrays()
    out = []
    for t in range(len(tx_start) - 1):
        pos = np.arange(int(tx_start[t]), int(tx_start[t + 1]), dtype=np.int64)
        codes = ((packed[pos >> 5] >> ((pos & 31) * 2).astype(np.uint64)) & np.uint64(3)).astype(np.uint8)
        out.append(np.frombuffer(b"ACGT", np.uint8)[codes].tobytes().decode())
    return out


@pytest.mark.parametrize("k", [24, 31])
def test_two_pass_order_on_the_synthetic_transcriptome_of_the_bench(k):
    """the index type bench.py times (csrc/synth.cpp: genes of exons, alternative transcripts, 5 % paralog copies), a 3000-gene
    slice of it: the reference's two-pass sharded order and the product's builder give the same nodes"""
    tx = pa.Txome.synthesize(3000, 10500, 7)
    seqs = _txome_strings(tx)
    assert len(seqs) >= 9000
    want, n_shards, cyc, n_pass1 = reference_order_nodes(seqs, k, "identity")
    got = product_nodes(pa.HostIndex.from_txome(tx, k, 4))
    assert n_shards > 20 and n_pass1 > len(want)
    assert cyc == 0
    assert got == want, "%d nodes only in the product, %d only in the reference's order" % (len(got - want), len(want - got))


def _rotations(s, k):
    """k-mers of the closed walk whose sequence (period + k - 1 bases) is s"""
    return {s[i:i + k] for i in range(len(s) - k + 1)}


@pytest.mark.parametrize("k", [20, 31])
def test_a_pure_kmer_cycle_is_cut_once_and_everything_else_agrees(tmp_path, k):
    """A transcript that is a tandem repeat (three periods of a 57-base unit) makes a closed loop of 57 k-mers, every one with
    exactly one left and one right extension and one colour: a PURE cycle. The reference cuts it where its walk happens to
    start (A4), the product's builders where theirs do: the cut is the one listed exception. Asserted here: the
    emulation sees the cycle (57 k-mers on it); the product has exactly ONE node for it, of period + k - 1 bases, whose k-mers are
    the cycle's and whose single left / right extension closes the loop; every other node is identical."""
    _, seqs = helpers.read_fasta()
    seqs = [s.upper() for s in seqs[:200]]
    rng = np.random.RandomState(5)
    unit = "".join("ACGT"[i] for i in rng.randint(0, 4, 57))
    seqs.insert(100, unit * 3)
    fa = tmp_path / "cycle.fa"
    fa.write_text("".join(">t%d\n%s\n" % (i, s) for i, s in enumerate(seqs)))
    want, _, cyc, _ = reference_order_nodes(seqs, k, "identity")
    assert cyc == 57                                                                  # the listed exception, not cyc == 0
    got = product_nodes(pa.build_index(str(fa), k, 2))
    extra = got - want
    assert want - got == set() and len(extra) == 1, (len(want - got), len(extra))
    seq, cls, le, re = next(iter(extra))
    assert len(seq) == 57 + k - 1 and cls == (100,) and len(le) == 1 and len(re) == 1
    doubled = unit * 3
    assert _rotations(seq, k) == {doubled[i:i + k] for i in range(57)}                # the node is the loop, cut somewhere
    assert seq[:k - 1] == seq[57:57 + k - 1]                                          # it closes on itself: its last k-1 bases are its first
    assert le == seq[56] and re == seq[k - 1]                                         # the extensions are the loop's own bases
