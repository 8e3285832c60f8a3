"""CPU index construction (dbg_build.cpp) against a naive Python builder with the reference's semantics
(src/build_index.rs:127-179, src/equiv_classes.rs:62-91), determinism, and the container / flat-index round trips."""
import ctypes as C

import numpy as np
import pytest

import helpers

pa = helpers.pa


def pack(seqs):
    tx_start = np.zeros(len(seqs) + 1, np.uint64)
    tx_start[1:] = np.cumsum([len(s) for s in seqs])
    words = np.zeros(int(tx_start[-1]) // 32 + 3, np.uint64)
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    pos = 0
    for s in seqs:
        for ch in s:
            words[pos >> 5] |= np.uint64(code[ch]) << np.uint64(2 * (pos & 31))
            pos += 1
    return words, tx_start


def naive_graph(seqs, k):
    """stranded coloured compacted DBG: nodes = maximal paths of k-mers that are each other's unique extension and share
    a colour. Returns ({(sequence, colour tuple, left exts, right exts)} for paths with a start, set of k-mers on pure cycles)."""
    colour, lext, rext = {}, {}, {}
    for i, s in enumerate(seqs):
        for p in range(len(s) - k + 1):
            km = s[p:p + k]
            colour.setdefault(km, set()).add(i)
            lext.setdefault(km, set())
            rext.setdefault(km, set())
            if p > 0:
                lext[km].add(s[p - 1])
            if p + k < len(s):
                rext[km].add(s[p + k])
    colour = {km: tuple(sorted(v)) for km, v in colour.items()}

    def right_join(x):
        if len(rext[x]) != 1:
            return None
        y = x[1:] + next(iter(rext[x]))
        if y == x or len(lext[y]) != 1 or colour[y] != colour[x]:
            return None
        return y

    def left_joinable(x):
        if len(lext[x]) != 1:
            return False
        z = next(iter(lext[x])) + x[:-1]
        return z != x and len(rext[z]) == 1 and colour[z] == colour[x]

    nodes, seen = set(), set()
    for x in colour:
        if left_joinable(x):
            continue
        path, cur = x, x
        seen.add(x)
        while True:
            y = right_join(cur)
            if y is None or y in seen:
                break
            seen.add(y)
            path += y[-1]
            cur = y
        nodes.add((path, colour[x], "".join(sorted(lext[x])), "".join(sorted(rext[cur]))))
    return nodes, set(colour) - seen


def graph_of(host):
    a = host.arrays()
    k = a["k"]
    nodes = []
    for n in range(a["num_nodes"]):
        s, l = int(a["node_start"][n]), int(a["node_len"][n])
        pos = np.arange(s, s + l)
        codes = (a["node_seq"][pos >> 5] >> ((pos & 31) * 2).astype(np.uint64)) & np.uint64(3)
        seq = "".join("ACGT"[int(c)] for c in codes)
        c = int(a["node_colour"][n])
        cl = tuple(a["ec_ids"][int(a["ec_offset"][c]):int(a["ec_offset"][c + 1])].tolist())
        e = int(a["node_exts"][n])
        nodes.append((seq, cl, "".join("ACGT"[b] for b in range(4) if e >> (4 + b) & 1), "".join("ACGT"[b] for b in range(4) if e >> b & 1)))
    return k, nodes


def random_txome(rng, n_tx, alphabet="ACGT"):
    segs = ["".join(rng.choice(list(alphabet), rng.randint(5, 60))) for _ in range(12)]
    out = []
    for _ in range(n_tx):
        out.append("".join(segs[j] for j in rng.choice(len(segs), rng.randint(1, 6))))
    return out


@pytest.mark.parametrize("seed", range(12))
def test_builder_matches_naive_graph(built, seed):
    rng = np.random.RandomState(seed)
    k = int(rng.choice([8, 9, 12, 16, 33, 48, 64]))   # one- and two-word k-mers
    seqs = random_txome(rng, rng.randint(2, 25), "ACGT" if seed % 3 else "AC")
    words, tx_start = pack(seqs)
    host = pa.HostIndex.build_packed(words, tx_start, k, 1 + seed % 4)
    _, got = graph_of(host)
    want, cyc = naive_graph(seqs, k)
    got_set = set(got)
    assert len(got_set) == len(got)
    cyc_nodes = {g for g in got_set if g[0][:k] in cyc}
    assert got_set - cyc_nodes == want
    kmers_in_cyc_nodes = {g[0][p:p + k] for g in cyc_nodes for p in range(len(g[0]) - k + 1)}
    assert kmers_in_cyc_nodes == cyc
    # every class is referenced, sorted, non-empty
    a = host.arrays()
    assert set(a["node_colour"].tolist()) == set(range(a["num_classes"]))


def test_hand_made_graph(built):
    #   tx0 = P + M + Q,  tx1 = R + M + S  (shared middle M): M is one node with colour [0,1] and two exts on both sides
    P, M, Q, R, S = "ACGTACGGTTCA", "GGATCCTTAGCAAT", "TTTGACCGTA", "CCCATTGAGG", "AAGTCGGCAT"
    seqs = [P + M + Q, R + M + S]
    k = 8
    words, tx_start = pack(seqs)
    _, nodes = graph_of(pa.HostIndex.build_packed(words, tx_start, k, 2))
    by_colour = {}
    for seq, cl, le, re in nodes:
        by_colour.setdefault(cl, []).append((seq, le, re))
    shared = by_colour[(0, 1)]
    assert len(shared) == 1
    seq, le, re = shared[0]
    assert seq == M and sorted(le) == sorted({P[-1], R[-1]}) and sorted(re) == sorted({Q[0], S[0]})
    assert sorted(s for s, _, _ in by_colour[(0,)]) == sorted([(P + M)[: len(P) + k - 1], (M + Q)[len(M) - k + 1:]])
    assert sorted(s for s, _, _ in by_colour[(1,)]) == sorted([(R + M)[: len(R) + k - 1], (M + S)[len(M) - k + 1:]])


def test_short_transcripts_contribute_nothing(built):
    seqs = ["ACGTACG", "ACGTACGTTGCAAGGCT"]   # first is shorter than k = 8 (src/build_index.rs:134,148-150)
    words, tx_start = pack(seqs)
    host = pa.HostIndex.build_packed(words, tx_start, 8, 1)
    a = host.arrays()
    assert a["num_transcripts"] == 2 and a["num_classes"] == 1 and a["ec_ids"].tolist() == [1]


def test_builder_is_deterministic_across_thread_counts(built):
    i1 = pa.HostIndex.build_fasta(str(helpers.FASTA), 24, 1)
    i8 = pa.HostIndex.build_fasta(str(helpers.FASTA), 24, 8)
    h1, h8 = i1.arrays(), i8.arrays()
    for key in ("node_seq", "node_start", "node_len", "node_exts", "node_colour", "ec_offset", "ec_ids"):
        assert np.array_equal(h1[key], h8[key]), key
    # gencode_small at K=24 -> 1 165 762 distinct 24-mers
    assert int((h1["node_len"] - 23).sum()) == 1165762


def test_container_and_flat_round_trip(built, small_index, tmp_path):
    host = small_index(20)
    path = tmp_path / "idx.bin"
    host.save(str(path))
    back = pa.HostIndex.load(str(path))
    a, b = host.arrays(), back.arrays()
    for key in a:
        assert np.array_equal(a[key], b[key]), key
    assert back.tx_names()[:3] == host.tx_names()[:3] == ["ENST00000456328.2", "ENST00000450305.2", "ENST00000488147.1"]
    assert back.tx_genes()[0] == "ENSG00000223972.5"          # Gencode header format (src/utils.rs:126-134)
    reimported = pa.HostIndex.from_flat(host.flat())
    flat = reimported.arrays()
    for key in a:
        assert np.array_equal(a[key], flat[key]), key
    path.write_bytes(b"not an index")
    with pytest.raises(pa.PaError):
        pa.HostIndex.load(str(path))
    with pytest.raises(pa.PaError):
        pa.HostIndex.load(str(tmp_path / "missing.bin"))


def test_invalid_arguments_fail_loudly(built):
    words, tx_start = pack(["ACGTACGTACGTACGT"])
    for k in (0, 7, 65):
        with pytest.raises(pa.PaError):
            pa.HostIndex.build_packed(words, tx_start, k, 1)
    with pytest.raises(pa.PaError):
        pa.HostIndex.build_fasta("/nonexistent.fa", 20, 1)
    tiny = pa.HostIndex.build_packed(words, tx_start, 8, 1)
    flat = tiny.flat()
    bad = np.array([3], np.uint32)          # colour out of range
    flat.node_colour = bad.ctypes.data
    with pytest.raises(pa.PaError):
        pa.HostIndex.from_flat(flat)


def test_gene_level_collapse(small_index):
    """tx_gene_mapping (src/pseudoaligner.rs:32): class counts fold into per-gene counts; multi-gene classes are kept apart"""
    host = small_index(24)
    a = host.arrays()
    genes = host.tx_genes()
    tx_gene, names = host.genes()
    assert len(tx_gene) == a["num_transcripts"] and [names[g] for g in tx_gene] == genes
    assert names == list(dict.fromkeys(genes))                                   # numbered by first appearance
    rng = np.random.RandomState(11)
    counts = rng.randint(0, 50, a["num_classes"] + 3).astype(np.uint64)
    got = host.collapse_to_genes(counts)
    want = np.zeros(len(names) + 1, np.uint64)
    eo, ei = a["ec_offset"].astype(np.int64), a["ec_ids"]
    for c in range(a["num_classes"]):
        gs = set(tx_gene[ei[eo[c]:eo[c + 1]]].tolist())
        want[gs.pop() if len(gs) == 1 else len(names)] += counts[c]
    assert np.array_equal(got, want) and got.sum() == counts[:-3].sum()
    with pytest.raises(pa.PaError):
        host.collapse_to_genes(counts[:-1])


def test_mappability(tmp_path):
    """analyze_graph / write_mappability_tsv (src/mappability.rs:91-156) against a brute-force count over the transcripts'
    own k-mers: a k-mer shared by j transcripts (g genes) adds one to slot min(j, 11) - 1 (min(g, 11) - 1) of each."""
    k, W = 24, 11
    names, seqs = helpers.read_fasta()
    names, seqs = names[:80], [s.upper() for s in seqs[:80]]
    names.append("ENSTSHORT.1|ENSGSHORT.1|-|-|SHORT-201|SHORT|10|x|")          # shorter than K: no k-mers, fractions 0/0
    seqs.append("ACGTACGTAC")
    fa = tmp_path / "m.fa"
    fa.write_text("".join(">%s\n%s\n" % (n, s) for n, s in zip(names, seqs)))
    host = pa.build_index(str(fa), k, 4)
    genes = host.tx_genes()
    owners = {}
    for t, s in enumerate(seqs):
        for i in range(len(s) - k + 1):
            owners.setdefault(s[i:i + k], set()).add(t)
    want_t = np.zeros((len(seqs), W), np.uint64)
    want_g = np.zeros((len(seqs), W), np.uint64)
    for txs in owners.values():
        j = min(len(txs), W) - 1
        g = min(len({genes[t] for t in txs}), W) - 1
        for t in txs:
            want_t[t, j] += 1
            want_g[t, g] += 1
    tm, gm = host.mappability()
    assert np.array_equal(tm, want_t) and np.array_equal(gm, want_g)
    assert (tm.sum(1) == gm.sum(1)).all() and tm[-1].sum() == 0
    out = tmp_path / "tx_mappability.tsv"
    host.write_mappability_tsv(out)
    lines = out.read_text().split("\n")
    assert lines[0] == "tx_name\tgene_name\ttx_kmer_count\tfrac_kmer_unique_tx\tfrac_kmer_unique_gene" and lines[-1] == ""
    tx_names = host.tx_names()
    for t, line in enumerate(lines[1:-1]):
        name, gene, total, ft, fg = line.split("\t")
        assert (name, gene, int(total)) == (tx_names[t], genes[t], int(tm[t].sum()))
        if int(total) == 0:
            assert (ft, fg) == ("NaN", "NaN")
            continue
        for txt, num in ((ft, tm[t, 0]), (fg, gm[t, 0])):
            x = float(num) / float(total)
            assert float(txt) == x and "e" not in txt.lower()
            # Rust's `{}`: the shortest digits that round-trip ("1", "0.5", "0.3333333333333333")
            assert txt == (repr(x)[:-2] if repr(x).endswith(".0") else repr(x)) or "e" in repr(x)


def _flat_from(arr, k, num_tx):
    """a FlatIndex over numpy arrays (kept alive by the returned tuple), as an exporter on the Rust side would fill it"""
    f = pa._ffi.FlatIndex()
    keep = {n: np.ascontiguousarray(arr[n]) for n in ("node_seq", "node_start", "node_len", "node_exts", "node_colour", "ec_offset", "ec_ids")}
    f.k, f.num_nodes, f.num_classes, f.num_transcripts = k, len(keep["node_len"]), len(keep["ec_offset"]) - 1, num_tx
    f.seq_bases = int(keep["node_start"][-1])
    for n, v in keep.items():
        setattr(f, n, v.ctypes.data)
    f.node_redge = f.node_ledge = None
    return f, keep


def _repack_nodes(a, order, split=None):
    """node arrays of `a` in a new order (and, optionally, node `split` cut into two nodes that overlap by k-1 bases — a
    different unitig break point for the same k-mers)"""
    k = a["k"]
    seqs, exts, colour = [], [], []
    for n in order:
        st, ln = int(a["node_start"][n]), int(a["node_len"][n])
        bases = [(int(a["node_seq"][(st + i) >> 5]) >> (2 * ((st + i) & 31))) & 3 for i in range(ln)]
        if n == split:
            cut = ln // 2
            left, right = bases[:cut + k - 1], bases[cut:]
            seqs += [left, right]
            exts += [(int(a["node_exts"][n]) & 0xF0) | (1 << right[k - 1]), (int(a["node_exts"][n]) & 0x0F) | (16 << left[len(left) - k])]
            colour += [int(a["node_colour"][n])] * 2
        else:
            seqs.append(bases)
            exts.append(int(a["node_exts"][n]))
            colour.append(int(a["node_colour"][n]))
    start = np.zeros(len(seqs) + 1, np.uint64)
    start[1:] = np.cumsum([len(s) for s in seqs])
    words = np.zeros((int(start[-1]) + 31) // 32 + 2, np.uint64)
    flat = np.array([b for s in seqs for b in s], np.uint64)
    idx = np.arange(len(flat))
    np.bitwise_or.at(words, idx >> 5, flat << ((idx & 31) * 2).astype(np.uint64))
    return dict(node_seq=words, node_start=start, node_len=np.array([len(s) for s in seqs], np.uint32), node_exts=np.array(exts, np.uint8),
                node_colour=np.array(colour, np.uint32))


def test_index_compare_is_numbering_and_break_point_independent(built, tmp_path):
    """pa_host_index_compare, the diff tool of the index interchange: an index with its nodes in another order
    and its classes renumbered is identical; one with a unitig cut at another place is equivalent (k-mer level); one with a
    changed id list / a lost node is different"""
    fa = tmp_path / "t.fa"
    _, seqs = helpers.read_fasta()
    fa.write_text("".join(">t%d|g%d\n%s\n" % (i, i // 2, s) for i, s in enumerate(seqs[:12])))
    host = pa.HostIndex.build_fasta(str(fa), 20, 2)
    a = host.arrays()
    rng = np.random.RandomState(3)
    order = rng.permutation(a["num_nodes"])
    perm = rng.permutation(a["num_classes"])                       # class c is renumbered perm[c]
    off = a["ec_offset"].astype(np.int64)
    lists = [a["ec_ids"][off[c]:off[c + 1]] for c in range(a["num_classes"])]
    new_lists = [None] * a["num_classes"]
    for c, l in enumerate(lists):
        new_lists[perm[c]] = l
    ec_offset = np.zeros(a["num_classes"] + 1, np.uint64)
    ec_offset[1:] = np.cumsum([len(l) for l in new_lists])
    base = dict(ec_offset=ec_offset, ec_ids=np.concatenate(new_lists).astype(np.uint32))

    def imported(nodes, classes=base):
        arr = dict(nodes, **classes)
        arr["node_colour"] = perm[nodes["node_colour"]].astype(np.uint32)
        f, keep = _flat_from(arr, a["k"], a["num_transcripts"])
        return pa.HostIndex.from_flat(f)

    rc, why = host.compare(imported(_repack_nodes(a, order)))
    assert rc == 0 and why.startswith("identical"), why
    long_node = int(np.argmax(a["node_len"]))
    rc, why = host.compare(imported(_repack_nodes(a, order, split=long_node)))
    assert rc == 0 and why.startswith("equivalent"), why
    rc, why = host.compare(imported(_repack_nodes(a, order, split=long_node)), max_kmers=10)
    assert rc == 2, why                                            # undecided: the k-mer level check was not allowed
    wrong = dict(base, ec_ids=base["ec_ids"].copy())
    single = next(c for c in range(a["num_classes"]) if ec_offset[c + 1] - ec_offset[c] == 1)
    wrong["ec_ids"][int(ec_offset[single])] = (int(wrong["ec_ids"][int(ec_offset[single])]) + 1) % a["num_transcripts"]   # one id list changed
    rc, why = host.compare(imported(_repack_nodes(a, order), wrong))
    assert rc == 1 and "different class" in why, why
    rc, why = host.compare(imported(_repack_nodes(a, order[:-1])))
    assert rc == 1, why                                            # a node (its k-mers) is missing
    assert host.compare(pa.HostIndex.build_fasta(str(fa), 21, 2))[0] == 1


def test_fasta_id_ends_at_any_white_space(built, tmp_path):
    """record.id() of bio 1.5's FASTA reader cuts the header at the first white-space character of any kind: the id
    names the transcript and seeds the bases substituted for N (from_acgt_bytes_hashn, src/utils.rs:76), so the same record
    with a space- or a tab-separated description must give the same names AND the same packed transcript."""
    rng = np.random.RandomState(5)
    body = "".join("ACGT"[i] for i in rng.randint(0, 4, 300))
    body = body[:100] + "NNNNN" + body[105:200] + "n" + body[201:]
    texts = {"space": ">tx1 gene=G1 extra\n%s\n>tx2 gene=G1\n%s\n" % (body, body[::-1]),
             "tab": ">tx1\tgene=G1 extra\r\n%s\r\n>tx2\tgene=G1  \t\r\n%s\r\n" % (body, body[::-1]),
             "ff": ">tx1\x0cgene=G1 extra\n%s\n>tx2\x0bgene=G1\n%s\n" % (body, body[::-1])}
    packed, names = {}, {}
    for tag, text in texts.items():
        p = tmp_path / (tag + ".fa")
        p.write_bytes(text.encode())
        host = pa.build_index(str(p), 20, 2)
        names[tag] = host.tx_names()
        packed[tag] = [a.copy() for a in host.transcripts()]
    assert names["space"] == ["tx1", "tx2"] == names["tab"] == names["ff"]
    for tag in ("tab", "ff"):
        assert all(np.array_equal(x, y) for x, y in zip(packed["space"], packed[tag])), tag
