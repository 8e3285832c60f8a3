"""Shared cases of the index hand-over tests: indexes whose arrays were NOT produced by the product's builders —
what a Rust exporter over `Pseudoaligner<K>`'s pub fields (/root/reference/src/pseudoaligner.rs:26-33) hands to
pa_host_index_from_flat / pa_index_create: the reference's node order, the reference's unitig break points, the reference's
class numbering. Used by tests/test_import_emu.py (CPU tier: the kernel's lane steps on the host) and tests/test_gpu_import.py
(GPU tier: the HIP path through the C ABI)."""
import numpy as np

import helpers

pa = helpers.pa


def reads_for(host, read_len, seed=11):
    """small.fq-like clean reads + 1 % and 4 % substitution reads of the index's own transcripts: (tiles, lens, wpr) list"""
    out = []
    for ppm, n in ((0, 6000), (10000, 12000), (40000, 12000)):
        out.append((ppm,) + helpers.error_reads(host, read_len, n, ppm, seed + ppm // 10000))
    return out


def check_foreign(own, foreign, mapper, read_len=150, alloweds=(0, 1, 2, 3), what=""):
    """`mapper(foreign, tiles, lens, wpr, allowed)` -> (results, coff, ids) must equal the oracle built on the SAME foreign arrays;
    the oracle on the product builder's maximal-unitig index must differ somewhere (break points are observable: a node visit
    resets the mismatch budget, /root/reference/src/pseudoaligner.rs:215-219) or the test would be vacuous; and the diff tool
    must call the two indexes equivalent."""
    rc, why = own.compare(foreign)
    assert rc == 0 and (why.startswith("equivalent") or why.startswith("identical")), why
    o_foreign, o_own = helpers.Oracle(foreign), helpers.Oracle(own)
    differ = 0
    for ppm, tiles, lens, wpr in reads_for(own, read_len):
        for allowed in alloweds:
            want = o_foreign.map_tiles(tiles, lens, wpr, allowed, 8)
            res, coff, ids = mapper(foreign, tiles, lens, wpr, allowed)
            helpers.assert_same_as_oracle(res, coff, ids, want[0], want[1], want[2], "%s ppm=%d allowed=%d" % (what, ppm, allowed))
            if ppm:
                mine = o_own.map_tiles(tiles, lens, wpr, allowed, 8)
                differ += int((mine[0] != want[0]).sum())
    return differ


def small_fq_tiles():
    _, seqs = helpers.read_fastq()
    return pa.encode_reads_host(seqs)


def two_pass_index(seqs, k, num_tx):
    """the node set the reference's two-pass sharded build produces (tests/test_reference_build_order.py), as a flat index"""
    import test_reference_build_order as rbo
    nodes, n_shards, cyc, n_pass1 = rbo.reference_order_nodes(seqs, k, "identity")
    assert cyc == 0
    return helpers.index_from_node_set(nodes, k, num_tx, seed=k)


def pass_one_index(seqs, k, num_tx):
    """the node set after the reference's FIRST pass only (every MSP shard compressed on its own: paths end at shard seams):
    a legal, non-maximal unitig set with the reference's own kind of break points"""
    import test_reference_build_order as rbo
    nodes = rbo.reference_order_nodes(seqs, k, "identity", pass_one_only=True)[0]
    return helpers.index_from_node_set(nodes, k, num_tx, seed=k + 1)
