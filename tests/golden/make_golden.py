#!/usr/bin/env python3
"""Regenerates the golden result fixtures in this directory FROM THE ORACLE (oracle/pa_oracle.c), after the oracle
itself has passed the reference's own known-answer vectors (tests/test_oracle_*.py). The reference is Rust and cannot be
run in this image, so these are restatement outputs, not reference outputs — see DESIGN.md "Parity".

  small_fq_k{20,24,31}.tsv   one line per read of small.fq: "<id>\t<ids comma separated>\t<coverage>\t<mismatches>" or "<id>\tNone"
                            (the line format whose SHA-256 SURVEY.md appendix B lists for an independent Python model)
  synth_err_k31.tsv          400 simulated 150 bp reads with 1 % substitutions on gencode_small at K=31 (read seed 4):
                            "<read ascii>\t<ids>\t<coverage>\t<mismatches>" or "<read ascii>\tNone"
  synth_err_k64.tsv          the same at K=64 (two-word k-mers; the other k the reference's CLI accepts), 0.5 % substitutions
Inputs gencode_small.fa / small.fq are the reference's own test data files (test/gencode_small.fa, test/small.fq).
"""
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
sys.path.insert(0, str(HERE.parent.parent))
import helpers  # noqa: E402


def main():
    pa = helpers.pa
    ids, seqs = helpers.read_fastq()
    for k in (20, 24, 31):
        host = pa.build_index(str(helpers.FASTA), k, 8)
        oracle = helpers.Oracle(host)
        res, coff, cids, _ = oracle.map_reads(seqs, 2, 4)
        lines = helpers.result_lines(ids, res["mapped"], res["coverage"], res["mismatches"], coff, cids)
        (HERE / ("small_fq_k%d.tsv" % k)).write_text("".join(lines))
        print("k=%d sha256=%s" % (k, helpers.sha256_lines(lines)))
        if k == 31:
            tx = pa.Txome.from_host_index(host)
            tiles, lens = tx.simulate_host(150, 4, 400, 10000)
            reads = pa.unpack_tiles(tiles, lens, 5)
            res, coff, cids, _ = oracle.map_tiles(tiles, lens, 5, 2, 4)
            lines = helpers.result_lines(reads, res["mapped"], res["coverage"], res["mismatches"], coff, cids)
            (HERE / "synth_err_k31.tsv").write_text("".join(lines))
    host = pa.build_index(str(helpers.FASTA), 64, 8)
    tx = pa.Txome.from_host_index(host)
    tiles, lens = tx.simulate_host(150, 4, 400, 5000)
    reads = pa.unpack_tiles(tiles, lens, 5)
    res, coff, cids, _ = helpers.Oracle(host).map_tiles(tiles, lens, 5, 2, 4)
    lines = helpers.result_lines(reads, res["mapped"], res["coverage"], res["mismatches"], coff, cids)
    (HERE / "synth_err_k64.tsv").write_text("".join(lines))
    print("k=64 sha256=%s" % helpers.sha256_lines(lines))


if __name__ == "__main__":
    main()
