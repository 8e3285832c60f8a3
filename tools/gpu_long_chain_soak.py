"""Soak of the LEFT path across chain blocks on the GPU box: many seeds of helpers.long_chain_case (nested transcripts cut on and around
multiples of 64, reads of 300..500 bases whose first hit lies far into the read, allowed 6 / 12) against the oracle, bit exact.
Usage: python tools/gpu_long_chain_soak.py [seeds]"""
import importlib, sys, tempfile
from pathlib import Path
sys.path.insert(0, "."); sys.path.insert(0, "tests")
pa = importlib.import_module("rust-pseudoaligner_amd")
import helpers
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
bad = 0
for seed in range(100, 100 + n):
    with tempfile.TemporaryDirectory() as d:
        host, reads, allowed = helpers.long_chain_case(seed, Path(d))
        a = pa.Pseudoaligner(host)
        res, coff, cids = a.map_batch(reads, allowed)
        o_res, o_coff, o_ids, _ = helpers.Oracle(host).map_reads(reads, allowed, 4)
        try:
            helpers.assert_same_as_oracle(res, coff, cids, o_res, o_coff, o_ids, "long chains seed %d" % seed)
        except AssertionError as e:
            bad += 1
            print("MISMATCH", str(e)[:300])
print("long-chain seeds %d, mismatching %d" % (n, bad))
