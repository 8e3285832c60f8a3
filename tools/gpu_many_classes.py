"""One-off soak of the many-classes list-mode cases (tests/helpers.many_classes_case) over more seeds than the tests run.
Usage: python tools/gpu_many_classes.py [seeds]   (GPU box)"""
import importlib, sys, tempfile
from pathlib import Path
sys.path.insert(0, "."); sys.path.insert(0, "tests")
pa = importlib.import_module("rust-pseudoaligner_amd")
import helpers
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = 0
for seed in range(2, 2 + n):
    for ordered in (False, True):
        with tempfile.TemporaryDirectory() as d:
            host, reads = helpers.many_classes_case(seed, Path(d), ordered=ordered)
            allowed = seed % 4
            res, coff, cids = pa.Pseudoaligner(host).map_batch(reads, allowed)
            o_res, o_coff, o_ids, _ = helpers.Oracle(host).map_reads(reads, allowed, 4)
            try:
                helpers.assert_same_as_oracle(res, coff, cids, o_res, o_coff, o_ids, "seed %d ordered %d" % (seed, ordered))
            except AssertionError as e:
                bad += 1
                print("MISMATCH", str(e)[:200])
print("many-classes cases %d, mismatching %d" % (2 * n, bad))
sys.exit(1 if bad else 0)
