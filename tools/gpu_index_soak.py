"""Soak of the GPU index builder (csrc/index_build.hip) against the CPU builder: many random small transcriptomes (repeats, pure
cycles on a two-letter alphabet, every k from 8 to 64), the flat indexes must be identical array for array.
Usage (GPU box): python tools/gpu_index_soak.py [seeds]"""
import importlib, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
pa = importlib.import_module("rust-pseudoaligner_amd")
from test_index_build import pack, random_txome

KEYS = ("node_seq", "node_start", "node_len", "node_exts", "node_colour", "ec_offset", "ec_ids")
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = 0
for seed in range(1000, 1000 + nseeds):
    rng = np.random.RandomState(seed)
    k = int(rng.randint(8, 65))
    seqs = random_txome(rng, rng.randint(1, 40), "ACGT" if seed % 3 else "AC")
    if seed % 4 == 0:
        seqs += ["AC" * int(rng.randint(10, 60)), "CA" * int(rng.randint(10, 60)), "A" * int(rng.randint(1, 90)), "ACG" * int(rng.randint(5, 40))]
    if seed % 7 == 0:
        seqs.insert(int(rng.randint(0, len(seqs))), "")
    words, tx_start = pack(seqs)
    hg, hc = pa.HostIndex.build_packed_device(words, tx_start, k, 0), pa.HostIndex.build_packed(words, tx_start, k, 1 + seed % 5)
    g, c = hg.arrays(), hc.arrays()   # (views into the two indexes: both objects stay alive while they are compared)
    same = all(g[x] == c[x] for x in ("k", "num_nodes", "num_classes", "num_transcripts")) and all(np.array_equal(g[x], c[x]) for x in KEYS)
    if not same:
        bad += 1
        print("MISMATCH seed %d k=%d" % (seed, k))
print("index builder soak: %d seeds, mismatching %d" % (nseeds, bad))
