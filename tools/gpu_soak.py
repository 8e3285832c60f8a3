"""Soak run on the GPU box: many seeds of the differential fuzz (tests/helpers.random_txome_case) and a determinism check of
two launches on 10 M config-2 reads (results + class ids identical). Usage: python tools/gpu_soak.py [seeds]"""
import importlib, sys, tempfile
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
pa = importlib.import_module("rust-pseudoaligner_amd")
import helpers
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = 0
for seed in range(10, 10 + nseeds):
    with tempfile.TemporaryDirectory() as d:
        host, k, reads, clean, allowed = helpers.random_txome_case(seed, Path(d), big=seed % 3 == 0, max_read=2000 if seed % 5 == 0 else 250)   # every third: list-mode heavy; every fifth: long reads
        if host is None:
            continue
        a = pa.Pseudoaligner(host)
        res, coff, cids = a.map_batch(reads, allowed)
        o_res, o_coff, o_ids, _ = helpers.Oracle(host).map_reads(clean, allowed, 4)
        try:
            helpers.assert_same_as_oracle(res, coff, cids, o_res, o_coff, o_ids, "seed %d k=%d" % (seed, k))
            # fused count table (window table / class-list table lookups of strict-subset results) through the device API
            tiles, lens, wpr = pa.encode_reads_host(reads)
            dev = torch.device("cuda", 0)
            d_tiles = torch.from_numpy(tiles.view(np.int64)).to(dev)
            d_lens = torch.from_numpy(np.asarray(lens, np.uint32).view(np.int32)).to(dev)
            n = len(reads)
            cap = a.arena_hint(n)
            d_res = torch.zeros(n * 4, dtype=torch.int32, device=dev)
            d_arena = torch.zeros(cap, dtype=torch.int32, device=dev)
            d_counts = torch.zeros(a.counts_len(), dtype=torch.int64, device=dev)
            a.map_count_batch_device(d_tiles.data_ptr(), d_lens.data_ptr(), n, wpr, d_res.data_ptr(), d_arena.data_ptr(), cap, d_counts.data_ptr(), allowed)
            a.map_finish()
            assert np.array_equal(d_counts.cpu().numpy(), helpers.counts_reference(o_res, o_coff, o_ids, host)), "seed %d: count table differs" % seed
        except AssertionError as e:
            bad += 1
            print("MISMATCH", str(e)[:300])
print("fuzz seeds %d, mismatching %d" % (nseeds, bad))
hi = pa.HostIndex.build_fasta(str(helpers.FASTA), 24, 8)
a = pa.Pseudoaligner(hi)
tx = pa.Txome.from_host_index(hi)
n, wpr = 10_000_000, 4
dev = torch.device("cuda", 0)
d_tiles = torch.zeros(pa.lib().pa_tiles_words(n, wpr), dtype=torch.int64, device=dev)
d_lens = torch.zeros(n, dtype=torch.int32, device=dev)
tx.simulate_device(100, 3, n, d_tiles.data_ptr(), d_lens.data_ptr(), 5000, 0, wpr)
cap = a.arena_hint(n)
outs = []
for it in range(3):
    d_res = torch.zeros(n * 4, dtype=torch.int32, device=dev)
    d_arena = torch.zeros(cap, dtype=torch.int32, device=dev)
    d_counts = torch.zeros(a.counts_len(), dtype=torch.int64, device=dev)
    a.map_count_batch_device(d_tiles.data_ptr(), d_lens.data_ptr(), n, wpr, d_res.data_ptr(), d_arena.data_ptr(), cap, d_counts.data_ptr(), 2)
    used, _ = a.map_finish()
    res = d_res.cpu().numpy().view(pa.RESULT_DTYPE)
    coff, ids = pa.gather_classes(res, d_arena[: max(used, 1)].cpu().numpy().view(np.uint32), hi)
    outs.append((res[["coverage", "mismatches", "class_len"]].copy(), coff, ids, d_counts.cpu().numpy()))
same = all(np.array_equal(outs[0][j], o[j]) for o in outs[1:] for j in range(4))
print("three launches of 10 M reads identical:", same)
sys.exit(0 if (bad == 0 and same) else 1)
