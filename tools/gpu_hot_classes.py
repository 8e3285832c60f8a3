"""Experiment: how much do a few hot classes cost the fused count table? Tiny transcriptome (first N transcripts of
gencode_small), 10 M simulated reads, map with and without counts (GPU box)."""
import importlib, sys, time
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
pa = importlib.import_module("rust-pseudoaligner_amd")
import helpers
ntx = int(sys.argv[1]) if len(sys.argv) > 1 else 50
names, seqs = helpers.read_fasta()
seqs = [s for s in seqs if len(s) >= 200][:ntx]
fa = "/tmp/hot.fa"
open(fa, "w").write("".join(">t%d\n%s\n" % (i, s) for i, s in enumerate(seqs)))
hi = pa.HostIndex.build_fasta(fa, 24, 8)
a = pa.Pseudoaligner(hi)
tx = pa.Txome.from_host_index(hi)
n, L, wpr = 10_000_000, 100, 4
dev = torch.device("cuda", 0)
d_tiles = torch.zeros(pa.lib().pa_tiles_words(n, wpr), dtype=torch.int64, device=dev)
d_lens = torch.zeros(n, dtype=torch.int32, device=dev)
tx.simulate_device(L, 4, n, d_tiles.data_ptr(), d_lens.data_ptr(), 0, 0, wpr)
cap = a.arena_hint(n)
d_res = torch.zeros(n * 4, dtype=torch.int32, device=dev)
d_arena = torch.zeros(cap, dtype=torch.int32, device=dev)
d_counts = torch.zeros(a.counts_len(), dtype=torch.int64, device=dev)
def run(counts):
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        if counts: a.map_count_batch_device(d_tiles.data_ptr(), d_lens.data_ptr(), n, wpr, d_res.data_ptr(), d_arena.data_ptr(), cap, d_counts.data_ptr(), 2)
        else: a.map_batch_device(d_tiles.data_ptr(), d_lens.data_ptr(), n, wpr, d_res.data_ptr(), d_arena.data_ptr(), cap, 2, 0)
        a.map_finish(); dt = time.time() - t0
    return dt
print("transcripts %d classes %d: map %.3f ms, map+count %.3f ms" % (ntx, a.counts_len() - 3, run(False) * 1e3, run(True) * 1e3))
