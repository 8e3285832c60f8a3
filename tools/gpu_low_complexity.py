"""Experiment: cost of the list-mode steps (SCAN / COOP) on low-complexity sequence. A transcriptome over a two-letter
alphabet makes most k-mers shared by many transcripts, so nearly every read leaves window mode and intersects long class
lists. Usage: python tools/gpu_low_complexity.py [ntx] [alternative .so]   (GPU box)"""
import importlib, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
pa = importlib.import_module("rust-pseudoaligner_amd")
if len(sys.argv) > 2:
    pa._ffi._build.PRODUCT_SO = Path(sys.argv[2]).resolve()
ntx = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.default_rng(7)
motifs = ["".join(rng.choice(list("AC"), size=40)) for _ in range(60)]
fa = "/tmp/lowc.fa"
with open(fa, "w") as f:
    for i in range(ntx):
        body = "".join(motifs[j] for j in rng.integers(0, len(motifs), size=12))
        tail = "".join(rng.choice(list("ACGT"), size=60))
        f.write(">t%d\n%s%s\n" % (i, body, tail))
hi = pa.HostIndex.build_fasta(fa, 20, 8)
a = pa.Pseudoaligner(hi)
tx = pa.Txome.from_host_index(hi)
n, L, wpr = 2_000_000, 100, 4
dev = torch.device("cuda", 0)
d_tiles = torch.zeros(pa.lib().pa_tiles_words(n, wpr), dtype=torch.int64, device=dev)
d_lens = torch.zeros(n, dtype=torch.int32, device=dev)
tx.simulate_device(L, 4, n, d_tiles.data_ptr(), d_lens.data_ptr(), 0, 0, wpr)
cap = max(a.arena_hint(n), 64 * n)
d_res = torch.zeros(n * 4, dtype=torch.int32, device=dev)
d_arena = torch.zeros(cap, dtype=torch.int32, device=dev)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    a.map_batch_device(d_tiles.data_ptr(), d_lens.data_ptr(), n, wpr, d_res.data_ptr(), d_arena.data_ptr(), cap, 2, 0)
    used, _ = a.map_finish(); dt = time.time() - t0
res = d_res.cpu().numpy().view(pa.RESULT_DTYPE)
print("transcripts %d classes %d: %d reads in %.3f ms (%.1f M reads/s), arena used %d, mapped %.1f %%, mean class %.1f" % (
    ntx, a.counts_len() - 3, n, dt * 1e3, n / dt / 1e6, used, 100 * np.mean(res["class_len"] > 0), res["class_len"].mean()))
