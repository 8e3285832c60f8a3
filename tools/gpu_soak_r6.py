"""Round-6 soak on the GPU box: the families added this round, by the hundred — repeat-family transcriptomes (pending classes answered from the class
bitmaps; the fused count table), branch points on block seams, tandem repeats with reads of exactly K bases, foreign layouts of all of them, and reads of
20-200 kb (the wide lane state). Every read bit-exact against the oracle. usage: python tools/gpu_soak_r6.py [scale]   (scale 1: ~10 minutes)"""
import importlib, sys, tempfile
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
pa = importlib.import_module("rust-pseudoaligner_amd")
import helpers

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
bad = 0
done = {}


def check(host, reads_or_tiles, allowed, what, counts=False):
    global bad
    a = pa.Pseudoaligner(host)
    try:
        if isinstance(reads_or_tiles, tuple):
            tiles, lens, wpr = reads_or_tiles
            dev = torch.device("cuda", 0)
            n = len(lens)
            d_tiles = torch.from_numpy(tiles.view(np.int64)).to(dev)
            d_lens = torch.from_numpy(np.ascontiguousarray(lens).view(np.int32)).to(dev)
            cap = a.arena_hint(n)
            d_res = torch.zeros(n * 4, dtype=torch.int32, device=dev)
            d_arena = torch.zeros(cap, dtype=torch.int32, device=dev)
            d_counts = torch.zeros(a.counts_len(), dtype=torch.int64, device=dev)
            a.map_count_batch_device(d_tiles.data_ptr(), d_lens.data_ptr(), n, wpr, d_res.data_ptr(), d_arena.data_ptr(), cap, d_counts.data_ptr(), allowed)
            used, _ = a.map_finish()
            res = d_res.cpu().numpy().view(pa.RESULT_DTYPE)
            coff, cids = pa.gather_classes(res, d_arena[: max(used, 1)].cpu().numpy().view(np.uint32), host)
            want = helpers.Oracle(host).map_tiles(tiles, lens, wpr, allowed, 16)
            helpers.assert_same_as_oracle(res, coff, cids, want[0], want[1], want[2], what)
            assert np.array_equal(d_counts.cpu().numpy(), helpers.counts_reference_fast(want[0], want[1], want[2], host)), what + ": count table differs"
        else:
            res, coff, cids = a.map_batch(reads_or_tiles, allowed)
            want = helpers.Oracle(host).map_reads(reads_or_tiles, allowed, 16)
            helpers.assert_same_as_oracle(res, coff, cids, want[0], want[1], want[2], what)
    except AssertionError as e:
        bad += 1
        print("MISMATCH", str(e)[:300], flush=True)


# 1. repeat-family transcriptomes of many shapes (transcript counts decide the bitmap's words; divergence decides the class sizes)
rng = np.random.RandomState(1)
n1 = int(60 * scale)
for i in range(n1):
    genes = int(rng.randint(200, 3000))
    tx = pa.Txome.synthesize_repeats(genes, int(genes * rng.uniform(1.5, 4.0)), 100 + i, families=int(rng.randint(1, 30)), young_families=int(rng.randint(0, 8)),
                                     element_len=int(rng.choice([60, 150, 300, 600])), gene_fraction_ppm=int(rng.choice([50000, 200000, 600000])),
                                     young_div_lo_ppm=int(rng.choice([0, 10000, 30000])), young_div_hi_ppm=60000, low_complexity_genes=int(rng.randint(0, 40)))
    k = int(rng.choice([16, 24, 31, 33, 64]))
    host = pa.HostIndex.from_txome_device(tx, k, 0)
    ppm = int(rng.choice([0, 10000, 30000]))
    L = int(rng.choice([100, 150, 250]))
    tiles, lens = tx.simulate_host(L, i, 150000, ppm)
    check(host, (tiles, lens, pa.lib().pa_words_per_read(L)), int(rng.choice([0, 1, 2, 3])), "repeats %d genes=%d k=%d L=%d ppm=%d" % (i, genes, k, L, ppm))
    if i % 4 == 0:
        foreign, _ = helpers.foreign_index(host, 7000 + i, cut_frac=0.5)
        check(foreign, (tiles, lens, pa.lib().pa_words_per_read(L)), 2, "repeats %d foreign" % i)
done["repeat transcriptomes"] = n1
print("repeat-family transcriptomes: %d, mismatching so far %d" % (n1, bad), flush=True)

# 2. branch points / tandem repeats, and their foreign layouts
n2 = int(600 * scale)
for seed in range(200, 200 + n2):
    with tempfile.TemporaryDirectory() as d:
        for fam in (helpers.branch_case, helpers.tandem_case):
            host, reads, allowed = fam(seed, Path(d))
            if host is None:
                continue
            check(host, reads, allowed, "%s seed %d" % (fam.__name__, seed))
            if seed % 4 == 0:
                foreign, _ = helpers.foreign_index(host, seed, cut_frac=0.6)
                check(foreign, reads, allowed, "%s seed %d foreign" % (fam.__name__, seed))
done["branch + tandem seeds"] = n2
print("branch / tandem seeds: %d, mismatching so far %d" % (n2, bad), flush=True)

# 3. long reads (the wide lane state): transcripts of 20-200 kb as reads, with substitutions, at several k
n3 = int(12 * scale)
for i in range(n3):
    with tempfile.TemporaryDirectory() as d:
        k = int([20, 24, 31, 40, 64][i % 5])
        host, seqs = helpers.long_transcript_case(Path(d), k=k, long_len=int(rng.randint(20000, 200000)), seed=50 + i)
        r2 = np.random.RandomState(i)
        reads = list(seqs)
        for s in seqs[-2:] + seqs[:8]:
            t = list(s)
            for j in r2.randint(0, len(t), max(1, len(t) // int(r2.choice([100, 300, 1000])))):
                t[j] = "ACGT"[r2.randint(4)]
            reads.append("".join(t))
        check(host, reads, int(r2.choice([0, 2, 6])), "long reads %d k=%d" % (i, k))
done["long-read cases"] = n3
print(done, "mismatching %d" % bad, flush=True)
sys.exit(0 if bad == 0 else 1)
