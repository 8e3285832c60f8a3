#!/usr/bin/env python3
"""Host-to-host rate of pa_map_tiles_host on config 3: pinned host tiles -> compact records + count table on the host.
usage (GPU box): python tools/bench_e2e.py [--reads N] ; variants through PA_HB_IN / PA_HB_BACK / chunk / streams in a knobs build (PA_PRODUCT_SO)."""
import argparse, importlib, json, os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402  (before the product library: one HIP runtime)
import numpy as np
pa = importlib.import_module("rust-pseudoaligner_amd")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=100_000_000)
    ap.add_argument("--variants", default="1,1,2000000,4")     # in,back,chunk,streams ; ...
    args = ap.parse_args()
    tx = pa.Txome.synthesize(58000, 203000, 7)
    al = pa.Pseudoaligner(pa.HostIndex.from_txome_device(tx, 24, 0))
    n, L, wpr = args.reads, 150, 5
    dev = torch.device("cuda", 0)
    d_tiles = torch.empty(pa.lib().pa_tiles_words(n, wpr), dtype=torch.int64, device=dev)
    d_lens = torch.empty(n, dtype=torch.int32, device=dev)
    tx.simulate_device(L, 2, n, d_tiles.data_ptr(), d_lens.data_ptr(), 0, 0, wpr)
    h_tiles = torch.empty(d_tiles.numel(), dtype=torch.int64, pin_memory=True)
    h_tiles.copy_(d_tiles)
    del d_tiles, d_lens
    h_compact = torch.empty(n, dtype=torch.int64, pin_memory=True)
    h_packed = torch.empty(n // 2, dtype=torch.int32, pin_memory=True)
    h_counts = torch.empty(al.counts_len(), dtype=torch.int64, pin_memory=True)
    torch.cuda.synchronize()
    for var in args.variants.split(";"):
        i, b, chunk, ns = (int(x) for x in var.split(","))
        os.environ["PA_HB_IN"], os.environ["PA_HB_BACK"] = str(i), str(b)
        ts = []
        for _ in range(4):
            t0 = time.perf_counter()
            al.map_tiles_host(h_tiles.data_ptr(), n, wpr, h_compact.data_ptr(), h_packed.data_ptr(), h_packed.numel(), uniform_len=L, h_counts=h_counts.data_ptr(),
                              chunk_reads=chunk, n_streams=ns)
            ts.append(time.perf_counter() - t0)
        assert int(h_counts.sum()) == n
        print(json.dumps({"in": i, "back": b, "chunk": chunk, "streams": ns, "ms": [round(t * 1e3, 1) for t in ts], "best_Greads_per_s": n / min(ts[1:]) / 1e9,
                          "h2d_GBps": n * wpr * 8 / min(ts[1:]) / 1e9}), flush=True)


if __name__ == "__main__":
    main()
