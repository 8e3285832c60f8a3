#!/usr/bin/env python3
"""bench.py — reads/sec pseudoaligned on synthetic 150 bp reads (BASELINE.json metric).

One step = one pass of the hot path (pa_map_count_batch_device: mapping with the class-count table fused in) over one batch of reads that is
already resident in HBM as 2-bit tiles. Workload (default "config3", BASELINE.json configs[2]): synthetic GENCODE-like
transcriptome (58 k genes -> ~202 k transcripts, seed 7), K = 24, error-free 150 bp reads (seed 2); --steps x --batch
reads per GPU (defaults 10 x 10 M = the config's 100 M reads at N = 1). Multi-GPU: one process per GPU, reads sharded
by rank (weak scaling: every rank maps its own --steps x --batch reads), index replicated, one RCCL all-reduce of the
class-count table at the end of the timed region.

The oracle (oracle/pa_oracle.c, a CPU port of the reference path) is used here only as (1) the parity checker of a
sample and (2) the `cpu_baseline` leg; it is never part of the measured GPU path.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time
from pathlib import Path  # noqa: E402

ROOT = Path(__file__).resolve().parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

WORKLOADS = {
    # name: (genes, transcripts, txome seed, k, read_len, read seed, substitution ppm)
    "config3": dict(genes=58000, transcripts=203000, txome_seed=7, k=24, read_len=150, read_seed=2, ppm=0, batch=100_000_000,
                    desc="~202k-transcript synthetic GENCODE-scale index (K=24), error-free 150bp reads"),
    "config5": dict(genes=58000, transcripts=203000, txome_seed=7, k=31, read_len=150, read_seed=4, ppm=10000, batch=100_000_000,
                    desc="same transcriptome at K=31, 150bp reads with 1% substitutions"),
    "config3k64": dict(genes=58000, transcripts=203000, txome_seed=7, k=64, read_len=150, read_seed=2, ppm=0, batch=100_000_000,
                       desc="config 3 rebuilt at K=64 (two-word k-mers, the other k of the reference's CLI), error-free 150bp reads"),
    "config2": dict(fasta=str(ROOT / "tests" / "golden" / "gencode_small.fa"), k=24, read_len=100, read_seed=1, ppm=0, batch=10_000_000,
                    desc="gencode_small (1832 transcripts) index (K=24), error-free 100bp reads"),
}


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def algorithmic_bytes_per_read(ctr: dict, read_len: int, k: int) -> float:
    """B = ceil(2L/8)+4 + p(8+ceil(2K/8)+12) + n(12+1+4+4) + ceil(2c/8) + 8n + 4E + 12 + 4r per read,
    with p/n/c/E/r = per-read averages of the oracle's counters on a sample of the same reads."""
    reads = max(ctr["reads"], 1)
    p = ctr["probes"] / reads
    n = ctr["node_visits"] / reads
    c = ctr["bases_compared"] / reads
    e = ctr["class_sizes"] / reads
    r = ctr["result_sizes"] / reads
    return (math.ceil(2 * read_len / 8) + 4 + p * (8 + math.ceil(2 * k / 8) + 12) + n * 21 + 2 * c / 8 + 8 * n + 4 * e + 12 + 4 * r)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0, help="reads per step per GPU (default: the batch BASELINE.json quotes for the workload: "
                    "100 M reads for configs 3/5, 10 M for config 2)")
    ap.add_argument("--workload", default="config3", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target duration of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the host-to-host leg (pinned tiles -> records on the host), an extra report at N=1")
    ap.add_argument("--separate-count", action="store_true", help="class counts in their own kernel instead of fused into the map kernel")
    ap.add_argument("--index-cache", default="", help="optional path to save/load the host index container")
    ap.add_argument("--cpu-build", action="store_true", help="build the index with the CPU builder instead of the GPU builder")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        log("warning: WORLD_SIZE %d != --gpus %d; using WORLD_SIZE" % (world, args.gpus))
    n_gpus = world

    import numpy as np
    import torch
    import helpers
    pa = helpers.pa

    if not torch.cuda.is_available() or pa.lib().pa_device_count() < 1:
        raise SystemExit("bench.py needs a GPU (the product has no CPU fallback)")
    # rehearsal of the multi-rank path on a box with fewer GPUs than ranks: PA_BENCH_BACKEND=gloo lets several ranks share a
    # device (RCCL refuses that); the driver's runs use neither variable
    backend = os.environ.get("PA_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)

    def barrier():
        if dist is not None:
            dist.barrier()

    # the reduce of the count table goes through the PRODUCT's collective (pa_counts_allreduce: RCCL over xGMI, communicator
    # owned by the library; the 128-byte id travels over torch.distributed's store, which is only the bootstrap here).
    # torch.distributed.all_reduce (the same RCCL underneath) is the spare if the library cannot open a communicator.
    comm = None
    if world > 1 and backend == "nccl":
        try:
            box = [pa.Comm.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            comm = pa.Comm(local_rank, world, rank, box[0])
        except Exception as e:   # noqa: BLE001
            log("pa_comm_create failed (%r): reducing through torch.distributed" % (e,))
            comm = None

    wl = WORKLOADS[args.workload]
    k, read_len, ppm = wl["k"], wl["read_len"], wl["ppm"]
    t0 = time.time()
    if "fasta" in wl:
        txome = pa.Txome.from_fasta(wl["fasta"])
    else:
        txome = pa.Txome.synthesize(wl["genes"], wl["transcripts"], wl["txome_seed"])
    log("transcriptome: %d transcripts (%.1f s)" % (txome.num_transcripts, time.time() - t0))

    # ---- index: built on this rank's GPU (csrc/index_build.hip: the same index, array for array, as the CPU builder gives; every rank
    # builds its own copy in well under a second, so nothing is handed over between ranks). --cpu-build takes the CPU builder.
    t0 = time.time()
    host, how = None, "GPU builder"
    if args.index_cache and os.path.exists(args.index_cache):
        host, how = pa.HostIndex.load(args.index_cache), "cache file"
    else:
        how = "CPU builder" if args.cpu_build else how
        host = pa.HostIndex.from_txome(txome, k, 0) if args.cpu_build else pa.HostIndex.from_txome_device(txome, k, local_rank)
        if args.index_cache and rank == 0:
            host.save(args.index_cache + ".tmp")
            os.replace(args.index_cache + ".tmp", args.index_cache)
    t_build = time.time() - t0
    log("host index ready (%.1f s, %s)" % (t_build, how))
    t0 = time.time()
    aligner = pa.Pseudoaligner(host, local_rank)
    st = aligner.stats()
    t_create = time.time() - t0
    log("device index: %d k-mers, %d nodes, %d classes, %.2f GB in HBM (%.1f s)" %
        (st.num_kmers, st.num_nodes, st.num_classes, st.bytes_total / 1e9, t_create))
    barrier()

    # ---- resident inputs: distinct batches of packed reads in HBM, generated on the device ----
    B, K, W = args.batch or wl["batch"], args.steps, args.warmup
    wpr = pa.lib().pa_words_per_read(read_len)
    # distinct batches resident in HBM, used in rotation (consecutive steps never see the same reads; one batch is 4.4 GB of tiles,
    # far beyond every cache). THREE, as a streaming pipeline holds them (one being filled, one being mapped, one being drained);
    # holding more only grows the process's HBM footprint, and beyond ~60 GB allocated every launch gets slower on this chip,
    # whatever the extra memory holds (same box: 2 / 3 / 6 / 12 resident batches 9.85 / 9.93 / 9.97 / 10.27 ms per 100 M reads;
    # 3 batches + an untouched 40 / 100 GB tensor: 9.71 / 10.41 ms). PA_BENCH_BATCHES overrides.
    n_batches = max(2, min(K + W, int(os.environ.get("PA_BENCH_BATCHES", "3")), int(64e9 // (B * (wpr * 8 + 4)))))
    tile_words = pa.lib().pa_tiles_words(B, wpr)
    stream = torch.cuda.current_stream().cuda_stream
    tiles = [torch.empty(tile_words, dtype=torch.int64, device=dev) for _ in range(n_batches)]
    lens = [torch.empty(B, dtype=torch.int32, device=dev) for _ in range(n_batches)]
    # global read index: rank r owns reads [r*(K+W)*B, (r+1)*(K+W)*B) of one global stream (rank-count independent)
    for b in range(n_batches):
        first = (rank * (K + W) + b) * B
        txome.simulate_device(read_len, wl["read_seed"], B, tiles[b].data_ptr(), lens[b].data_ptr(), ppm, first, wpr, local_rank, stream)
    arena_cap = aligner.arena_hint(B)
    results = torch.empty(B * 4, dtype=torch.int32, device=dev)
    arena = torch.empty(arena_cap, dtype=torch.int32, device=dev)
    colour = torch.empty(B, dtype=torch.int32, device=dev)
    counts = torch.zeros(aligner.counts_len(), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()

    ev = [pa._ffi.vp() for _ in range(2 * max(K, 1))]
    import ctypes as C
    for e in ev:
        pa.check(pa.lib().pa_event_create(C.byref(e)))

    def step(i: int, timed_idx: int = -1):
        nonlocal arena, arena_cap
        b = i % n_batches
        if timed_idx >= 0:
            pa.check(pa.lib().pa_event_record(ev[2 * timed_idx], stream or None))
        if args.separate_count:
            aligner.map_batch_device(tiles[b].data_ptr(), lens[b].data_ptr(), B, wpr, results.data_ptr(), arena.data_ptr(), arena_cap,
                                     2, colour.data_ptr(), stream)
        else:   # class-count table fused into the mapping kernel
            aligner.map_count_batch_device(tiles[b].data_ptr(), lens[b].data_ptr(), B, wpr, results.data_ptr(), arena.data_ptr(),
                                           arena_cap, counts.data_ptr(), 2, stream)
        if timed_idx >= 0:
            pa.check(pa.lib().pa_event_record(ev[2 * timed_idx + 1], stream or None))
        if args.separate_count:
            aligner.counts_accumulate_device(results.data_ptr(), arena.data_ptr(), colour.data_ptr(), B, counts.data_ptr(), stream)
        try:
            return aligner.map_finish(stream)
        except pa.PaError as e:
            if e.code != pa._ffi.PA_ERR_ARENA_FULL:
                raise
            raise SystemExit("arena too small for this workload: %s" % e)

    for i in range(W):
        step(i)
    # the warm-up includes the reduce: the first collective on a fresh RCCL communicator sets up its channels over xGMI, which
    # takes longer than the ten timed steps together
    if comm is not None:
        aligner.counts_allreduce(counts.data_ptr(), comm, stream)
    elif dist is not None:
        dist.all_reduce(counts)
    torch.cuda.synchronize()
    counts.zero_()
    torch.cuda.synchronize()
    barrier()
    t_start = time.perf_counter()
    for i in range(K):
        used, _ = step(W + i, i)
    if comm is not None:
        aligner.counts_allreduce(counts.data_ptr(), comm, stream)   # RCCL reduce of the eq-class count table over xGMI (product ABI)
    elif dist is not None:
        dist.all_reduce(counts)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t_start
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    kernel_ms = []
    for i in range(K):
        ms = C.c_float()
        pa.check(pa.lib().pa_event_elapsed_ms(ev[2 * i], ev[2 * i + 1], C.byref(ms)))
        kernel_ms.append(ms.value)
    kernel_avg_ms = sum(kernel_ms) / max(len(kernel_ms), 1)

    total_reads = K * B * n_gpus
    value = total_reads / elapsed
    counts_host = counts.cpu().numpy()
    assert os.environ.get("PA_MAP_ABLATE") or int(counts_host.sum()) == total_reads, "count table does not add up: %d vs %d" % (int(counts_host.sum()), total_reads)

    out = {
        "metric": "reads/sec pseudoaligned (whole node) on synthetic 150bp reads",
        "value": value, "unit": "reads/s", "n_gpus": n_gpus, "steps": K, "warmup": W,
        "ms_per_step": 1000.0 * elapsed / max(K, 1), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {"workload": "%s: %s" % (args.workload, wl["desc"]), "reads_per_step_per_gpu": B, "read_len": read_len, "k": k,
                   "transcripts": txome.num_transcripts, "kmers": int(st.num_kmers), "index_bytes": int(st.bytes_total), "index_build_s": round(t_build, 3), "index_upload_s": round(t_create, 3),
                   "parallelism": "reads sharded over %d GPU(s), index replicated, RCCL all-reduce of class counts (%s)" %
                                  (n_gpus, "pa_counts_allreduce" if comm is not None else "torch.distributed" if world > 1 else "one GPU: no reduce")},
    }

    # ---- SURVEY §8d's wall-clock leg (extra keys, never `value`): the same batch from PINNED HOST tiles to per-read records +
    # count table back on the host. Chunks alternate between two streams of the one index handle: H2D of chunk i+1 and D2H
    # of chunk i-1 overlap the kernel of chunk i; the link (PCIe Gen5 x16, 63 GB/s per direction), not the kernel, bounds it.
    if n_gpus == 1 and not args.no_e2e:
        try:
            b = W % n_batches
            chunk = min(B, 20_000_000)
            n_chunks = (B + chunk - 1) // chunk
            words = pa.lib().pa_tiles_words
            h_tiles = torch.empty(tile_words, dtype=torch.int64, pin_memory=True)
            h_lens = torch.empty(B, dtype=torch.int32, pin_memory=True)
            h_results = torch.empty(B * 4, dtype=torch.int32, pin_memory=True)
            h_counts = torch.empty(aligner.counts_len(), dtype=torch.int64, pin_memory=True)
            h_tiles.copy_(tiles[b]); h_lens.copy_(lens[b])
            NS = int(os.environ.get("PA_E2E_STREAMS", "3"))   # chunks in flight (one stream + one set of staging buffers each)
            chunk = min(B, int(os.environ.get("PA_E2E_CHUNK", str(chunk))))
            n_chunks = (B + chunk - 1) // chunk
            streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
            stage = [dict(tiles=torch.empty(words(chunk, wpr), dtype=torch.int64, device=dev), lens=torch.empty(chunk, dtype=torch.int32, device=dev),
                          res=torch.empty(chunk * 4, dtype=torch.int32, device=dev), arena=torch.empty(aligner.arena_hint(chunk), dtype=torch.int32, device=dev),
                          busy=False) for _ in range(NS)]
            e_counts = torch.zeros(aligner.counts_len(), dtype=torch.int64, device=dev)
            def host_to_host():
                arena_ids = 0
                for st in stage:
                    st["busy"] = False
                e_counts.zero_()
                torch.cuda.synchronize()
                t_e2e = time.perf_counter()
                for c in range(n_chunks):
                    st, S = stage[c % NS], streams[c % NS]
                    lo, nn = c * chunk, min(chunk, B - c * chunk)          # chunk is a multiple of 64: tile aligned
                    if st["busy"]:
                        arena_ids += aligner.map_finish(S.cuda_stream)[0]   # the stream's previous chunk is done (its records are on the host)
                    with torch.cuda.stream(S):
                        st["tiles"][: words(nn, wpr)].copy_(h_tiles[(lo // 64) * wpr * 64: (lo // 64) * wpr * 64 + words(nn, wpr)], non_blocking=True)
                        st["lens"][:nn].copy_(h_lens[lo: lo + nn], non_blocking=True)
                        aligner.map_count_batch_device(st["tiles"].data_ptr(), st["lens"].data_ptr(), nn, wpr, st["res"].data_ptr(), st["arena"].data_ptr(),
                                                       st["arena"].numel(), e_counts.data_ptr(), 2, S.cuda_stream)
                        h_results[lo * 4: (lo + nn) * 4].copy_(st["res"][: nn * 4], non_blocking=True)
                    st["busy"] = True
                for i in range(NS):
                    if stage[i]["busy"]:
                        arena_ids += aligner.map_finish(streams[i].cuda_stream)[0]
                h_counts.copy_(e_counts)
                torch.cuda.synchronize()
                return time.perf_counter() - t_e2e, arena_ids
            host_to_host()                       # warm-up: the per-stream launch contexts (scratch rows, count replicas) are created on first use
            e2e_s, arena_ids = host_to_host()
            assert os.environ.get("PA_MAP_ABLATE") or int(h_counts.sum()) == B
            h2d_bytes = B * (wpr * 8 + 4)
            out["e2e_reads_per_s"] = B / e2e_s
            out["e2e_pcie_frac"] = h2d_bytes / e2e_s / 63e9
            out["e2e"] = {"ms": 1000.0 * e2e_s, "chunks": n_chunks, "reads_per_chunk": chunk, "streams": NS, "h2d_bytes": h2d_bytes,
                          "d2h_bytes": B * 16 + 8 * aligner.counts_len(), "novel_class_ids_left_on_device": int(arena_ids),
                          "what": "pinned host 2-bit tiles -> H2D || kernel || D2H of the 16-byte records on several streams of one index handle "
                                  "-> records + count table on the host; link = PCIe Gen5 x16, 63 GB/s per direction"}
            del h_tiles, h_lens, h_results, stage
        except Exception as e:   # the leg is a report, not the benchmark: never lose the bench line over it
            log("e2e leg failed: %r" % (e,))
            out["e2e_error"] = repr(e)

    if rank == 0:
        # ---- checker + CPU baseline (oracle = C port of the reference path), outside the timed region ----
        t0 = time.time()
        oracle = helpers.Oracle(host)
        log("oracle index built (%.1f s)" % (time.time() - t0))
        ncpu = usable_cpus()
        sample_n = 200_000
        first = (rank * (K + W) + (W % n_batches)) * B   # the first timed batch
        s_tiles, s_lens = txome.simulate_host(read_len, wl["read_seed"], sample_n, ppm, first, wpr)
        o_res, o_coff, o_ids, ctr = oracle.map_tiles(s_tiles, s_lens, wpr, 2, ncpu)
        # parity of the sample: GPU results of the same reads (batch W % n_batches is still resident)
        b = W % n_batches
        aligner.map_batch_device(tiles[b].data_ptr(), lens[b].data_ptr(), sample_n, wpr, results.data_ptr(), arena.data_ptr(), arena_cap,
                                 2, 0, stream)
        used, _ = aligner.map_finish(stream)
        g_res = results[: sample_n * 4].cpu().numpy().view(pa.RESULT_DTYPE)
        g_arena = arena[: max(used, 1)].cpu().numpy().view(np.uint32)
        if not os.environ.get("PA_MAP_ABLATE"):   # (an ablated kernel leaves records unwritten: nothing to compare)
            g_coff, g_ids = pa.gather_classes(g_res, g_arena, host)
            helpers.assert_same_as_oracle(g_res, g_coff, g_ids, o_res, o_coff, o_ids, "bench sample")
            out["parity_sample"] = {"reads": sample_n, "bit_exact_vs_oracle": True}
        bytes_per_read = algorithmic_bytes_per_read(ctr, read_len, k)
        achieved = bytes_per_read * B / (kernel_avg_ms * 1e-3) / 1e9
        requests = None
        raw_traffic = None
        traffic = None   # HBM bytes per launch from committed rocprofv3 PMC passes of this same workload (bench.py cannot run PMC itself)
        try:
            pmc = json.load(open(ROOT / "profiles" / "latest_pmc.json"))["workloads"][args.workload]
            if pmc.get("reads_per_launch") == B:
                raw = (pmc["FETCH_SIZE_KB"] + pmc["WRITE_SIZE_KB"]) * 1024.0
                # gfx950's FETCH_SIZE tallies a coalesced stream at half its bytes (calibrated: profiles/r02_pmc_calibration.txt; random
                # lines, stores and atomics are exact): the other half of the streamed input (read tiles + lengths) is added back
                traffic = raw + 0.5 * (8.0 * wpr + 4.0) * B
                raw_traffic = raw
                # the same traffic as 64-byte requests per second, next to what tools/microbench/gather.hip measures on MI355X
                # for nothing but random 64-byte lines: the bound this access pattern actually runs into
                per_launch = raw / 64.0
                requests = {"per_read": per_launch / B, "per_s": per_launch / (kernel_avg_ms * 1e-3), "gather_hbm_per_s": 49e9,
                            "gather_mall_per_s": 57e9, "frac_of_gather_hbm": per_launch / (kernel_avg_ms * 1e-3) / 49e9,
                            "what": "(FETCH_SIZE + WRITE_SIZE) / 64 B of the committed PMC passes over this run's kernel time; ceilings: "
                                    "tools/microbench/gather.hip, every lane a different random line, HBM- and MALL-resident tables"}
        except (OSError, ValueError, KeyError):
            pass
        out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                           "traffic": traffic, "traffic_raw_counters": raw_traffic, "kernel": "pa_map_pool_kernel", "kernel_ms": kernel_avg_ms,
                           "kernel_ms_min": min(kernel_ms) if kernel_ms else None, "kernel_ms_max": max(kernel_ms) if kernel_ms else None,
                           "kernel_ms_steps": [round(x, 3) for x in kernel_ms],
                           "algorithmic_bytes_per_read": bytes_per_read, "reads_per_launch": B, "requests": requests}
        if n_gpus == 1 and not args.no_cpu_baseline:
            rate = sample_n / max(1e-9, _time_oracle(oracle, s_tiles, s_lens, wpr, ncpu))
            big_n = int(min(max(rate * args.cpu_seconds, sample_n), 40_000_000))
            b_tiles, b_lens = txome.simulate_host(read_len, wl["read_seed"], big_n, ppm, first, wpr)
            secs = _time_oracle(oracle, b_tiles, b_lens, wpr, ncpu)
            out["cpu_baseline"] = {"value": big_n / secs, "unit": "reads/s", "cores": ncpu, "kind": "port",
                                   "sample": "first %d reads of the first timed batch, oracle/pa_oracle.c on %d pthreads (%d CPUs visible, quota/affinity %d), %.1f s"
                                             % (big_n, ncpu, os.cpu_count() or 1, ncpu, secs)}
        print(json.dumps(out), flush=True)
    barrier()
    if dist is not None:
        dist.destroy_process_group()


def usable_cpus() -> int:
    """CPUs this process may actually use: visible CPUs, affinity mask and the cgroup CPU quota (a container that sees 256
    CPUs may be limited to 16 CPUs' worth of time; timing 256 threads there would mislabel the baseline)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())         # cgroup v1
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return n


def _time_oracle(oracle, tiles, lens, wpr, threads) -> float:
    oracle.map_tiles(tiles, lens, wpr, 2, threads)
    return oracle.last_seconds   # wall time of the mapping threads only (excludes the serial output assembly)


if __name__ == "__main__":
    main()
