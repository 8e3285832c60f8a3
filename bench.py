#!/usr/bin/env python3
"""bench.py — reads/sec pseudoaligned on synthetic 150 bp reads (BASELINE.json metric).

One step = one pass of the hot path (pa_map_count_batch_device: the mapping kernel followed by the class-count kernels) over one
batch of reads that is already resident in HBM as 2-bit tiles. Workload (default "config3", BASELINE.json configs[2]): synthetic
GENCODE-like transcriptome (58 k genes -> ~202 k transcripts, seed 7), K = 24, error-free 150 bp reads (seed 2); one step = the
config's 100 M reads at N = 1. Multi-GPU: one process per GPU, reads sharded by rank (weak scaling: every rank maps its own
--steps x --batch reads), index replicated, one RCCL all-reduce of the class-count table at the end of the timed region.

The default N = 1 run also times BASELINE.json's error-read configuration (config 5: K = 31, 1 % substitutions; the same
transcriptome, its own index) for a few steps and reports it under the extra key `config5` (never `value`).

The oracle (oracle/pa_oracle.c, a CPU port of the reference path) is used here only as (1) the parity checker of a sample and
(2) the `cpu_baseline` leg; it is never part of the measured GPU path.
"""
from __future__ import annotations

import argparse
import hashlib
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
    "config3r": dict(genes=58000, transcripts=203000, txome_seed=7, k=24, read_len=150, read_seed=2, ppm=0, batch=100_000_000, repeats=True,
                     desc="config 3 with real-graph structure: 40 repeat families at 10-15 % divergence + 10 at 3-6 % (300-base elements in the last exon of 13 % of the genes) and 200 low-complexity tracts"),
    "config2": dict(fasta=str(ROOT / "tests" / "golden" / "gencode_small.fa"), k=24, read_len=100, read_seed=1, ppm=0, batch=10_000_000,
                    desc="gencode_small (1832 transcripts) index (K=24), error-free 100bp reads"),
}

# the device sources whose text decides what the PMC counters of profiles/latest_pmc.json were measured on
KERNEL_SOURCES = ["map_pool.hip", "map_pool_kernel.inc", "lane_steps.hpp", "lane_steps_body.hpp", "device_layout.hpp", "count_sort.hip", "kernels.hpp", "dict_slots.hpp", "resolve.hip", "kernel_utils.hpp",
                  "device_index.hip", "device_flatten.cpp", "index_fill.hip"]


def kernel_source_sha256() -> str:
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        h.update((ROOT / "rust-pseudoaligner_amd" / "csrc" / name).read_bytes())
    h.update((ROOT / "rust-pseudoaligner_amd" / "_build.py").read_bytes())   # (the compiler flags of the kernels)
    return h.hexdigest()


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


class Run:
    """One workload on this rank's GPU: index, resident batches, the timed steps, the checker."""

    def __init__(self, env, name, batch, index_cache="", cpu_build=False):
        self.env, self.name, self.wl = env, name, WORKLOADS[name]
        pa, torch = env["pa"], env["torch"]
        wl = self.wl
        t0 = time.time()
        self.txome = (pa.Txome.from_fasta(wl["fasta"]) if "fasta" in wl else
                      pa.Txome.synthesize_repeats(wl["genes"], wl["transcripts"], wl["txome_seed"]) if wl.get("repeats") else
                      pa.Txome.synthesize(wl["genes"], wl["transcripts"], wl["txome_seed"]))
        log("%s transcriptome: %d transcripts (%.1f s)" % (name, self.txome.num_transcripts, time.time() - t0))
        # index: built on this rank's GPU (csrc/index_build.hip: the same index, array for array, as the CPU builder gives; every rank builds
        # its own copy in well under a second, so nothing is handed over between ranks). --cpu-build takes the CPU builder.
        t0 = time.time()
        how = "GPU builder"
        if index_cache and os.path.exists(index_cache):
            self.host, how = pa.HostIndex.load(index_cache), "cache file"
        else:
            how = "CPU builder" if cpu_build else how
            self.host = pa.HostIndex.from_txome(self.txome, wl["k"], 0) if cpu_build else pa.HostIndex.from_txome_device(self.txome, wl["k"], env["local_rank"])
            if index_cache and env["rank"] == 0:
                self.host.save(index_cache + ".tmp")
                os.replace(index_cache + ".tmp", index_cache)
        self.t_build = time.time() - t0
        log("host index ready (%.1f s, %s)" % (self.t_build, how))
        t0 = time.time()
        self.aligner = pa.Pseudoaligner(self.host, env["local_rank"])
        self.aligner.set_timing(True)
        self.st = self.aligner.stats()
        self.t_create = time.time() - t0
        log("device index: %d k-mers, %d nodes, %d classes, %.2f GB in HBM (%.1f s)" %
            (self.st.num_kmers, self.st.num_nodes, self.st.num_classes, self.st.bytes_total / 1e9, self.t_create))
        self.B = batch or wl["batch"]
        self.wpr = pa.lib().pa_words_per_read(wl["read_len"])

    def make_batches(self, K, W, buffers=None):
        """distinct batches of packed reads resident in HBM, generated on the device, used in rotation (consecutive steps never see the
        same reads; one batch is 4.4 GB of tiles, far beyond every cache). THREE, as a streaming pipeline holds them (one being filled,
        one being mapped, one being drained); holding more only grows the process's HBM footprint, and beyond ~60 GB allocated every
        launch gets slower on this chip, whatever the extra memory holds. PA_BENCH_BATCHES overrides."""
        env, wl, B, wpr = self.env, self.wl, self.B, self.wpr
        pa, torch, dev = env["pa"], env["torch"], env["dev"]
        self.K, self.W = K, W
        self.n_batches = max(2, min(K + W, int(os.environ.get("PA_BENCH_BATCHES", "3")), int(64e9 // (B * (wpr * 8 + 4)))))
        self.tile_words = pa.lib().pa_tiles_words(B, wpr)
        self.stream = torch.cuda.current_stream().cuda_stream
        if buffers is None:
            buffers = dict(tiles=[torch.empty(self.tile_words, dtype=torch.int64, device=dev) for _ in range(self.n_batches)],
                           lens=[torch.empty(B, dtype=torch.int32, device=dev) for _ in range(self.n_batches)],
                           results=torch.empty(B * 4, dtype=torch.int32, device=dev), colour=torch.empty(B, dtype=torch.int32, device=dev))
        self.buffers = buffers
        self.tiles, self.lens, self.results, self.colour = buffers["tiles"], buffers["lens"], buffers["results"], buffers["colour"]
        # global read index: rank r owns reads [r*(K+W)*B, (r+1)*(K+W)*B) of one global stream (rank-count independent)
        for b in range(self.n_batches):
            first = (env["rank"] * (K + W) + b) * B
            self.txome.simulate_device(wl["read_len"], wl["read_seed"], B, self.tiles[b].data_ptr(), self.lens[b].data_ptr(), wl["ppm"], first, wpr,
                                       env["local_rank"], self.stream)
        self.arena_cap = self.aligner.arena_hint(B)
        if "arena" not in buffers or buffers["arena"].numel() < self.arena_cap:
            buffers["arena"] = torch.empty(self.arena_cap, dtype=torch.int32, device=dev)
        self.arena = buffers["arena"]
        self.counts = torch.zeros(self.aligner.counts_len(), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()

    def step(self, i, separate_count=False):
        """returns (device ms of the whole step's launches, (ms of the mapping kernel, of the resolve kernel, of the count kernels))"""
        pa, a = self.env["pa"], self.aligner
        b = i % self.n_batches
        import ctypes as C
        pa.check(pa.lib().pa_event_record(self.ev[0], self.stream or None))
        if separate_count:
            a.map_batch_device(self.tiles[b].data_ptr(), self.lens[b].data_ptr(), self.B, self.wpr, self.results.data_ptr(), self.arena.data_ptr(), self.arena_cap,
                               2, self.colour.data_ptr(), self.stream)
            a.counts_accumulate_device(self.results.data_ptr(), self.arena.data_ptr(), self.colour.data_ptr(), self.B, self.counts.data_ptr(), self.stream)
        else:   # class counts in the same call: mapping kernel, then the count kernels over its key streams
            a.map_count_batch_device(self.tiles[b].data_ptr(), self.lens[b].data_ptr(), self.B, self.wpr, self.results.data_ptr(), self.arena.data_ptr(),
                                     self.arena_cap, self.counts.data_ptr(), 2, self.stream)
        pa.check(pa.lib().pa_event_record(self.ev[1], self.stream or None))
        try:
            a.map_finish(self.stream)
        except pa.PaError as e:
            if e.code != pa._ffi.PA_ERR_ARENA_FULL:
                raise
            raise SystemExit("arena too small for this workload: %s" % e)
        ms = C.c_float()
        pa.check(pa.lib().pa_event_elapsed_ms(self.ev[0], self.ev[1], C.byref(ms)))
        return ms.value, a.map_stage_ms(self.stream)

    def timed(self, K, W, reduce_fn=None, barrier=lambda: None, separate_count=False):
        """W warm-up steps, then exactly K timed steps bracketed by barrier + synchronize; returns (elapsed s, [step device ms], [map kernel ms])"""
        pa, torch = self.env["pa"], self.env["torch"]
        import ctypes as C
        self.ev = [pa._ffi.vp(), pa._ffi.vp()]
        for e in self.ev:
            pa.check(pa.lib().pa_event_create(C.byref(e)))
        for i in range(W):
            self.step(i, separate_count)
        # the warm-up includes the reduce: the first collective on a fresh RCCL communicator sets up its channels over xGMI, which
        # takes longer than the timed steps together
        if reduce_fn:
            reduce_fn(self.counts)
        torch.cuda.synchronize()
        self.counts.zero_()
        torch.cuda.synchronize()
        barrier()
        t_start = time.perf_counter()
        step_ms, map_ms = [], []
        for i in range(K):
            s_ms, m_ms = self.step(W + i, separate_count)
            step_ms.append(s_ms)
            map_ms.append(m_ms)
        if reduce_fn:
            reduce_fn(self.counts)   # RCCL reduce of the eq-class count table over xGMI (product ABI)
        torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t_start
        for e in self.ev:
            pa.check(pa.lib().pa_event_destroy(e))
        return elapsed, step_ms, map_ms

    def check_sample(self, sample_n, ncpu):
        """GPU results of the first reads of the first timed batch against the oracle on the host-simulated twins; returns the oracle,
        its counters and the sample (for the CPU baseline)"""
        env, wl = self.env, self.wl
        pa, helpers, np = env["pa"], env["helpers"], env["np"]
        t0 = time.time()
        oracle = helpers.Oracle(self.host)
        log("oracle index built (%.1f s)" % (time.time() - t0))
        b = self.W % self.n_batches
        first = (env["rank"] * (self.K + self.W) + b) * self.B   # the first timed batch (still resident)
        s_tiles, s_lens = self.txome.simulate_host(wl["read_len"], wl["read_seed"], sample_n, wl["ppm"], first, self.wpr)
        o_res, o_coff, o_ids, ctr = oracle.map_tiles(s_tiles, s_lens, self.wpr, 2, ncpu)
        self.aligner.map_batch_device(self.tiles[b].data_ptr(), self.lens[b].data_ptr(), sample_n, self.wpr, self.results.data_ptr(), self.arena.data_ptr(),
                                      self.arena_cap, 2, 0, self.stream)
        used, _ = self.aligner.map_finish(self.stream)
        g_res = self.results[: sample_n * 4].cpu().numpy().view(pa.RESULT_DTYPE)
        g_arena = self.arena[: max(used, 1)].cpu().numpy().view(np.uint32)
        if not os.environ.get("PA_MAP_ABLATE"):   # (A/B builds only: an ablated kernel leaves records unwritten, nothing to compare)
            g_coff, g_ids = pa.gather_classes(g_res, g_arena, self.host)
            helpers.assert_same_as_oracle(g_res, g_coff, g_ids, o_res, o_coff, o_ids, "bench sample %s" % self.name)
        return oracle, ctr, first


def roofline_of(run, ctr, stage_ms, step_ms):
    """stage_ms: per timed step (mapping kernel, resolve kernel, count kernels) ms. The roofline is priced over the MAPPING STAGE =
    pa_map_pool_kernel + pa_resolve_kernel: the second writes the records of the reads whose class is looked up by content (about 3 %
    of them), so the batch is not mapped before it has run."""
    wl, B = run.wl, run.B
    bytes_per_read = algorithmic_bytes_per_read(ctr, wl["read_len"], wl["k"])
    n = max(len(stage_ms), 1)
    map_ms = [m[0] + m[1] for m in stage_ms]
    pool_avg_ms, resolve_avg_ms, count_avg_ms = (sum(m[i] for m in stage_ms) / n for i in range(3))
    kernel_avg_ms = sum(map_ms) / n
    achieved = bytes_per_read * B / (kernel_avg_ms * 1e-3) / 1e9
    requests = raw_traffic = traffic = miss_block_bytes = box_step_ms = None
    traffic_note = None
    # HBM bytes per step from committed rocprofv3 PMC passes of this same workload (bench.py cannot run PMC itself). The file names the
    # hash of the kernel sources it was measured on: a kernel change without a new PMC pass reports no traffic instead of stale bytes.
    try:
        doc = json.load(open(ROOT / "profiles" / "latest_pmc.json"))
        pmc = doc["workloads"].get(run.name)
        if pmc is None:
            traffic_note = "no committed PMC pass for workload %s (profiles/latest_pmc.json holds %s)" % (run.name, ", ".join(sorted(doc["workloads"])))
        elif doc.get("kernel_source_sha256") != kernel_source_sha256():
            traffic_note = "profiles/latest_pmc.json was measured on other kernel sources (sha256 %s..., built: %s...): no traffic figure" % (
                str(doc.get("kernel_source_sha256"))[:12], kernel_source_sha256()[:12])
        elif pmc.get("reads_per_launch") != B:
            traffic_note = "profiles/latest_pmc.json holds %s reads per launch, this run %d" % (pmc.get("reads_per_launch"), B)
        else:
            raw = (pmc["FETCH_SIZE_KB"] + pmc["WRITE_SIZE_KB"]) * 1024.0
            # gfx950 fetches 128 bytes for EVERY memory-side read request — streams, random 64-byte lines, 16-byte dictionary slots, chain
            # blocks (profiles/r05_pmc_calibration.txt: TCC_EA0_RDREQ_128B == TCC_EA0_RDREQ in every pattern) — and FETCH_SIZE tallies
            # each at 64 bytes. The bytes that crossed the memory interface are the requests by size (or 2 x FETCH_SIZE) + WRITE_SIZE
            # (exact at 32-byte sectors in the same calibration).
            rd_bytes = pmc.get("READ_REQUEST_BYTES") or 2.0 * pmc["FETCH_SIZE_KB"] * 1024.0
            traffic = rd_bytes + pmc["WRITE_SIZE_KB"] * 1024.0
            raw_traffic = raw
            box_step_ms = pmc.get("profiled_box_step_ms")
            miss_block_bytes = 128.0 * pmc["map_kernel_l2_misses"] / B if pmc.get("map_kernel_l2_misses") else None
            per_launch = raw / 64.0
            step_avg = sum(step_ms) / max(len(step_ms), 1)
            requests = {"per_read": per_launch / B, "per_s": per_launch / (step_avg * 1e-3), "gather_hbm_per_s": 49e9, "gather_mall_per_s": 57e9,
                        "frac_of_gather_hbm": per_launch / (step_avg * 1e-3) / 49e9,
                        # L2 misses of the map kernel (TCC_MISS of the same PMC passes) over this run's map-kernel time, against the rate at
                        # which the chip serves lanes that each fetch from a random 128-byte block of a table beyond the L2
                        # (tools/microbench/gather_multi.hip, profiles/r03_gather_multi.txt: 55-57 G blocks/s, MALL-sized table)
                        "map_kernel_l2_misses_per_read": (pmc["map_kernel_l2_misses"] / B) if pmc.get("map_kernel_l2_misses") else None,
                        "map_kernel_l2_misses_per_s": (pmc["map_kernel_l2_misses"] / (kernel_avg_ms * 1e-3)) if pmc.get("map_kernel_l2_misses") else None,
                        "random_block_ceiling_per_s": 55e9,
                        "what": "(FETCH_SIZE + WRITE_SIZE) / 64 B of the committed PMC passes (all kernels of a step) over this run's step time; ceilings: "
                                "tools/microbench/gather.hip, every lane a different random line, HBM- and MALL-resident tables"}
    except (OSError, ValueError, KeyError) as e:
        traffic_note = "profiles/latest_pmc.json unusable: %r" % (e,)
    step_avg = sum(step_ms) / max(len(step_ms), 1)
    return {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
            "traffic": traffic, "traffic_raw_counters": raw_traffic, "traffic_note": traffic_note,
            # the counters' bytes of a step: memory-side requests LEAVING THE L2 (x 128 B) + WRITE_SIZE. The guide says these counters also count
            # requests the Infinity Cache answers (the chain blocks, 0.38 GB, mostly live there), so this is traffic out of the L2, an upper bound of
            # HBM traffic — and it was measured on the builder's box (profiles/latest_pmc.json), not on this one
            "traffic_what": "bytes of memory-side requests leaving the L2 per launch (Infinity-Cache hits included), rocprofv3 --pmc on the profiled box",
            "traffic_gbps": (traffic / (step_avg * 1e-3) / 1e9) if traffic else None,                      # ... over THIS run's step time
            "traffic_frac_of_peak": (traffic / (step_avg * 1e-3) / 1e9 / 8000.0) if traffic else None,
            # the same bytes over the step time of the box they were measured on, against the 6.29 TB/s a streaming copy sustains (MI355X_MICROARCH.md)
            "l2_outbound_frac_of_6_3_on_profiled_box": (traffic / (box_step_ms * 1e-3) / 1e9 / 6290.0) if traffic and box_step_ms else None,
            "profiled_box_step_ms": box_step_ms,
            "l2_miss_block_bytes_per_read": miss_block_bytes,
            "kernel": "pa_map_pool_kernel + pa_resolve_kernel", "kernel_ms": kernel_avg_ms, "map_pool_kernel_ms": pool_avg_ms, "resolve_kernel_ms": resolve_avg_ms,
            "kernel_ms_min": min(map_ms) if map_ms else None, "kernel_ms_max": max(map_ms) if map_ms else None,
            "kernel_ms_steps": [round(x, 3) for x in map_ms],
            "step_device_ms": step_avg, "count_kernels_ms": count_avg_ms,
            "timing": "HIP events on the launch stream, recorded by the library between the kernels of a launch (pa_map_stage_ms): kernel_ms = "
                      "pa_map_pool_kernel + pa_resolve_kernel (the mapping stage), count_kernels_ms = the pa_keys_* class-count kernels, "
                      "step_device_ms the whole pa_map_count_batch_device call",
            "algorithmic_bytes_per_read": bytes_per_read, "reads_per_launch": B, "requests": requests}


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
    ap.add_argument("--no-ingest", action="store_true", help="skip the FASTQ-text leg (pa_process_reads / pa_record_stream_* on a bounded sample), an extra report at N=1")
    ap.add_argument("--ingest-reads", type=int, default=8_000_000, help="reads of the FASTQ-text leg")
    ap.add_argument("--no-config5", action="store_true", help="skip the extra config-5 measurement of the default N=1 run")
    ap.add_argument("--separate-count", action="store_true", help="class counts from the stored records (pa_counts_accumulate_device) instead of the key streams")
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
            # the reduce of the count table is part of the product: a run that cannot open the library's communicator is not
            # the run BASELINE.json describes. PA_BENCH_ALLOW_TORCH_REDUCE=1 (diagnosis only) lets it go on through torch.distributed,
            # and the line then says so (rccl_ranks 0)
            if os.environ.get("PA_BENCH_ALLOW_TORCH_REDUCE") != "1":
                raise SystemExit("bench.py: pa_comm_create failed on rank %d of %d: %r (PA_BENCH_ALLOW_TORCH_REDUCE=1 reduces through torch.distributed instead)" % (rank, world, e))
            log("pa_comm_create failed (%r): reducing through torch.distributed" % (e,))
            comm = None
    rccl_ranks = comm.size if comm is not None else 0
    if world > 1 and backend == "nccl" and comm is not None:
        assert rccl_ranks == world, "the product's RCCL communicator spans %d ranks, the job %d" % (rccl_ranks, world)

    env = dict(pa=pa, torch=torch, helpers=helpers, np=np, rank=rank, local_rank=local_rank, dev=dev)
    wl = WORKLOADS[args.workload]
    run = Run(env, args.workload, args.batch, args.index_cache, args.cpu_build)
    barrier()
    K, W = args.steps, args.warmup
    run.make_batches(K, W)
    B = run.B
    stream = run.stream

    def reduce_counts(counts):
        if comm is not None:
            run.aligner.counts_allreduce(counts.data_ptr(), comm, stream)
        elif dist is not None:
            dist.all_reduce(counts)

    elapsed, step_ms, map_ms = run.timed(K, W, reduce_counts if world > 1 else None, barrier, args.separate_count)
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # every rank's own kernel times travel to rank 0 (north_star: reads/s AND HBM GB/s at every N): mean mapping-kernel, resolve,
    # count and whole-step device time per rank
    n_t = max(len(map_ms), 1)
    mine = [sum(m[0] for m in map_ms) / n_t, sum(m[1] for m in map_ms) / n_t, sum(m[2] for m in map_ms) / n_t, sum(step_ms) / max(len(step_ms), 1)]
    per_rank = [mine]
    if dist is not None:
        t = torch.tensor(mine, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")   # (the gloo rehearsal gathers on the host)
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        per_rank = [g.tolist() for g in gathered]

    total_reads = K * B * n_gpus
    value = total_reads / elapsed
    counts_host = run.counts.cpu().numpy()
    assert os.environ.get("PA_MAP_ABLATE") or int(counts_host.sum()) == total_reads, "count table does not add up: %d vs %d" % (int(counts_host.sum()), total_reads)

    st, k, read_len = run.st, wl["k"], wl["read_len"]
    out = {
        "metric": "reads/sec pseudoaligned (whole node) on synthetic 150bp reads",
        "value": value, "unit": "reads/s", "n_gpus": n_gpus, "steps": K, "warmup": W,
        "ms_per_step": 1000.0 * elapsed / max(K, 1), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic", "rccl_ranks": rccl_ranks,
        "config": {"workload": "%s: %s" % (args.workload, wl["desc"]), "reads_per_step_per_gpu": B, "read_len": read_len, "k": k,
                   "transcripts": run.txome.num_transcripts, "kmers": int(st.num_kmers), "index_bytes": int(st.bytes_total), "index_build_s": round(run.t_build, 3),
                   "index_upload_s": round(run.t_create, 3),
                   "parallelism": "reads sharded over %d GPU(s), index replicated, RCCL all-reduce of class counts (%s)" %
                                  (n_gpus, "pa_counts_allreduce, %d RCCL ranks" % rccl_ranks if comm is not None else "torch.distributed" if world > 1 else "one GPU: no reduce")},
    }

    # ---- SURVEY §8d's wall-clock leg (extra keys, never `value`): the same batch from PINNED HOST tiles to per-read records +
    # count table back on the host. Chunks alternate between streams of the one index handle: H2D of chunk i+1 and D2H
    # of chunk i-1 overlap the kernel of chunk i; the link (PCIe Gen5 x16, 63 GB/s per direction), not the kernel, bounds it.
    if n_gpus == 1 and not args.no_e2e:
        try:
            out.update(host_to_host_leg(env, run))
        except Exception as e:   # the leg is a report, not the benchmark: never lose the bench line over it
            log("e2e leg failed: %r" % (e,))
            out["e2e_error"] = repr(e)

    # ---- the drop-in entry points on this round's code, extra keys: FASTQ text -> Debug tuples through
    # pa_process_reads (a path) and pa_record_stream_* (records pushed by the caller), with the host stages' seconds ----
    if n_gpus == 1 and rank == 0 and not args.no_ingest and args.workload == "config3":
        try:
            out.update(ingest_leg(env, run, args.ingest_reads))
        except Exception as e:   # an extra report: never lose the bench line over it
            log("ingest leg failed: %r" % (e,))
            out["ingest_error"] = repr(e)

    if n_gpus == 1 and rank == 0 and not args.no_ingest and args.workload == "config3":
        try:
            out.update(per_call_leg(env, run))
        except Exception as e:   # an extra report: never lose the bench line over it
            log("per-call leg failed: %r" % (e,))
            out["per_call_error"] = repr(e)

    if rank == 0:
        # ---- checker + CPU baseline (oracle = C port of the reference path), outside the timed region ----
        ncpu = usable_cpus()
        sample_n = 200_000
        oracle, ctr, first = run.check_sample(sample_n, ncpu)
        out["parity_sample"] = {"reads": sample_n, "bit_exact_vs_oracle": True}
        if "_e2e_sample" in out:   # the host-to-host leg's own outputs (records and ids as they arrived on the host) against the oracle
            part, coff, cids, nn, bb = out.pop("_e2e_sample")
            e_first = (rank * (K + W) + bb) * B
            e_tiles, e_lens = run.txome.simulate_host(read_len, wl["read_seed"], nn, wl["ppm"], e_first, run.wpr)
            o_res, o_coff, o_ids, _ = oracle.map_tiles(e_tiles, e_lens, run.wpr, 2, ncpu)
            helpers.assert_same_as_oracle(part, coff, cids, o_res, o_coff, o_ids, "host-to-host leg, first chunk")
            out["e2e"]["parity_sample"] = {"reads": nn, "bit_exact_vs_oracle": True, "taken_from": "the host-side COMPACT records and packed classes of this leg, unpacked"}
        out["roofline"] = roofline_of(run, ctr, map_ms, step_ms)
        rf = out["roofline"]
        kms = [r[0] + r[1] for r in per_rank]
        out["per_rank"] = {"ranks": len(per_rank), "kernel_ms": [round(x, 3) for x in kms], "kernel_ms_min": min(kms), "kernel_ms_max": max(kms), "kernel_ms_mean": sum(kms) / len(kms),
                           "step_device_ms": [round(r[3], 3) for r in per_rank],
                           # achieved algorithmic GB/s and counter-derived HBM GB/s of every rank (each maps its own B reads per step)
                           "achieved_gbps": [rf["algorithmic_bytes_per_read"] * B / (k * 1e-3) / 1e9 for k in kms],
                           "traffic_gbps": [(rf["traffic"] / (r[3] * 1e-3) / 1e9) if rf.get("traffic") else None for r in per_rank],
                           "what": "mapping stage (pa_map_pool_kernel + pa_resolve_kernel) and whole-step device time of every rank, means over the timed steps; "
                                   "roofline{} above is rank 0's"}
        if n_gpus == 1 and not args.no_cpu_baseline:
            s_tiles, s_lens = run.txome.simulate_host(read_len, wl["read_seed"], sample_n, wl["ppm"], first, run.wpr)
            rate = sample_n / max(1e-9, _time_oracle(oracle, s_tiles, s_lens, run.wpr, ncpu))
            big_n = int(min(max(rate * args.cpu_seconds, sample_n), 40_000_000))
            b_tiles, b_lens = run.txome.simulate_host(read_len, wl["read_seed"], big_n, wl["ppm"], first, run.wpr)
            secs = _time_oracle(oracle, b_tiles, b_lens, run.wpr, ncpu)
            out["cpu_baseline"] = {"value": big_n / secs, "unit": "reads/s", "cores": ncpu, "cpus_visible": os.cpu_count() or 1, "kind": "port",
                                   "sample": "first %d reads of the first timed batch, oracle/pa_oracle.c on %d pthreads (%d CPUs visible, quota/affinity %d), %.1f s"
                                             % (big_n, ncpu, os.cpu_count() or 1, ncpu, secs)}
            del b_tiles, b_lens
        del oracle

    # ---- the other workloads on the same box, driver-visible extra keys (never `value`): BASELINE.json's error-read configuration (config 5:
    # K = 31, 1 % substitutions), the headline's robustness to real graph structure (config3r: repeat families, low-complexity tracts) and the
    # the other k of the reference's CLI (config3k64: two-word k-mers) and the one index built from real gene structure (config 2: gencode_small)
    # at the headline's 100 M reads per launch
    if n_gpus == 1 and rank == 0 and args.workload == "config3" and not args.no_config5 and not args.batch:
        buffers = run.buffers
        run.aligner = None
        run.host = None
        for name, batch in (("config5", 0), ("config3r", 0), ("config3k64", 0), ("config2", 100_000_000)):
            try:
                torch.cuda.empty_cache()
                r5 = Run(env, name, batch)
                K5, W5 = min(K, 10), min(W, 2)
                r5.make_batches(K5, W5, buffers if r5.wpr == run.wpr else None)
                e5, step5, map5 = r5.timed(K5, W5)
                assert int(r5.counts.sum().item()) == K5 * r5.B
                _, ctr5, _ = r5.check_sample(100_000, usable_cpus())
                rf = roofline_of(r5, ctr5, map5, step5)
                out[name] = {"workload": "%s: %s" % (name, WORKLOADS[name]["desc"]), "value": K5 * r5.B / e5, "unit": "reads/s", "steps": K5, "warmup": W5,
                             "reads_per_step": r5.B, "ms_per_step": 1000.0 * e5 / K5, "kernel_ms": rf["kernel_ms"], "step_device_ms": rf["step_device_ms"],
                             "roofline": {kk: rf[kk] for kk in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_note", "algorithmic_bytes_per_read",
                                                               "reads_per_launch", "kernel_ms_steps")},
                             "parity_sample": {"reads": 100_000, "bit_exact_vs_oracle": True},
                             "index": {"kmers": int(r5.st.num_kmers), "nodes": int(r5.st.num_nodes), "classes": int(r5.st.num_classes), "max_class_len": int(r5.st.max_class_len),
                                       "bytes": int(r5.st.bytes_total)},
                             "oracle_counters_per_read": {kk: round(ctr5[kk] / max(ctr5["reads"], 1), 3) for kk in ("probes", "node_visits", "bases_compared", "class_sizes", "result_sizes")},
                             "index_build_s": round(r5.t_build, 3), "index_upload_s": round(r5.t_create, 3)}
                if name == "config3r":
                    out[name]["frac_of_config3"] = rf["frac"] / out["roofline"]["frac"] if out.get("roofline", {}).get("frac") else None
                    out[name]["class_structure"] = class_structure(np, r5.host)
                r5.aligner = None
                del r5
            except Exception as e:   # an extra report: never lose the bench line over it
                log("%s leg failed: %r" % (name, e))
                out[name + "_error"] = repr(e)

    out.pop("_e2e_sample", None)
    if rank == 0:
        print(json.dumps(out), flush=True)
    barrier()
    if dist is not None:
        dist.destroy_process_group()


def class_structure(np, host):
    """how far the index's classes are from the window mode's home ground: classes that do not fit two windows of 32 consecutive transcript
    ids (the reads that meet one go through the list tiers of map_pool.hip), and the share of the graph's k-mers that carry such a class"""
    a = host.arrays()
    off = a["ec_offset"].astype(np.int64)
    ids = a["ec_ids"].astype(np.int64)
    nc = int(a["num_classes"])
    clen = off[1:] - off[:-1]
    first = ids[off[:-1]]
    owner = np.repeat(np.arange(nc), clen)
    beyond1 = ids > first[owner] + 31
    big = np.int64(1) << 40
    second = np.minimum.reduceat(np.where(beyond1, ids, big), off[:-1])
    beyond2 = beyond1 & (ids > second[owner] + 31)
    is_list = np.add.reduceat(beyond2.astype(np.int64), off[:-1]) > 0
    kmers = (a["node_len"].astype(np.int64) - int(a["k"]) + 1)
    list_kmers = int(kmers[is_list[a["node_colour"]]].sum())
    return {"classes": nc, "classes_beyond_two_windows": int(is_list.sum()), "kmer_share_of_those_classes": list_kmers / max(int(kmers.sum()), 1),
            "classes_over_32_ids": int((clen > 32).sum()), "classes_over_100_ids": int((clen > 100).sum()), "max_class_len": int(clen.max()),
            "kmer_weighted_mean_class_len": float((kmers * clen[a["node_colour"]]).sum() / max(int(kmers.sum()), 1))}


def write_fastq(txome, path, n, read_len, read_seed, wpr, np):
    """n simulated reads as four-line FASTQ records "@r%09d" / bases / "+" / qualities; returns the file size"""
    lut = np.frombuffer(b"ACGT", np.uint8)
    with open(path, "wb") as f:
        for first in range(0, n, 1 << 20):
            m = min(1 << 20, n - first)
            tiles, _ = txome.simulate_host(read_len, read_seed, m, 0, first, wpr)
            words = tiles.reshape(-1, wpr, 64).transpose(0, 2, 1).reshape(-1, wpr)[:m]    # [read][word]
            shifts = (2 * np.arange(32, dtype=np.uint64))[None, None, :]
            bases = ((words[:, :, None] >> shifts) & np.uint64(3)).astype(np.uint8).reshape(m, wpr * 32)[:, :read_len]
            rec = np.empty((m, 16 + 2 * read_len), np.uint8)
            ids = np.char.zfill(np.arange(first, first + m).astype("U9"), 9)
            rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
            rec[:, 2:11] = np.frombuffer("".join(ids).encode(), np.uint8).reshape(m, 9)
            rec[:, 11] = 10
            rec[:, 12:12 + read_len] = lut[bases]
            rec[:, 12 + read_len:15 + read_len] = np.frombuffer(b"\n+\n", np.uint8)
            rec[:, 15 + read_len:15 + 2 * read_len] = ord("I")
            rec[:, 15 + 2 * read_len] = 10
            f.write(rec.tobytes())
    return os.path.getsize(path)


def ingest_leg(env, run, n):
    """FASTQ text (page cache) -> the reference's Debug tuples -> /dev/null, through both drop-in forms. pa_process_reads (round 6): the
    host only reads windows of the file into pinned memory (stage "pack"), the windows go to HBM as they are and the GPU finds the records;
    "gpu_wait" is the time the caller waits for a window's copy + scan — the link — and "scan" is what the host's own scan still does (the
    last MiB of the text). The rate is bounded by max(read, link): FASTQ of 150-base reads is 316 bytes per read over PCIe."""
    pa, np = env["pa"], env["np"]
    wl = run.wl
    ncpu = usable_cpus()
    d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    fq = os.path.join(d, "pa_bench_ingest_%d.fq" % os.getpid())
    try:
        size = write_fastq(run.txome, fq, n, wl["read_len"], wl["read_seed"], run.wpr, np)
        pa.process_reads(fq, run.aligner, "/dev/null", ncpu)           # warm-up: page cache, pinned buffers
        best = None
        runs = []
        for _ in range(6):
            t0 = time.perf_counter()
            got, _ = pa.process_reads(fq, run.aligner, "/dev/null", ncpu)
            dt = time.perf_counter() - t0
            assert got == n
            runs.append(round(n / dt / 1e6, 1))
            st = pa.process_reads_stage_seconds()
            if best is None or dt < best[0]:
                best = (dt, st)
        dt, st = best
        stages = {k: round(st[k], 4) for k in ("scan_s", "pack_s", "gpu_wait_s", "launch_s", "render_s", "writer_wait_s", "total_s")}
        bound = max(("scan_s", "pack_s", "gpu_wait_s", "render_s", "writer_wait_s"), key=lambda k: st[k])
        out = {"ingest_reads_per_s": n / dt, "ingest_bound_stage": bound.replace("_s", ""),
               "ingest": {"what": "pa_process_reads: %d reads of %d bp, %.2f GB FASTQ in the page cache -> tuples to /dev/null, %d worker threads (the box's CPU quota); "
                                  "best of six calls (the host's read of the file varies from call to call: runs_Mreads_per_s); stages: scan = the host's own record scan (the text's last MiB), pack = reading the windows into pinned memory, "
                                  "gpu_wait = waiting for a window's copy and scan" % (n, wl["read_len"], size / 1e9, ncpu),
                          "seconds": dt, "fastq_GBps": size / dt / 1e9, "stages": stages, "runs_Mreads_per_s": runs,
                          "text_bytes_per_read": size / n,
                          "link_bound_reads_per_s_at_57GBps": 57e9 / (size / n),
                          "host_scan_plus_gather_ms_per_8M_reads": round(st["scan_s"] * 8e6 / n * 1e3, 2),
                          "reference_counterparts": "one reader behind a mutex (utils.rs:152-157), one println! per read on the consumer thread (pseudoaligner.rs:490)"}}
        # pa_process_reads_multi with the handle listed twice: two lanes on this GPU (on a node, one lane per GPU: each has a link of its own)
        t0 = time.perf_counter()
        got, _ = pa.process_reads_multi(fq, [run.aligner, run.aligner], "/dev/null", ncpu)
        dt2 = time.perf_counter() - t0
        t0 = time.perf_counter()
        got, _ = pa.process_reads_multi(fq, [run.aligner, run.aligner], "/dev/null", ncpu)
        dt2 = min(dt2, time.perf_counter() - t0)
        assert got == n
        out["ingest"]["two_lanes_one_gpu_reads_per_s"] = n / dt2
        # the record-stream form: the caller reads the file (here: numpy slices of the same text) and pushes records
        text = np.fromfile(fq, np.uint8).reshape(n, 16 + 2 * wl["read_len"])
        ids = np.ascontiguousarray(text[:, 1:11])
        seqs = np.ascontiguousarray(text[:, 12:12 + wl["read_len"]])
        rs = pa.RecordStream(run.aligner, ncpu)
        import ctypes as C
        chunk = 1 << 20
        id_off = (np.arange(chunk + 1, dtype=np.uint64) * 10)
        seq_off = (np.arange(chunk + 1, dtype=np.uint64) * wl["read_len"])
        buf = C.create_string_buffer(1 << 28)   # (one pull per rendered batch: the library copies big pieces on its worker pool)
        nb = C.c_size_t()
        def stream_all(rs):
            pulled = 0
            for lo in range(0, n, chunk):
                m = min(chunk, n - lo)
                pa.check(pa.lib().pa_records_push(rs._h, ids[lo:lo + m].ctypes.data, id_off.ctypes.data, seqs[lo:lo + m].ctypes.data, seq_off.ctypes.data, m))
                while True:
                    pa.check(pa.lib().pa_records_pull(rs._h, buf, len(buf), C.byref(nb)))
                    if nb.value == 0:
                        break
                    pulled += nb.value
            rs.flush()
            while True:
                pa.check(pa.lib().pa_records_pull(rs._h, buf, len(buf), C.byref(nb)))
                if nb.value == 0:
                    break
                pulled += nb.value
            return pulled
        # warm-up, as for pa_process_reads above: the stream shares its pinned batch buffers with pa_process_reads through the index, and
        # its batches (2 Mi reads) are larger than the windows that call scans (buffers sized by those are regrown on first use)
        stream_all(rs)
        rs.close()
        rs = pa.RecordStream(run.aligner, ncpu)
        t0 = time.perf_counter()
        pulled = stream_all(rs)
        dt_rs = time.perf_counter() - t0
        st_rs = rs.stage_seconds()
        assert rs.stats()[0] == n
        rs.close()
        out["ingest"]["record_stream"] = {"reads_per_s": n / dt_rs, "seconds": dt_rs, "tuple_bytes": pulled,
                                          "stages": {k: round(st_rs[k], 4) for k in ("pack_s", "gpu_wait_s", "launch_s", "render_s", "total_s")},
                                          "what": "pa_records_push / pull of the same reads in chunks of 1 Mi records (the copy into the batch and the pull are the caller's time)"}
        return out
    finally:
        if os.path.exists(fq):
            os.unlink(fq)


def host_to_host_leg(env, run):
    """SURVEY §8d's literal metric: the batch from PINNED HOST tiles to COMPLETE per-read outputs on the host — through the product's own
    host-to-host entry (pa_map_tiles_host, csrc/host_batch.cpp): chunks of 1 M reads rotate over four streams of the one index handle (H2D
    of chunk i + 1, the kernels of chunk i and D2H of chunk i - 1 overlap); back come the COMPACT 8-byte records, the packed classes that
    are no index classes ({length, ids...} in read order) and the count table. The batch is uniform (every read has read_len bases): no
    length array crosses the link."""
    pa, torch, dev, np = env["pa"], env["torch"], env["dev"], env["np"]
    aligner, B, wpr = run.aligner, run.B, run.wpr
    read_len = WORKLOADS[run.name]["read_len"]
    b = run.W % run.n_batches
    h_tiles = torch.empty(run.tile_words, dtype=torch.int64, pin_memory=True)
    h_compact = torch.empty(B, dtype=torch.int64, pin_memory=True)
    h_counts = torch.empty(aligner.counts_len(), dtype=torch.int64, pin_memory=True)
    h_tiles.copy_(run.tiles[b])
    assert bool((run.lens[b] == read_len).all())
    NS = int(os.environ.get("PA_E2E_STREAMS", "4"))
    chunk = min(B, int(os.environ.get("PA_E2E_CHUNK", str(min(B, 1_000_000))))) // 64 * 64 or B
    h_packed = torch.empty(max(aligner.arena_hint(chunk), B // 2), dtype=torch.int32, pin_memory=True)
    torch.cuda.synchronize()

    def host_to_host():
        t0 = time.perf_counter()
        words = aligner.map_tiles_host(h_tiles.data_ptr(), B, wpr, h_compact.data_ptr(), h_packed.data_ptr(), h_packed.numel(), uniform_len=read_len,
                                       h_counts=h_counts.data_ptr(), chunk_reads=chunk, n_streams=NS)
        return time.perf_counter() - t0, words
    host_to_host()                       # warm-up: the streams' launch contexts and the staging buffers are created on first use
    runs = [host_to_host() for _ in range(6)]
    e2e_s, packed_words = min(runs, key=lambda r: r[0])
    assert os.environ.get("PA_MAP_ABLATE") or int(h_counts.sum()) == B
    # parity of THIS leg's outputs: the first reads' compact records + packed classes as they arrived on the host, unpacked, against the oracle
    nn = min(chunk, 100_000)
    lo32 = (h_compact[:nn].numpy().view(np.uint64) & np.uint64(0xFFFFFFFF))
    n_pk = int(((lo32 & np.uint64(pa.PA_COMPACT_PACKED)) != 0).sum())
    first_packed = h_packed[: int(packed_words)].numpy().view(np.uint32)
    pos = 0
    for _ in range(n_pk):                # (the sample's packed entries are the first n_pk of the stream: walk that many)
        pos += 1 + int(first_packed[pos])
    part, coff, cids = pa.unpack_compact(h_compact[:nn].numpy().view(np.uint64).copy(), first_packed[:pos].copy(), run.host)
    sample = (part, coff, cids, nn, b)       # compared with the oracle once it is built (main, checker section)
    h2d_bytes = B * wpr * 8
    n_chunks = (B + chunk - 1) // chunk
    return {"e2e_reads_per_s": B / e2e_s, "e2e_pcie_frac": h2d_bytes / e2e_s / 63e9, "e2e_link_frac_of_measured_57": h2d_bytes / e2e_s / 57e9,
            "e2e": {"ms": 1000.0 * e2e_s, "runs_ms": [round(1000.0 * r[0], 2) for r in runs], "chunks": n_chunks, "reads_per_chunk": chunk, "streams": NS, "h2d_bytes": h2d_bytes,
                    "d2h_bytes": B * 8 + 4 * int(packed_words) + 8 * aligner.counts_len(), "record_bytes": 8, "packed_class_words_on_host": int(packed_words),
                    "novel_class_ids_left_on_device": 0, "parity_sample": None,
                    "what": "pa_map_tiles_host: pinned host 2-bit tiles (uniform batch: no length array) -> H2D || kernels || D2H of the COMPACT 8-byte records "
                            "(pa_results_compact_device) and of each chunk's packed classes ({length, ids...} of the classes that are no index classes, no padding) on several "
                            "streams of one index handle -> records + ids + count table on the host; best of six calls (runs_ms lists all); link = PCIe Gen5 x16, 63 GB/s per direction spec, "
                            "57 GB/s measured (profiles/r02_pcie_bw.json)"},
            "_e2e_sample": sample}


def per_call_leg(env, run, n_single=2000, n_batch=1_000_000):
    """What a caller that loops over `map_read` pays (the reference's own test does: src/build_index.rs:309) against the batch form
    of the same boundary: pa_map_read_packed, one call per read (H2D + launch + D2H each), and pa_map_batch_packed on n_batch reads
    held 2-bit packed (what amd::map_reads binds). Extra keys: `map_read_us` (mean wall time of one call) and `map_reads_batch_reads_per_s`."""
    pa, np = env["pa"], env["np"]
    wl, wpr = WORKLOADS[run.name], run.wpr
    tiles, lens = run.txome.simulate_host(wl["read_len"], wl["read_seed"], n_batch, wl["ppm"], 0, wpr)
    words = np.ascontiguousarray(tiles.reshape(-1, wpr, 64).transpose(0, 2, 1).reshape(-1, wpr)[:n_batch])       # [read][word], LSB-first
    a = run.aligner
    for i in range(20):                                                                     # warm-up (the per-thread buffers of the single-read path)
        a.map_read_packed(words[i], int(lens[i]))
    t0 = time.perf_counter()
    for i in range(n_single):
        a.map_read_packed(words[i], int(lens[i]))
    single_s = (time.perf_counter() - t0) / n_single
    off = (np.arange(n_batch + 1, dtype=np.uint64) * np.uint64(wpr))
    a.map_batch_packed(words.reshape(-1)[: 4096 * wpr], off[:4097], lens[:4096])
    t0 = time.perf_counter()
    res, coff, ids = a.map_batch_packed(words.reshape(-1), off, lens)
    batch_s = time.perf_counter() - t0
    one = a.map_read_packed(words[7], int(lens[7]))
    assert one is not None and one[0] == ids[int(coff[7]):int(coff[8])].tolist() and one[1] == int(res["coverage"][7])
    return {"map_read_us": 1e6 * single_s, "map_reads_batch_reads_per_s": n_batch / batch_s,
            "per_call": {"calls": n_single, "batch_reads": n_batch, "what": "pa_map_read_packed once per read vs pa_map_batch_packed on host-resident packed reads "
                                                                           "(pack to tiles, H2D, map, D2H, class ids resolved on the host), same index"}}


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
