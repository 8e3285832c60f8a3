#!/usr/bin/env python3
"""End-to-end rate of pa_process_reads: FASTQ text in page cache -> tuples to /dev/null, reads/s.

Run on the GPU box: python tools/bench_ingest.py [--reads N] [--index-cache PATH]. Config-3 index, 150 bp reads."""
import argparse, importlib, json, os, sys, time
from pathlib import Path
import numpy as np

if any(a.startswith("--ballast") for a in sys.argv):
    import torch  # noqa: F401  (before the product library: one HIP runtime in the process)

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
pa = importlib.import_module("rust-pseudoaligner_amd")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=16_000_000)
    ap.add_argument("--index-cache", default="/tmp/g.idx")
    ap.add_argument("--dir", default="/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
    ap.add_argument("--threads", default="")
    ap.add_argument("--lanes", type=int, default=1, help="pa_process_reads_multi with the handle listed this many times (lanes of ONE GPU)")
    ap.add_argument("--ballast-device-gb", type=float, default=0, help="diagnosis: this much device memory allocated (torch) before the calls")
    ap.add_argument("--ballast-pinned-gb", type=float, default=0, help="diagnosis: this much pinned host memory allocated (torch) before the calls")
    ap.add_argument("--ballast-touch", action="store_true", help="diagnosis: the device ballast is written once")
    args = ap.parse_args()
    t0 = time.time()
    tx = pa.Txome.synthesize(58000, 203000, 7)
    if os.path.exists(args.index_cache):
        hi = pa.HostIndex.load(args.index_cache)
    else:
        hi = pa.HostIndex.from_txome_device(tx, 24, 0)   # the GPU builder (0.2 s); nothing worth caching
    al = pa.Pseudoaligner(hi)
    print("[ingest] index ready %.1f s" % (time.time() - t0), file=sys.stderr)
    n, L, wpr = args.reads, 150, 5
    fq = Path(args.dir) / "pa_ingest_bench.fq"
    t0 = time.time()
    with open(fq, "wb") as f:   # "@r%09d\n" + seq + "\n+\n" + qual + "\n", written in chunks of 1 M reads
        lut = np.frombuffer(b"ACGT", np.uint8)
        for first in range(0, n, 1 << 20):
            m = min(1 << 20, n - first)
            tiles, lens = tx.simulate_host(L, 2, m, 0, first, wpr)
            t = tiles.reshape(-1, wpr, 64)                       # [tile][word][read]
            words = t.transpose(0, 2, 1).reshape(-1, wpr)[:m]    # [read][word]
            shifts = (2 * np.arange(32, dtype=np.uint64))[None, None, :]
            bases = ((words[:, :, None] >> shifts) & np.uint64(3)).astype(np.uint8).reshape(m, wpr * 32)[:, :L]
            rec = np.empty((m, 16 + 2 * L), np.uint8)
            ids = np.char.zfill(np.arange(first, first + m).astype("U9"), 9)
            rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
            rec[:, 2:11] = np.frombuffer("".join(ids).encode(), np.uint8).reshape(m, 9)
            rec[:, 11] = 10
            rec[:, 12:12 + L] = lut[bases]
            rec[:, 12 + L:12 + L + 3] = np.frombuffer(b"\n+\n", np.uint8)
            rec[:, 15 + L:15 + 2 * L] = ord("I")
            rec[:, 15 + 2 * L] = 10
            f.write(rec.tobytes())
    size = fq.stat().st_size
    print("[ingest] %d reads, %.2f GB FASTQ written in %.1f s" % (n, size / 1e9, time.time() - t0), file=sys.stderr)
    try:   # where the file's page cache lives (NUMA nodes of a sample of its pages) and where the GPU hangs
        import mmap
        with open(fq, "rb") as f:
            mm = mmap.mmap(f.fileno(), 0, prot=mmap.PROT_READ)
            for off in range(0, size, 1 << 22):
                mm[off]
            base = None
            for line in open("/proc/self/numa_maps"):
                if "pa_ingest_bench.fq" in line:
                    base = line.strip()
            mm.close()
        gpu_nodes = [open(p).read().strip() for p in sorted(__import__("glob").glob("/sys/class/drm/card*/device/numa_node"))]
        print("[ingest] file pages: %s ; GPU numa nodes by card: %s ; writer thread ran on cpu %d" % (base, ",".join(gpu_nodes), os.sched_getcpu() if hasattr(os, "sched_getcpu") else -1), file=sys.stderr)
    except Exception as e:  # noqa: BLE001
        print("[ingest] numa probe failed: %r" % (e,), file=sys.stderr)
    ncpu = os.cpu_count() or 1
    try:
        print("[ingest] cpu_count %d, affinity %d, cgroup cpu.max %s" % (ncpu, len(os.sched_getaffinity(0)),
              open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "?"), file=sys.stderr)
    except OSError:
        pass
    threads = [int(x) for x in args.threads.split(",") if x] or sorted({ncpu, max(1, ncpu // 2), max(1, ncpu // 4), min(16, ncpu)}, reverse=True)
    ballast = []
    if args.ballast_device_gb or args.ballast_pinned_gb:
        import torch
        for _ in range(int(args.ballast_device_gb)):
            ballast.append(torch.empty(1 << 30, dtype=torch.uint8, device="cuda"))
            if args.ballast_touch:
                ballast[-1].zero_()
        for _ in range(int(args.ballast_pinned_gb)):
            ballast.append(torch.empty(1 << 30, dtype=torch.uint8, pin_memory=True))
        torch.cuda.synchronize()
    pa.process_reads(str(fq), al, "/dev/null", ncpu)   # warm-up (page cache, pinned buffers, kernels)
    os.environ["PA_VERBOSE"] = "1"
    def throttled():   # the cgroup's CPU-quota throttling so far (cgroup v2): periods throttled, microseconds
        try:
            st = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat").read().splitlines())
            return int(st.get("nr_throttled", 0)), int(st.get("throttled_usec", 0))
        except OSError:
            return 0, 0
    for t in threads:
        th0 = throttled()
        t0 = time.time()
        got, flagged = pa.process_reads(str(fq), al, "/dev/null", t) if args.lanes == 1 else pa.process_reads_multi(str(fq), [al] * args.lanes, "/dev/null", t)
        dt = time.time() - t0
        th1 = throttled()
        assert got == n
        print("[ingest] threads %d: %.1f M reads/s, cgroup throttled %d periods / %.1f ms during the call" % (t, n / dt / 1e6, th1[0] - th0[0], (th1[1] - th0[1]) / 1e3), file=sys.stderr)
        print(json.dumps({"metric": "reads/sec FASTQ text -> Debug tuples (pa_process_reads, /dev/null)", "value": n / dt, "unit": "reads/s",
                          "threads": t, "reads": n, "seconds": dt, "fastq_GBps": size / dt / 1e9, "flagged": flagged}))
    fq.unlink()


if __name__ == "__main__":
    main()
