#!/usr/bin/env python3
"""profiles/latest_pmc.json from the PMC summaries of tools/gpurun/profile_r03.sh: FETCH_SIZE + WRITE_SIZE per STEP (every kernel of
a pa_map_count_batch_device call) and the sha256 of the kernel sources they were measured on (bench.py compares it with the built
sources and reports no traffic figure on a mismatch). usage: make_latest_pmc.py config3=<pmc.txt>[:<bench.json of the same box>] config5=<pmc.txt>[:...] [reads_per_launch]
(the bench line of the profiled box gives its own step time: bytes measured there are divided by the time measured THERE)"""
import importlib.util
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
spec = importlib.util.spec_from_file_location("bench", ROOT / "bench.py")
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)

reads = 100_000_000
out = {"note": "FETCH_SIZE / WRITE_SIZE (KB) per step = sum over the kernels of one pa_map_count_batch_device call (pa_map_pool_kernel, pa_resolve_kernel, "
               "pa_keys_hist / scan / scatter / count_kernel), rocprofv3 --pmc, separate passes, per-launch averages over the full-batch launches. "
               "Calibration of the counters on this chip: profiles/r05_pmc_calibration.txt — EVERY memory-side read request of gfx950 is 128 bytes "
               "(TCC_EA0_RDREQ_128B == TCC_EA0_RDREQ for streams, random 64-byte lines, 16-byte slots and chain blocks alike) and FETCH_SIZE tallies each at 64 bytes; "
               "WRITE_SIZE is exact at 32-byte sectors. READ_REQUEST_BYTES = sum over request sizes of count x size.",
       "kernel_source_sha256": bench.kernel_source_sha256(), "workloads": {}}
for arg in sys.argv[1:]:
    if "=" not in arg:
        reads = int(arg)
        continue
    name, path = arg.split("=", 1)
    box_step_ms = None
    if ":" in path:
        path, bench_json = path.split(":", 1)
        try:
            box_step_ms = json.load(open(bench_json))["roofline"]["step_device_ms"]
        except (OSError, ValueError, KeyError):
            box_step_ms = None
    per_kernel = {}
    for line in open(path):
        f = line.split()
        if len(f) >= 5 and f[2] in ("FETCH_SIZE", "WRITE_SIZE", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum",
                                     "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"):
            per_kernel.setdefault(f[1], {})[f[2]] = float(f[4].split("=")[1])
    out["workloads"][name] = {"reads_per_launch": reads, "kernels": per_kernel,
                              "FETCH_SIZE_KB": sum(k.get("FETCH_SIZE", 0.0) for k in per_kernel.values()),
                              "WRITE_SIZE_KB": sum(k.get("WRITE_SIZE", 0.0) for k in per_kernel.values()),
                              # memory-side read bytes from the request counters by size (profiles/r05_pmc_calibration.txt: FETCH_SIZE's formula
                              # tallies gfx950's 128-byte requests at 64 bytes)
                              "READ_REQUEST_BYTES": sum(128.0 * k.get("TCC_EA0_RDREQ_128B_sum", 0.0) + 64.0 * k.get("TCC_EA0_RDREQ_64B_sum", 0.0) +
                                                        32.0 * k.get("TCC_EA0_RDREQ_32B_sum", 0.0) for k in per_kernel.values()),
                              "read_requests": sum(k.get("TCC_EA0_RDREQ_sum", 0.0) for k in per_kernel.values()),
                              "write_requests": sum(k.get("TCC_EA0_WRREQ_sum", 0.0) for k in per_kernel.values()),
                              "map_kernel_l2_misses": per_kernel.get("pa_map_pool", {}).get("TCC_MISS_sum"), "profiled_box_step_ms": box_step_ms,
                              "source": "profiles/r06_%s_pmc.txt" % name}
json.dump(out, open(ROOT / "profiles" / "latest_pmc.json", "w"), indent=1)
print(json.dumps({k: (v["FETCH_SIZE_KB"], v["WRITE_SIZE_KB"]) for k, v in out["workloads"].items()}))
