#!/usr/bin/env python3
"""Per-launch averages of rocprofv3 --pmc csv output, per kernel (largest launches of each kernel only).
usage: pmc_summary.py <dir with pmc_* subdirectories> [kernel substring ...]   (default: pa_map_pool pa_keys_)"""
import collections
import csv
import glob
import sys

root = sys.argv[1]
kernels = sys.argv[2:] or ["pa_map_pool", "pa_resolve", "pa_keys_hist", "pa_keys_scan", "pa_keys_scatter", "pa_keys_count"]
print("# rocprofv3 --pmc (one pass per directory) averages per launch, launches with the full batch only")
for d in sorted(glob.glob(root + "/pmc_*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        allrows = list(csv.DictReader(open(f)))
        for kernel in kernels:
            rows = [r for r in allrows if kernel in r["Kernel_Name"]]
            if not rows:
                continue
            gmax = max(int(r["Grid_Size"]) for r in rows)
            acc = collections.defaultdict(list)
            for r in rows:
                if int(r["Grid_Size"]) == gmax:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, v in sorted(acc.items()):
                if not kernel.startswith("pa_map_pool"):   # every launch of a batch has the same grid: keep the launches of the full batch (largest values)
                    top = max(v)
                    v = [x for x in v if x > 0.5 * top] if top > 0 else v
                print("%-12s %-18s %-32s launches=%d avg_per_launch=%.6g" % (d.split("/")[-1], kernel, k, len(v), sum(v) / len(v)))
