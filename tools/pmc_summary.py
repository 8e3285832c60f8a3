#!/usr/bin/env python3
"""Per-launch averages of rocprofv3 --pmc csv output for one kernel (largest launches only)."""
import collections
import csv
import glob
import sys

root, kernel = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "pa_map_"
print("# rocprofv3 --pmc (one pass per directory) averages per launch of %s, launches with the full batch only" % kernel)
for d in sorted(glob.glob(root + "/pmc_*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if kernel in r["Kernel_Name"]]
        if not rows:
            continue
        gmax = max(int(r["Grid_Size"]) for r in rows)
        acc = collections.defaultdict(list)
        for r in rows:
            if int(r["Grid_Size"]) == gmax:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print("%-12s %-32s launches=%d avg_per_launch=%.6g" % (d.split("/")[-1], k, len(v), sum(v) / len(v)))
