#!/usr/bin/env python3
"""median / max duration per kernel of a rocprofv3 --kernel-trace csv directory, full names. usage: kernel_times.py <dir> [substr ...]"""
import collections
import csv
import glob
import sys

acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if len(sys.argv) > 2 and not any(s in k for s in sys.argv[2:]):
        continue
    print("%-70s n=%-4d med=%9.1f us max=%9.1f us" % (k.replace("pa::(anonymous namespace)::", "").split("(")[0][:70], len(v), sorted(v)[len(v) // 2], max(v)))
