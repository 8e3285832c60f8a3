#!/usr/bin/env python3
"""bench.py on a SCALED config-3 transcriptome (genes and transcripts multiplied by PA_SCALE, e.g. 0.5: chain blocks that fit the Infinity
Cache whole): how much of the kernel's time depends on the index's size? Usage: PA_SCALE=0.5 python tools/bench_scaled.py [bench.py flags]"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
f = float(os.environ.get("PA_SCALE", "0.5"))
w = bench.WORKLOADS["config3"]
w["genes"], w["transcripts"] = int(w["genes"] * f), int(w["transcripts"] * f)
w["desc"] += " (transcriptome scaled by %g)" % f
bench.main()
