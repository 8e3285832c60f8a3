#!/bin/bash
# pa_map_tiles_host (tools/bench_e2e.py, 100 M reads, four calls): which copies the runtime runs on the DMA engines, which as blit kernels, and how busy the link is
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/e2e_timeline
rm -rf $O
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -o t -- python $R/tools/bench_e2e.py > $O.log 2>&1
tail -3 $O.log | cut -c1-300
python - <<PY
import csv, glob, numpy as np
ks = glob.glob("$O/**/*kernel_trace.csv", recursive=True); cs = glob.glob("$O/**/*memory_copy_trace.csv", recursive=True)
K = [r for f in ks for r in csv.DictReader(open(f))]
C = [r for f in cs for r in csv.DictReader(open(f))]
ev = []
for r in C:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "dma " + r["Direction"].replace("MEMORY_COPY_", "")))
for r in K:
    n = r["Kernel_Name"]
    if "copyBuffer" in n: ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "blit"))
    elif "pa_map_pool" in n: ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "map"))
ev.sort()
maps = [e for e in ev if e[2] == "map"]
# the calls: 50 map kernels each (plus warm-up): take the last 200 map kernels = four calls
calls = [maps[i:i + 50] for i in range(len(maps) - 200, len(maps), 50)]
for c in calls:
    t0, t1 = c[0][0] - 2_000_000, c[-1][1] + 1_000_000
    sel = [e for e in ev if e[0] >= t0 and e[1] <= t1]
    for kind in ("dma HOST_TO_DEVICE", "dma DEVICE_TO_HOST", "blit"):
        d = np.array([(e[1] - e[0]) / 1e6 for e in sel if e[2] == kind])
        big = d[d > 0.2]
        print("%-22s %4d events, %3d over 0.2 ms: sum %.1f ms, mean %.2f ms, max %.2f ms" % (kind, len(d), len(big), big.sum(), big.mean() if len(big) else 0, big.max() if len(big) else 0))
    print("call span %.1f ms" % ((t1 - t0) / 1e6 - 3.0))
PY
find $O -name "*.csv" -size +20M -delete
