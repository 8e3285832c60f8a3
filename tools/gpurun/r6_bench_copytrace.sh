#!/bin/bash
# inside bench.py's process: are the windows' copies DMA copies or blit kernels? kernel + memory-copy trace of a short bench run with the FASTQ leg
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/bench_copytrace
rm -rf $O
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-config5 --no-e2e > $O.json 2> $O.err
python - <<PY
import csv, glob, collections
ks = glob.glob("$O/**/*kernel_trace.csv", recursive=True); cs = glob.glob("$O/**/*memory_copy_trace.csv", recursive=True)
print(ks, cs)
kc = collections.Counter(); kt = collections.Counter()
for f in ks:
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"][:60]; kc[n] += 1; kt[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for n, t in kt.most_common(25): print("%-62s %7d calls %10.2f ms" % (n, kc[n], t))
cc = collections.Counter(); ct = collections.Counter(); cb = collections.Counter()
for f in cs:
    rd = csv.DictReader(open(f))
    for r in rd:
        d = r.get("Direction", "?"); b = int(r.get("Size", r.get("Bytes", 0)) or 0)
        key = (d, "64MiB" if b == 67108864 else ">1MB" if b > (1 << 20) else "small")
        cc[key] += 1; ct[key] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; cb[key] += b
for k in cc: print(k, cc[k], "copies", "%.2f ms" % ct[k], "%.1f GB/s" % (cb[k] / max(ct[k], 1e-9) / 1e6))
PY
find $O -name "*.csv" -size +20M -delete
