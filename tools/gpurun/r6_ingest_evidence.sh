#!/bin/bash
# round 6 evidence of the window pipeline (pa_process_reads): rates by window size, the kernel / copy trace of one run
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_ingest
mkdir -p $O
bash $R/tools/gpurun/ingest_probe.sh > $O/windows.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats -d $O/prof -o ingest -- python $R/tools/bench_ingest.py --reads 8000000 --threads 16,16 > $O/prof.log 2>&1
cd $R
python tools/rocprof_summary.py $O/prof/ingest_results.db "python tools/bench_ingest.py --reads 8000000 --threads 16,16  (index build + one warm-up + two timed pa_process_reads calls)" > $O/kernel_trace_summary.txt
python - <<'PY' > $O/timeline.txt
import sqlite3
db=sqlite3.connect("gpurun_out/r06_ingest/prof/ingest_results.db")
cur=db.cursor()
ev=[]
for r in cur.execute("select name,start,end,size from memory_copies").fetchall():
    if r[3] > 1000000: ev.append((r[1],r[2],"COPY %s %d bytes"%(r[0].replace("MEMORY_COPY_",""),r[3])))
for r in cur.execute("select name,start,end from kernels").fetchall():
    n=r[0].replace("pa::(anonymous namespace)::","").split("(")[0][:44]
    if r[2]-r[1] > 15000: ev.append((r[1],r[2],"KERNEL "+n))
ev.sort()
big=[e for e in ev if e[2].startswith("COPY HOST_TO_DEVICE") and int(e[2].split()[2])>60000000]
t0=big[-20][0]
print("# GPU timeline of pa_process_reads (last call of the profiled process), ms from the copy of one 64 MiB window: start, duration, what (kernels > 15 us, copies > 1 MB)")
for e in ev:
    if t0 <= e[0] < t0+6e6: print("%8.3f %7.3f  %s"%((e[0]-t0)/1e6,(e[1]-e[0])/1e6,e[2]))
PY
rm -rf $O/prof
tail -4 $O/windows.txt | cut -c1-200; head -12 $O/kernel_trace_summary.txt | cut -c1-150; head -30 $O/timeline.txt
