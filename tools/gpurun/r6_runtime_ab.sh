#!/bin/bash
# the same legs on the system's HIP runtime (ROCm 7.2: what the product links) and on the one torch bundles (7.0: what a process gets that imported torch first)
cd $GRAFT_REPO_ROOT
summ() { grep -v "Done Mapping" | grep "windows to\|value\|ms" | cut -c1-200 | sed 's/"metric": "reads.sec FASTQ text -> Debug tuples (pa_process_reads, .dev.null)", //g; s/"unit": "reads.s", "threads": 16, "reads": 8000000, //g; s/\[pa ingest\] windows to the GPU: 2013.3 MB in//' | awk '{printf "%s | ", $0} END {print ""}'; }
for rep in 1 2; do
  echo "== ingest, system runtime"
  PA_VERBOSE=1 python tools/bench_ingest.py --reads 8000000 --threads 16,16,16,16,16,16 2>&1 | summ
  echo "== ingest, torch imported first"
  PA_VERBOSE=1 python -c "import torch, runpy, sys; sys.argv=['tools/bench_ingest.py','--reads','8000000','--threads','16,16,16,16,16,16']; runpy.run_path('tools/bench_ingest.py', run_name='__main__')" 2>&1 | summ
done
