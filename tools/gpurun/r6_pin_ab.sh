#!/bin/bash
# (the knob this script sets existed only in the build of that experiment: profiles/r06_ingest_link_diagnosis.txt)
# readers pinned next to the GPU (default) against unpinned (PA_INGEST_NO_PIN=1): standalone and inside bench.py's process
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for pin in "" 1; do
  echo "== standalone, PA_INGEST_NO_PIN=$pin"
  PA_INGEST_NO_PIN=$pin PA_VERBOSE=1 python tools/bench_ingest.py --reads 8000000 --threads 16,16,16,16,16,16 2>&1 | grep -v "Done Mapping" | grep "windows to\|value" | cut -c1-200 | awk '{printf "%s | ", $0} END {print ""}' | sed 's/"metric": "reads.sec FASTQ text -> Debug tuples (pa_process_reads, .dev.null)", //g; s/"unit": "reads.s", "threads": 16, "reads": 8000000, //g'
done
done
for pin in "" 1; do
  echo "== bench.py, PA_INGEST_NO_PIN=$pin"
  PA_INGEST_NO_PIN=$pin PA_VERBOSE=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config5 2> /tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ingest']['runs_Mreads_per_s'], d['ingest']['stages'], d['ingest'].get('two_lanes_one_gpu_reads_per_s'), 'e2e', d.get('e2e',{}).get('runs_ms'))"
  grep "windows to" /tmp/err.txt | cut -c1-120
done
