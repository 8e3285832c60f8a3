#!/bin/bash
# (the knob this script sets existed only in the build of that experiment: profiles/r06_ingest_link_diagnosis.txt)
# a window's scan and the tuples' way back on streams of their own (default) against everything on the lane's one stream (PA_INGEST_ONE_STREAM=1)
cd $GRAFT_REPO_ROOT
summ() { grep -v "Done Mapping" | grep "windows to\|value\|Error\|error" | cut -c1-200 | sed 's/"metric": "reads.sec FASTQ text -> Debug tuples (pa_process_reads, .dev.null)", //g; s/"unit": "reads.s", "threads": 16, "reads": 8000000, //g; s/\[pa ingest\] windows to the GPU: 2013.3 MB in//; s/"seconds".*//' | awk '{printf "%s | ", $0} END {print ""}'; }
python tools/bench_ingest.py --reads 2000000 --threads 16 > /dev/null 2>&1   # (a fresh box's first process is slow whatever it runs)
for rep in 1 2 3; do
for one in "" 1; do
  echo "== standalone, PA_INGEST_ONE_STREAM=$one"
  PA_INGEST_ONE_STREAM=$one PA_VERBOSE=1 python tools/bench_ingest.py --reads 8000000 --threads 16,16,16,16,16,16 2>&1 | summ
done
done
for one in "" 1 "" 1; do
  echo "== bench.py, PA_INGEST_ONE_STREAM=$one"
  PA_INGEST_ONE_STREAM=$one python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config5 --no-e2e 2> /tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ingest']['runs_Mreads_per_s'], d['ingest']['stages'], d['ingest'].get('two_lanes_one_gpu_reads_per_s'), d['ingest']['record_stream']['reads_per_s'])"
done
