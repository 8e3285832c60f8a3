#!/bin/bash
# the FASTQ leg inside bench.py's process: with and without the host-to-host leg before it (PA_VERBOSE: stage seconds and copy rates of every call)
cd $GRAFT_REPO_ROOT
for extra in "--no-e2e" ""; do
  echo "== bench.py --no-cpu-baseline --no-config5 $extra"
  PA_VERBOSE=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config5 $extra 2> /tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ingest']['runs_Mreads_per_s'], d['ingest']['stages'], d['ingest'].get('two_lanes_one_gpu_reads_per_s'))"
  grep "pa ingest\]" /tmp/err.txt | grep -v "Done Mapping" | cut -c1-330
done
