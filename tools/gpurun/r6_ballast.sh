#!/bin/bash
# what in bench.py's process slows the windows' copies: device memory held, pinned memory held?
cd $GRAFT_REPO_ROOT
summ() { grep -v "Done Mapping" | grep "windows to\|value\|Error\|error" | cut -c1-200 | sed 's/"metric": "reads.sec FASTQ text -> Debug tuples (pa_process_reads, .dev.null)", //g; s/"unit": "reads.s", "threads": 16, "reads": 8000000, //g; s/\[pa ingest\] windows to the GPU: 2013.3 MB in//; s/"seconds".*//' | awk '{printf "%s | ", $0} END {print ""}'; }
for v in "" "--ballast-device-gb 24" "--ballast-device-gb 24 --ballast-touch" "--ballast-pinned-gb 8" "--ballast-device-gb 100 --ballast-touch" ""; do
  echo "== $v"
  PA_VERBOSE=1 python tools/bench_ingest.py --reads 8000000 --threads 16,16,16,16,16 $v 2>&1 | summ
done
