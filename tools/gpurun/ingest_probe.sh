#!/bin/bash
# pa_process_reads on 8 M reads of config 3 (2.53 GB FASTQ in the page cache): window sizes and thread counts
for W in 134217728 67108864 33554432 16777216; do
  echo "== PA_INGEST_WINDOW=$W"
  PA_INGEST_WINDOW=$W python tools/bench_ingest.py --reads 8000000 --threads 16,16,16 2>&1 | grep -v "Done Mapping" | grep "pa ingest\] 8000000\|value" | tail -5
done
