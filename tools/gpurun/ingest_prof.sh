#!/bin/bash
# per-kernel time of the window pipeline (pa_process_reads): rocprofv3 kernel + memory-copy (+ HIP API with "api") trace of tools/bench_ingest.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
EXTRA=""
[ "$1" = "api" ] && EXTRA="--hip-runtime-trace"
rocprofv3 --kernel-trace --memory-copy-trace $EXTRA --stats -d $R/gpurun_out/ingest_prof -o ingest -- python $R/tools/bench_ingest.py --reads 8000000 --threads 16,16 2>&1 | grep -v "Done Mapping" | grep -v simple_timer | tail -8
