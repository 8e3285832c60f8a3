#!/bin/bash
# timeline of the host-to-host leg of bench.py (pa_map_tiles_host): rocprofv3 kernel + memory-copy trace
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --memory-copy-trace -d $R/gpurun_out/e2e_prof -o e2e -- python $R/bench.py --steps 1 --warmup 0 --no-ingest --no-cpu-baseline --no-config5 2>&1 | grep -v simple_timer | tail -2 | cut -c1-200
