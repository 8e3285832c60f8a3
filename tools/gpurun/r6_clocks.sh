#!/bin/bash
# the GPU's clocks and power while the map kernel runs back to back (bench.py --steps 400 = 3 s of launches), sampled every 0.25 s
cd $GRAFT_REPO_ROOT
(for i in $(seq 1 120); do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|fclk\|socclk\|Power" | tr -s ' ' | tr '\n' ';'; echo; sleep 0.25; done) > gpurun_out/r6_clock_samples.txt &
SM=$!
python bench.py --steps 400 --warmup 2 --no-cpu-baseline --no-e2e --no-config5 --no-ingest 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.3e ms_step %.3f kernel_ms %.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"
kill $SM 2>/dev/null
sort gpurun_out/r6_clock_samples.txt | uniq -c | sort -rn | head -12 | cut -c1-300
