# round-2 evidence in one call: gpu tier + smoke + default bench line + 2-rank rehearsal (r2_full.sh), rocprofv3 trace + PMC of
# configs 3 and 5 (profile_r02.sh), full bench lines (with cpu_baseline) of the other workloads. usage: bash tools/gpurun/r2_evidence.sh <tag>
tag=${1:-ev}
bash tools/gpurun/r2_full.sh ${tag}_full
bash tools/gpurun/profile_r02.sh ${tag}_config3 config3
bash tools/gpurun/profile_r02.sh ${tag}_config5 config5
for wl in config5 config2 config3k64; do
  timeout 900 python bench.py --workload $wl --no-e2e 2> gpurun_out/${tag}_bench_$wl.err > gpurun_out/${tag}_bench_$wl.json; cut -c1-300 gpurun_out/${tag}_bench_$wl.json
done
