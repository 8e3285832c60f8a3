# full bench lines (with cpu_baseline) of every workload on one box. usage: bash tools/gpurun/bench_lines.sh <tag>
tag=${1:-bl}
timeout 900 python bench.py --steps 20 --warmup 5 2> gpurun_out/${tag}_bench_config3.err > gpurun_out/${tag}_bench_config3.json; cut -c1-200 gpurun_out/${tag}_bench_config3.json
for wl in config5 config2 config3k64; do
  timeout 900 python bench.py --workload $wl --no-e2e 2> gpurun_out/${tag}_bench_$wl.err > gpurun_out/${tag}_bench_$wl.json; cut -c1-200 gpurun_out/${tag}_bench_$wl.json
done
