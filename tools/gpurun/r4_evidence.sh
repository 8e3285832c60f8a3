# round-4 evidence in one call: gpu tier + smoke + default bench line, rocprofv3 trace + PMC passes of configs 3 and 5
# (profile_r04.sh), full bench lines of the other workloads. usage: bash tools/gpurun/r4_evidence.sh <tag>
tag=${1:-ev}
O=gpurun_out/${tag}_full
mkdir -p $O
python -m pytest tests -m gpu -x -q --durations=8 2>&1 | grep -E "passed|failed|error|^[0-9.]+s " | tail -12 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1
timeout 1200 python bench.py 2> $O/bench.err > $O/bench.json; cut -c1-2500 $O/bench.json
bash tools/gpurun/profile_r04.sh ${tag}_config3 config3
bash tools/gpurun/profile_r04.sh ${tag}_config5 config5
for wl in config5 config2 config3k64; do
  timeout 900 python bench.py --workload $wl --no-e2e 2> gpurun_out/${tag}_bench_$wl.err > gpurun_out/${tag}_bench_$wl.json; cut -c1-300 gpurun_out/${tag}_bench_$wl.json
done
