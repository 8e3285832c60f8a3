# round-5 evidence in one call: gpu tier + smoke + default bench line, rocprofv3 trace + PMC passes of configs 3 and 5 (profile_r05.sh: now with
# the memory-side request counters by size), full bench lines of the other workloads, profiles/latest_pmc.json. usage: bash tools/gpurun/r5_evidence.sh <tag>
tag=${1:-r5ev}
O=gpurun_out/${tag}_full
mkdir -p $O
python -m pytest tests -m gpu -x -q --durations=6 2>&1 | grep -E "passed|failed|rror|^[0-9.]+s " | tail -10 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1
bash tools/gpurun/profile_r05.sh ${tag}_config3 config3
bash tools/gpurun/profile_r05.sh ${tag}_config5 config5
python tools/make_latest_pmc.py config3=gpurun_out/${tag}_config3/pmc.txt config5=gpurun_out/${tag}_config5/pmc.txt && cp profiles/latest_pmc.json $O/latest_pmc.json
timeout 1500 python bench.py 2> $O/bench.err > $O/bench.json; cut -c1-1500 $O/bench.json
for wl in config5 config2 config3k64; do
  timeout 900 python bench.py --workload $wl --no-e2e 2> gpurun_out/${tag}_bench_$wl.err > gpurun_out/${tag}_bench_$wl.json; cut -c1-300 gpurun_out/${tag}_bench_$wl.json
done
