# round-4 development iteration: GPU test tier (optionally a -k subset), then bench lines of config 3 and config 5
# usage: bash tools/gpurun/r4_iter.sh <tag> [pytest -k expression]
tag=$1; kexpr=$2
mkdir -p gpurun_out
if [ -n "$kexpr" ]; then
  python -m pytest tests -m gpu -x -q -k "$kexpr" 2>&1 | tail -15 > gpurun_out/${tag}_pytest.txt
else
  python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/${tag}_pytest.txt
fi
cat gpurun_out/${tag}_pytest.txt
timeout 900 python bench.py --no-e2e 2> gpurun_out/${tag}_c3.err > gpurun_out/${tag}_c3.json; cut -c1-1200 gpurun_out/${tag}_c3.json; tail -3 gpurun_out/${tag}_c3.err
timeout 900 python bench.py --workload config5 --no-e2e 2> gpurun_out/${tag}_c5.err > gpurun_out/${tag}_c5.json; cut -c1-600 gpurun_out/${tag}_c5.json; tail -3 gpurun_out/${tag}_c5.err
