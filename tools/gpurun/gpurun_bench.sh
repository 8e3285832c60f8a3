# usage: bash gpurun_bench.sh <tag> [bench args...]   (run on the GPU box through gpurun)
tag=$1; shift
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 900 python bench.py "$@" 2> gpurun_out/bench_$tag.err | tee gpurun_out/bench_$tag.json
grep -v "amdgpu.ids" gpurun_out/bench_$tag.err | tail -4
