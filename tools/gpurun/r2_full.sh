# full check of a round-2 build on the GPU box: whole gpu tier, smoke, the default bench line, a 2-rank rehearsal of bench.py
# on one GPU (gloo: RCCL refuses two ranks on one device), usage: bash tools/gpurun/r2_full.sh <tag>
tag=${1:-full}
O=gpurun_out/$tag
mkdir -p $O
python -m pytest tests -m gpu -x -q --durations=8 2>&1 | grep -E "passed|failed|error|^[0-9.]+s " | tail -12 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1
timeout 1200 python bench.py  2> $O/bench.err > $O/bench.json; cat $O/bench.json | cut -c1-2000
PA_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --batch 10000000 --no-cpu-baseline 2> $O/bench_world2.err > $O/bench_world2.json; echo "world2 rc=$?"; cut -c1-600 $O/bench_world2.json; tail -3 $O/bench_world2.err
