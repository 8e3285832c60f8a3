# one development iteration on the GPU box: quick parity (fuzz + small.fq + scale 2 M), then A/B of bench.py under knobs
# usage: bash tools/gpurun/r2_iter.sh <tag> [ab configs...]
tag=$1; shift
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "small_fq or random_transcriptomes or many_classes or simulated or ragged or hot_classes or overflow or rccl or long_reads or node_traces" 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/${tag}_parity.txt
cat gpurun_out/${tag}_parity.txt
bash tools/gpurun/gpurun_ab.sh "$@" > gpurun_out/${tag}_ab.txt 2>&1
cat gpurun_out/${tag}_ab.txt
