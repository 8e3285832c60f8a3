# scheduler statistics (knobs build) + instruction counters of variants; usage: bash tools/gpurun/r5_stats.sh <tag> [so ...]
tag=$1; shift
bash tools/gpurun/gpurun_stats.sh "--workload config3" "--workload config5" > gpurun_out/${tag}_stats.txt 2>&1; cat gpurun_out/${tag}_stats.txt
bash tools/gpurun/r5_insts.sh $tag "$@"
