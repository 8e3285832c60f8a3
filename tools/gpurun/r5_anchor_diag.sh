# why do anchors cost time? footprint microbench + scheduler statistics with and without the table
tag=${1:-r5b}
mkdir -p gpurun_out
./tools/microbench/footprint > gpurun_out/${tag}_footprint.txt 2>&1; cat gpurun_out/${tag}_footprint.txt
for ratio in 0 0.5; do
  for wl in config3; do
    echo "== $wl ratio $ratio"
    PA_ANCHOR_RATIO=$ratio PA_PRODUCT_SO=tools/baseline/knobs.so PA_MAP_STATS=1 python bench.py --workload $wl --no-cpu-baseline --no-e2e --no-config5 --no-ingest --steps 3 --warmup 1 2> /tmp/stats.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value=%.3e ms_step=%.3f kernel_ms=%.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"
    grep "pa map stats" /tmp/stats.err | sed -n 2p
  done
done 2>&1 | tee gpurun_out/${tag}_stats.txt
