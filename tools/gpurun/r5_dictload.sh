# does a denser dictionary (smaller footprint: address translation reach ~4 GB, profiles/r05_tlb_footprint.txt) pay? knobs build, same box
tag=${1:-r5dl}
mkdir -p gpurun_out
for load in ${LOADS:-0.5 0.4 0.33 0.25 0.2 0.5}; do
  for wl in config3 config5; do
    PA_DICT_LOAD=$load PA_PRODUCT_SO=tools/baseline/knobs.so python bench.py --workload $wl --no-cpu-baseline --no-e2e --no-config5 --no-ingest --steps 8 --warmup 2 2>/tmp/e.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$wl load $load value=%.4e map_ms=%.3f' % (d['value'], r.get('map_pool_kernel_ms', r['kernel_ms'])), d.get('parity_sample'))"
    grep -h "device index" /tmp/e.err | tail -1
  done
done 2>&1 | tee gpurun_out/${tag}.txt
