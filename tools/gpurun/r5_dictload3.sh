mkdir -p gpurun_out
run() { PA_DICT_LOAD=$2 PA_PRODUCT_SO=tools/baseline/knobs.so python bench.py --workload $1 --no-cpu-baseline --no-e2e --no-config5 --no-ingest --steps 10 --warmup 2 2>/tmp/e.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 load $2 value=%.4e map_ms=%.3f' % (d['value'], r.get('map_pool_kernel_ms', r['kernel_ms'])), d.get('parity_sample'))"; }
for rep in 1 2 3; do for l in 0.5 0.25; do run config3 $l; run config5 $l; done; done
