mkdir -p gpurun_out
run() { PA_DICT_LOAD=$2 PA_PRODUCT_SO=tools/baseline/knobs.so python bench.py --workload $1 --no-cpu-baseline --no-e2e --no-config5 --no-ingest --steps 8 --warmup 2 2>/tmp/e.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 load $2 value=%.4e map_ms=%.3f' % (d['value'], r.get('map_pool_kernel_ms', r['kernel_ms'])), d.get('parity_sample'))"; grep -h "device index" /tmp/e.err | tail -1; }
for l in 0.333 0.2 0.125 0.333; do run config3k64 $l; done
for l in 0.5 0.25 0.125 0.5; do run config2 $l; done
for l in 0.25 0.167 0.125; do run config3 $l; run config5 $l; done
