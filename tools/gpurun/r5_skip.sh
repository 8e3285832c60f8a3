mkdir -p gpurun_out
run() { PA_DICT_LOAD=$3 PA_PRODUCT_SO=$2 python bench.py --workload $1 --no-cpu-baseline --no-e2e --no-config5 --no-ingest --steps 10 --warmup 2 2>/tmp/e.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 $2 load $3 value=%.4e map_ms=%.3f' % (d['value'], r.get('map_pool_kernel_ms', r['kernel_ms'])), d.get('parity_sample'))"; }
python -m pytest tests -m gpu -x -q -k "parity or fuzz" 2>&1 | grep -E "passed|failed|rror|assert" | tail -5
for rep in 1 2; do for so in tools/baseline/serial.so tools/baseline/knobs.so; do for l in 0.25 0.5; do run config3 $so $l; run config5 $so $l; done; done; done
