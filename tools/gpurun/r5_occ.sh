# new probe protocol (seek1) and 4 workgroups per CU (occ4: 128 registers, pools of 96 slots) against HEAD's build; same box
mkdir -p gpurun_out
run() { env $3 PA_PRODUCT_SO=$2 python bench.py --workload $1 --no-cpu-baseline --no-e2e --no-config5 --no-ingest --steps 10 --warmup 2 2>/tmp/e.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 %-24s %-26s value=%.4e ms_step=%.3f map_ms=%.3f' % ('$2', '$3', d['value'], d['ms_per_step'], r.get('map_pool_kernel_ms', r['kernel_ms'])), d.get('parity_sample'))"; }
PA_PRODUCT_SO=tools/baseline/seek1.so python -m pytest tests -m gpu -x -q -k "parity and not scale" 2>&1 | grep -E "passed|failed|rror|assert" | tail -5
PA_MAP_BLOCKS_PER_CU=4 PA_PRODUCT_SO=tools/baseline/occ4.so python -m pytest tests -m gpu -x -q -k "parity and not scale" 2>&1 | grep -E "passed|failed|rror|assert" | tail -5
for rep in 1 2; do for wl in config3 config5 config2; do
  run $wl tools/baseline/knobs.so X=0
  run $wl tools/baseline/seek1.so X=0
  run $wl tools/baseline/occ4.so PA_MAP_BLOCKS_PER_CU=4
  run $wl tools/baseline/occ4.so PA_MAP_BLOCKS_PER_CU=3
done; done
