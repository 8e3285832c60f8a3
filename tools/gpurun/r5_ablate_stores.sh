for rep in 1 2; do for a in ${ABL:-0 1}; do PA_MAP_ABLATE=$a PA_PRODUCT_SO=tools/baseline/knobs.so python bench.py --workload config3 --no-cpu-baseline --no-e2e --no-config5 --no-ingest --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('ablate $a map_ms=%.3f' % r.get('map_pool_kernel_ms', r['kernel_ms']))"; done; done
