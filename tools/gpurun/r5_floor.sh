# cache-resident floor of the map kernel (reads drawn from 20 transcripts: every table of the index is cache-resident) against the full workload
for wl in config3 config5; do for lim in 0 20; do
  echo "== $wl PA_SIM_TX_LIMIT=$lim"
  env PA_PRODUCT_SO=tools/baseline/knobs.so PA_SIM_TX_LIMIT=$lim python bench.py --workload $wl --no-cpu-baseline --no-e2e --no-config5 --no-ingest --steps 5 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('value=%.4e ms_step=%.3f map_ms=%.3f resolve_ms=%.3f' % (d['value'], d['ms_per_step'], r.get('map_pool_kernel_ms', r['kernel_ms']), r.get('resolve_kernel_ms', 0)))"
done; done
