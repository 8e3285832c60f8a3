for wl in config3 config5; do echo "== $wl"; PA_PRODUCT_SO=tools/baseline/knobs.so PA_MAP_STATS=1 python bench.py --workload $wl --no-cpu-baseline --no-e2e --no-config5 --no-ingest --steps 3 --warmup 1 2>/tmp/stats.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value=%.3e kernel_ms=%.3f' % (d['value'], d['roofline']['kernel_ms']))"; grep "pa map stats" /tmp/stats.err | sed -n 2p; done
