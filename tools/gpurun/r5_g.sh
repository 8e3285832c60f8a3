tag=r5g
python -m pytest tests -m gpu -x -q -k "process_reads or record_stream or c_client" 2>&1 | grep -E "passed|failed|rror|assert" | tail -8 > gpurun_out/${tag}_pytest.txt; cat gpurun_out/${tag}_pytest.txt
python tools/gpu_fastq_fuzz.py 60 2>&1 | tail -3
for cfg in "2000000 4" "4000000 3" "5000000 3" "10000000 3"; do set -- $cfg
  PA_E2E_CHUNK=$1 PA_E2E_STREAMS=$2 python bench.py --no-config5 --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('chunk $1 streams $2: e2e %.4g reads/s, %.1f ms; ingest %.4g (%s) rs %.4g' % (d.get('e2e_reads_per_s') or 0, (d.get('e2e') or {}).get('ms',0), d.get('ingest_reads_per_s') or 0, (d.get('ingest') or {}).get('stages'), ((d.get('ingest') or {}).get('record_stream') or {}).get('reads_per_s',0)), d.get('e2e_error'), d.get('ingest_error'))"
done
for so in "" tools/baseline/rare16.so tools/baseline/rare24.so; do
  for wl in config5 config2 config3; do
    env ${so:+PA_PRODUCT_SO=$so} python bench.py --workload $wl --no-cpu-baseline --no-e2e --no-config5 --no-ingest --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-10s %-28s value=%.4e map_ms=%.3f' % ('$wl', '${so:-HEAD}', d['value'], r.get('map_pool_kernel_ms', r['kernel_ms'])))"
  done
done
