for cfg in "PA_E2E_CHUNK=20000000 PA_E2E_STREAMS=3" "PA_E2E_CHUNK=10000000 PA_E2E_STREAMS=3" "PA_E2E_CHUNK=5000000 PA_E2E_STREAMS=3" "PA_E2E_CHUNK=5000000 PA_E2E_STREAMS=4" "PA_E2E_CHUNK=2500000 PA_E2E_STREAMS=4" "PA_E2E_CHUNK=10000000 PA_E2E_STREAMS=2" "PA_E2E_CHUNK=20000000 PA_E2E_STREAMS=3"; do
  env $cfg python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-50s e2e=%.3e ms=%.1f frac=%.3f' % ('$cfg', d['e2e_reads_per_s'], d['e2e']['ms'], d['e2e_pcie_frac']))"
done
