# round-5 development iteration: GPU test tier (optionally a -k subset; "none" skips it), then bench lines of configs 3, 5 and 2
# (kernel times from the library's HIP events), optional A/B against older builds: usage: bash tools/gpurun/r5_iter.sh <tag> [pytest -k expr|none] [baseline.so ...]
tag=$1; kexpr=$2; shift; shift
mkdir -p gpurun_out
if [ "$kexpr" != "none" ]; then
  if [ -n "$kexpr" ]; then python -m pytest tests -m gpu -x -q -k "$kexpr" 2>&1 | tail -8 > gpurun_out/${tag}_pytest.txt
  else python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/${tag}_pytest.txt; fi
  cat gpurun_out/${tag}_pytest.txt
fi
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-10s %-28s value=%.4e ms_step=%.3f map_ms=%.3f resolve_ms=%.3f count_ms=%.3f' % (sys.argv[1], sys.argv[2], d['value'], d['ms_per_step'], r.get('map_pool_kernel_ms', r['kernel_ms']), r.get('resolve_kernel_ms', 0), r.get('count_kernels_ms', 0)))" "$1" "$2"; }
for so in "" "$@"; do
  for wl in config3 config5 config2; do
    env ${so:+PA_PRODUCT_SO=$so} python bench.py --workload $wl --no-cpu-baseline --no-e2e --no-config5 --no-ingest --steps 8 --warmup 2 2>gpurun_out/${tag}_$wl.err | line $wl "${so:-HEAD}" | tee -a gpurun_out/${tag}_bench.txt
  done
done
