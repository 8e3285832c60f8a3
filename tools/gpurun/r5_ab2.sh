# same-box A/B of variant builds: usage: bash tools/gpurun/r5_ab2.sh <tag> "<pytest -k expr|none>" so1 so2 ...
tag=$1; kexpr=$2; shift; shift
mkdir -p gpurun_out
run() { PA_PRODUCT_SO=$2 python bench.py --workload $1 --no-cpu-baseline --no-e2e --no-config5 --no-ingest --steps 10 --warmup 2 2>/tmp/e.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 %-28s value=%.4e ms_step=%.3f map_ms=%.3f' % ('$2', d['value'], d['ms_per_step'], r.get('map_pool_kernel_ms', r['kernel_ms'])), d.get('parity_sample'))"; }
if [ "$kexpr" != "none" ]; then python -m pytest tests -m gpu -x -q -k "$kexpr" 2>&1 | grep -E "passed|failed|rror|assert" | tail -5; fi
for rep in 1 2; do for so in "$@"; do for wl in ${WLS:-config3 config5 config2}; do run $wl $so; done; done; done
