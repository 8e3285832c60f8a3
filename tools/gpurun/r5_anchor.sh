# anchors A/B: parity subset, then bench lines of configs 3, 5, 2 at several table sizes (PA_ANCHOR_RATIO=0: no anchors)
tag=${1:-r5a}; kexpr=${2:-"parity or fuzz or emu"}; ratios=${3:-"0 0.5 1 2"}
mkdir -p gpurun_out
if [ "$kexpr" != "none" ]; then
  python -m pytest tests -m gpu -x -q -k "$kexpr" 2>&1 | grep -E "passed|failed|rror|assert" | tail -8 > gpurun_out/${tag}_pytest.txt; cat gpurun_out/${tag}_pytest.txt
fi
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-8s ratio %-4s value=%.4e ms_step=%.3f map_ms=%.3f parity=%s' % (sys.argv[1], sys.argv[2], d['value'], d['ms_per_step'], r.get('map_pool_kernel_ms', r['kernel_ms']), d.get('parity_sample')))" "$1" "$2"; }
for ratio in $ratios; do
  for wl in config3 config5 config2; do
    PA_VERBOSE=1 PA_ANCHOR_RATIO=$ratio python bench.py --workload $wl --no-cpu-baseline --no-e2e --no-config5 --no-ingest --steps 8 --warmup 2 2>gpurun_out/${tag}_${wl}_$ratio.err | line $wl $ratio | tee -a gpurun_out/${tag}_bench.txt
    grep -h "pa index\] anchors" gpurun_out/${tag}_${wl}_$ratio.err | tail -1 | tee -a gpurun_out/${tag}_bench.txt
  done
done
