# A/B runs of bench.py under tuning knobs, one host index build shared through --index-cache
# usage: bash gpurun_ab.sh "ENV=.. [--bench-flag ..]" ...
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-e2e --no-config5 --steps 6 --warmup 2"
for cfg in "$@"; do
  envs=""; flags=""
  for tok in $cfg; do case "$tok" in --*) flags="$flags $tok";; *=*) envs="$envs $tok";; *) flags="$flags $tok";; esac; done
  env $envs $B --index-cache /tmp/g_$(echo $flags | tr -d " -").idx $flags 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-60s value=%.3e ms_step=%.3f kernel_ms=%.3f' % ('$cfg', d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"
done
