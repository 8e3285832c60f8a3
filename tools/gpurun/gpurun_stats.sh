# scheduler statistics of bench.py under tuning knobs: usage: bash gpurun_stats.sh "ENV=.. [--bench-flag ..]" ...
mkdir -p gpurun_out
for cfg in "$@"; do
  envs=""; flags=""
  for tok in $cfg; do case "$tok" in --*) flags="$flags $tok";; *=*) envs="$envs $tok";; *) flags="$flags $tok";; esac; done
  echo "== $cfg"
  env PA_PRODUCT_SO=tools/baseline/knobs.so $envs PA_MAP_STATS=1 python bench.py --no-cpu-baseline --no-e2e --no-config5 --index-cache /tmp/g_$(echo $flags | tr -d ' -').idx --steps 3 --warmup 1 $flags 2> /tmp/stats.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value=%.3e ms_step=%.3f kernel_ms=%.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"
  grep "pa map stats" /tmp/stats.err | sed -n 2p
done
