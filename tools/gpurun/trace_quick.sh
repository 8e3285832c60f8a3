# per-kernel times of one bench run (rocprofv3 --kernel-trace --stats): usage: bash tools/gpurun/trace_quick.sh <tag> [bench flags / ENV=..]
tag=${1:-t}; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
envs=""; flags=""
for tok in "$@"; do case "$tok" in --*) flags="$flags $tok";; *=*) envs="$envs $tok";; *) flags="$flags $tok";; esac; done
cd /tmp && export TMPDIR=/tmp
timeout 600 env $envs rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --no-config5 $flags > $O/trace.log 2>&1
echo "trace rc=$?"
cd $R && python tools/rocprof_summary.py $O/trace $flags | tee $O/kernel_trace_summary.txt | cut -c1-150 | head -16
