# round 5, first GPU call: the GPU tier (incl. the full-size bit-exact comparisons), the counter calibration, and the evidence files the
# round-4 review asked for as FILES: scheduler statistics, cache-resident floor, which request stream costs what (ablations, knobs build)
tag=${1:-r5a}
R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out
( time python -m pytest tests -m gpu -x -q --durations=10 ) 2>&1 | grep -E "passed|failed|error|Error|^[0-9.]+s |real|bit-exact" | tail -25 > gpurun_out/${tag}_pytest.txt; cat gpurun_out/${tag}_pytest.txt
bash tools/gpurun/pmc_calib.sh > gpurun_out/${tag}_calib.txt 2>&1; tail -60 gpurun_out/${tag}_calib.txt
bash tools/gpurun/gpurun_stats.sh "--workload config3" "--workload config5" "--workload config2" > gpurun_out/${tag}_stats.txt 2>&1; cat gpurun_out/${tag}_stats.txt
{ for wl in config3 config5; do for lim in 0 20; do
  echo "== $wl PA_SIM_TX_LIMIT=$lim"
  env PA_PRODUCT_SO=tools/baseline/knobs.so PA_SIM_TX_LIMIT=$lim python bench.py --workload $wl --no-cpu-baseline --no-e2e --no-config5 --no-ingest --steps 5 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('value=%.4e ms_step=%.3f map_ms=%.3f resolve_ms=%.3f' % (d['value'], d['ms_per_step'], r.get('map_pool_kernel_ms', r['kernel_ms']), r.get('resolve_kernel_ms', 0)))"
done; done; } > gpurun_out/${tag}_txlimit.txt 2>&1; cat gpurun_out/${tag}_txlimit.txt
# request streams: L2 misses and memory-side read requests by size, with result stores / keys ablated (knobs build)
O=$R/gpurun_out/${tag}_abl; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for a in 0 1 2 3; do
  B="env PA_PRODUCT_SO=$R/tools/baseline/knobs.so PA_MAP_ABLATE=$a python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-config5 --no-ingest"
  timeout 600 rocprofv3 --kernel-trace --pmc TCC_MISS_sum TCC_HIT_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/l2_$a -- $B > $O/l2_$a.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $O/rd_$a -- $B > $O/rd_$a.log 2>&1
done
cd $R
python - $tag <<'PY' > gpurun_out/${tag}_ablate.txt 2>&1
import csv,glob,collections,sys
tag=sys.argv[1]
for a in (0,1,2,3):
    for k in ('l2','rd'):
        tot=collections.defaultdict(list)
        for f in glob.glob('gpurun_out/%s_abl/%s_%d/**/*counter_collection.csv'%(tag,k,a), recursive=True):
            for r in csv.DictReader(open(f)):
                if 'pa_map_pool' in r.get('Kernel_Name',''):
                    tot[r['Counter_Name']].append(float(r['Counter_Value']))
        for c,v in sorted(tot.items()):
            big=[x for x in v if x>0.5*max(v)] if max(v)>0 else v
            print('PA_MAP_ABLATE=%d %-28s avg over %d full launches: %.5g'%(a,c,len(big),sum(big)/len(big)))
PY
cat gpurun_out/${tag}_ablate.txt
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
