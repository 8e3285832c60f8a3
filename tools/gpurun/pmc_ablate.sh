R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/abl; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for a in 0 2 1; do
  B="env PA_MAP_ABLATE=$a python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e"
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f$a -- $B > $O/f$a.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w$a -- $B > $O/w$a.log 2>&1
done
cd $R
python - <<'PY'
import csv,glob,collections
for a in (0,2,1):
    for k in ('f','w'):
        tot=collections.defaultdict(list)
        for f in glob.glob('gpurun_out/abl/%s%d/**/*counter_collection.csv'%(k,a), recursive=True):
            for r in csv.DictReader(open(f)):
                if 'pa_map_pool' in r.get('Kernel_Name',''):
                    tot[r['Counter_Name']].append(float(r['Counter_Value']))
        for c,v in tot.items():
            big=[x for x in v if x>0.5*max(v)]
            print('ablate',a,c,'avg over',len(big),'full launches: %.4g'%(sum(big)/len(big)))
PY
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
