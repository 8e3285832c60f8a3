# instruction counters of the map kernel for a list of builds ("" = the in-tree library), configs 3 and 5: one PMC pass each
# usage: bash tools/gpurun/r5_insts.sh <tag> [so ...]
tag=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${tag}_insts; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for so in "" "$@"; do
  for wl in config3 config5; do
    B="env ${so:+PA_PRODUCT_SO=$R/$so} python $R/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-config5 --no-ingest"
    timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/p${i}_$wl -- $B > $O/p${i}_$wl.log 2>&1
    timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $O/q${i}_$wl -- $B > $O/q${i}_$wl.log 2>&1
    echo "build=${so:-HEAD} $wl" >> $O/index.txt; echo "p${i}_$wl q${i}_$wl" >> $O/index.txt
  done
  i=$((i+1))
done
cd $R
python - $tag "$@" <<'PY' | tee gpurun_out/${tag}_insts.txt
import csv,glob,collections,sys
tag=sys.argv[1]; sos=["HEAD"]+sys.argv[2:]
for i,so in enumerate(sos):
    for wl in ("config3","config5"):
        tot=collections.defaultdict(list)
        for k in ("p","q"):
            for f in glob.glob('gpurun_out/%s_insts/%s%d_%s/**/*counter_collection.csv'%(tag,k,i,wl), recursive=True):
                for r in csv.DictReader(open(f)):
                    if 'pa_map_pool' in r.get('Kernel_Name',''):
                        tot[r['Counter_Name']].append(float(r['Counter_Value']))
        out=[]
        for c,v in sorted(tot.items()):
            big=[x for x in v if x>0.5*max(v)] if max(v)>0 else v
            out.append('%s=%.4g'%(c,sum(big)/len(big)))
        print('%-28s %-8s %s'%(so,wl,' '.join(out)))
PY
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
