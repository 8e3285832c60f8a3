# FETCH_SIZE / WRITE_SIZE of the calibration kernels (tools/microbench/pmc_calib.hip), one counter per pass
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/calib; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -- $R/tools/microbench/pmc_calib > $O/$c.log 2>&1; echo "$c rc=$?"
  f=$(find $O/$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$c" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if r.get("Counter_Name") == sys.argv[2]:
        print("%-14s %s = %s KB" % (r["Kernel_Name"].split("(")[0], sys.argv[2], r["Counter_Value"]))
PY
done
