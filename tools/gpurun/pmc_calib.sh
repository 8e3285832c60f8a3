# Counters of the calibration kernels (tools/microbench/pmc_calib.hip), one group per pass: FETCH_SIZE / WRITE_SIZE as rocprofv3 derives
# them, and the raw memory-side request counters by size (gfx950 has 32 / 64 / 128-byte read requests; FETCH_SIZE's formula is gfx942's)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/calib; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pass() { name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$name -- $R/tools/microbench/pmc_calib > $O/$name.log 2>&1; echo "$name rc=$?"
  f=$(find $O/$name -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-22s grid=%-8s %-28s %s" % (r["Kernel_Name"].split("(")[0][:22], r["Grid_Size"], r["Counter_Name"], r["Counter_Value"]))
PY
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
pass l2 TCC_MISS_sum TCC_HIT_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
