#!/bin/bash
# round 6 evidence: for every workload the bench line of this box, the rocprofv3 kernel trace summary and the PMC passes (each in its own
# run: --pmc is never combined with other trace domains). usage: bash tools/gpurun/profile_r06.sh [workload ...]
R=$GRAFT_REPO_ROOT
WLS="${@:-config3 config5 config3r config3k64 config2}"
for wl in $WLS; do
  O=$R/gpurun_out/r06_$wl
  mkdir -p $O
  extra=""
  [ "$wl" = "config2" ] && extra="--batch 100000000"
  cd /tmp && export TMPDIR=/tmp
  B="python $R/bench.py --workload $wl $extra --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --no-config5 --no-ingest"
  $B > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B > $O/trace.log 2>&1; echo "$wl trace rc=$?"
  run() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$name -- $B > $O/$name.log 2>&1; echo "$wl $name rc=$?"; }
  run pmc_sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY
  run pmc_sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
  run pmc_l2 TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum
  run pmc_fetch FETCH_SIZE
  run pmc_write WRITE_SIZE
  run pmc_rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
  run pmc_wrreq TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum TCC_READ_sum
  cd $R
  python tools/rocprof_summary.py $O/trace "$B" > $O/kernel_trace_summary.txt 2>&1
  python tools/pmc_summary.py $O > $O/pmc.txt 2>&1
  head -6 $O/kernel_trace_summary.txt | cut -c1-170
  # keep only the summaries (the raw csv trees are large)
  find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
done
