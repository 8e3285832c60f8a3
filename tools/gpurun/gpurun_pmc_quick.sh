# quick PMC passes of the bench (no tests): usage: bash tools/gpurun/gpurun_pmc_quick.sh <tag> [ENV=..]
tag=${1:-q}; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="env $@ python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-config5 "
run() { name=$1; shift; timeout 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$name -- $B > $O/$name.log 2>&1; echo "$name rc=$?"; }
run pmc_sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY
run pmc_sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
run pmc_sq3 SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC
run pmc_l2 TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum
run pmc_fetch FETCH_SIZE
run pmc_write WRITE_SIZE
cd $R && python tools/pmc_summary.py $O
