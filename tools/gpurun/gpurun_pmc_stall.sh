# stall-analysis PMC passes (instruction cache, VMEM queue levels, LDS conflicts): usage: bash gpurun_pmc_stall.sh <tag> [ENV=..]
tag=${1:-stall}; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="env $@ python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "
run() { name=$1; shift; timeout 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$name -- $B > $O/$name.log 2>&1; echo "$name rc=$?"; }
run pmc_ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL
run pmc_lvl SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LEVEL_WAVES SQ_INST_LEVEL_LDS SQ_CYCLES
run pmc_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_ACTIVE_INST_VALU2 SQ_BUSY_CU_CYCLES
run pmc_tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TA_BUSY_avr
cd $R && python tools/pmc_summary.py $O
