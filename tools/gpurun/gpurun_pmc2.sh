cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --index-cache /tmp/g.idx"
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$name -- $B > $O/$name.log 2>&1; echo "$name rc=$?"; }
PA_MAP_STATS=1 $B 2>&1 | grep "map stats" | tail -2
run p1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU
run p2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
run p3 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
run p4 FETCH_SIZE
run p5 WRITE_SIZE
