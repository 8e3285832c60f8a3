# TA / TCP / TD busy and stall counters of the mapping kernel (is the vector-memory front end the limit?). usage: bash tools/gpurun/pmc_ta.sh <tag> [ENV=.. --flags]
tag=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "\b\(TA\|TCP\|TD\|SQ\|TCC\|GRBM\)_[A-Za-z0-9_]*" | sort -u > $O/avail.txt; wc -l $O/avail.txt
envs=""; flags=""
for tok in "$@"; do case "$tok" in --*) flags="$flags $tok";; *=*) envs="$envs $tok";; *) flags="$flags $tok";; esac; done
envs=$(echo "$envs" | sed "s#PA_PRODUCT_SO=tools#PA_PRODUCT_SO=$R/tools#")
B="env $envs python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-config5 $flags"
run() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$name -- $B > $O/$name.log 2>&1; echo "$name rc=$?"; }
run pmc_ta1 TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE
#run pmc_ta2 (hangs rocprofv3 on this pool) TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
run pmc_tcp1 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum
run pmc_tcp2 TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
#run pmc_td (hangs rocprofv3 on this pool) TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum
run pmc_tlb1 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum
run pmc_tlb2 TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_LFIFO_NO_RES_sum
run pmc_tcp3 TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
run pmc_tcp4 TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TA_TOTAL_WAVEFRONTS_sum
cd $R
python tools/pmc_summary.py $O > $O/pmc.txt 2>&1
grep "pa_map_pool" $O/pmc.txt
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
