# Full evidence run of one round (on the GPU box through gpurun): tests, bench, rocprofv3 kernel trace + PMC passes.
# usage: bash tools/gpurun/profile_round.sh <tag>
tag=${1:-r01}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -2 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1
timeout 900 python bench.py --index-cache /tmp/g.idx 2> $O/bench.err > $O/bench.json; cat $O/bench.json | cut -c1-300
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --index-cache /tmp/g.idx"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B > $O/trace.log 2>&1; echo "trace rc=$?"
run() { name=$1; shift; timeout 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$name -- $B > $O/$name.log 2>&1; echo "$name rc=$?"; }
run pmc_fetch FETCH_SIZE
run pmc_write WRITE_SIZE
run pmc_l2 TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum
run pmc_sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY
run pmc_sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
find $O -name "*.csv" | wc -l
