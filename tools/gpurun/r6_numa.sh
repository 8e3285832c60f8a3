#!/bin/bash
# where the host memory of a copy lives and what that does to the link; then pa_process_reads five times with its stage seconds
R=$GRAFT_REPO_ROOT
echo "== topology"
for n in /sys/devices/system/node/node*; do echo "$(basename $n): cpus $(cat $n/cpulist) mem $(grep MemTotal $n/meminfo | awk '{print $4/1048576 " GB"}')"; done
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  cpuset: $(cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null)"
grep -i "cpus_allowed_list\|mems_allowed_list" /proc/self/status
for c in /sys/class/drm/card*/device; do [ -f $c/numa_node ] && echo "$c numa_node $(cat $c/numa_node) $(cat $c/vendor)"; done
echo "== h2d_numa"
timeout 300 $R/tools/microbench/h2d_numa
echo "== pa_process_reads"
cd $R
PA_VERBOSE=1 timeout 600 python tools/bench_ingest.py --reads 8000000 --threads 16,16,16,16,16,16 2>&1 | grep -v "Done Mapping" | grep "pa ingest\]\|value" | cut -c1-400
