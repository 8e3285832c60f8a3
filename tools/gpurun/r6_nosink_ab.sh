#!/bin/bash
# -mllvm -disable-machine-sink on map_pool.hip against the base build: config 3 / 5 / 3r, interleaved
cd $GRAFT_REPO_ROOT
for wl in config3 config5 config3r; do
for pass in 1 2 3; do
for v in base nosink; do
  bash tools/gpurun/gpurun_ab.sh "PA_PRODUCT_SO=$GRAFT_REPO_ROOT/tools/baseline/$v.so --workload $wl" 2>&1 | tail -1 | sed "s#$GRAFT_REPO_ROOT/##"
done
done
done
