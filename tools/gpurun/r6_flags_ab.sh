#!/bin/bash
# compiler-flag variants of map_pool.hip (tools/baseline/<name>.so, everything else from the base variant) against the base, same box, two passes
cd $GRAFT_REPO_ROOT
for pass in 1 2; do
for v in ${VARIANTS:-base nosink nosink2 nolicm nopm phi4 phi8 notd nsphi4 nsnolicm base}; do
  bash tools/gpurun/gpurun_ab.sh "PA_PRODUCT_SO=$GRAFT_REPO_ROOT/tools/baseline/$v.so" 2>&1 | tail -1 | sed "s#$GRAFT_REPO_ROOT/##"
done
done
