# build tools/baseline/<name>.so = the product library compiled under extra flags, for same-box A/B runs (PA_PRODUCT_SO=...).
# Every variant is built with -DPA_DEBUG_KNOBS: the A/B knobs of DESIGN.md §8 (PA_MAP_ABLATE, PA_MAP_STATS, PA_POOL_SLOTS, ...)
# exist in these builds only, never in the shipped library.
# usage: bash tools/build_variant.sh <name> [-DPA_NT=5 ...]
set -e
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p tools/baseline _build/variant_$name
src=rust-pseudoaligner_amd/csrc
pids=""
for f in host_index.cpp dbg_build.cpp device_flatten.cpp synth.cpp fastq.cpp record_stream.cpp host_batch.cpp kernels.hip map_pool.hip device_index.hip collective.hip barcode_counts.hip index_build.hip index_fill.hip count_sort.hip resolve.hip render.hip fastq_scan.hip compact.hip; do
  extra=""; [ "$f" = "map_pool.hip" ] && extra="-mllvm -disable-machine-sink"   # (as rust-pseudoaligner_amd/_build.py: EXTRA_FLAGS)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread -Wall -Wno-unused-function -DPA_DEBUG_KNOBS $extra "$@" -x hip -c $src/$f -o _build/variant_$name/$f.o &
  pids="$pids $!"
done
for p in $pids; do wait $p || { echo "compile failed"; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread _build/variant_$name/*.o -ldl -lz -o tools/baseline/$name.so
echo tools/baseline/$name.so
