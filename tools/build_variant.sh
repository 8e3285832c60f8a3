# build tools/baseline/<name>.so = the product library with map_pool.hip compiled under extra flags (A/B runs: PA_PRODUCT_SO)
# usage: bash tools/build_variant.sh <name> [-DPA_NT=5 ...]
set -e
name=$1; shift
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > /dev/null
mkdir -p tools/baseline _build
obj=rust-pseudoaligner_amd/_build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread -Wall -Wno-unused-function "$@" -x hip -c rust-pseudoaligner_amd/csrc/map_pool.hip -o _build/map_pool_$name.o
objs=$(ls $obj/*.o | grep -v map_pool.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread $objs _build/map_pool_$name.o -ldl -lz -o tools/baseline/$name.so
echo tools/baseline/$name.so
