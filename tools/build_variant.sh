#!/bin/bash
# build an experiment variant of the library: tools/build_variant.sh <name> [extra hipcc flags for tower_kernels.hip / mcts_kernels.hip ...]
# -> muzero_amd/lib/exp/<name>.so (select it with MUZERO_AMD_LIB=...).  Only the named source is recompiled.
set -e
name=$1; src=${2:-tower_kernels}; shift; shift || true
cd "$(dirname "$0")/../muzero_amd/csrc"
make -s -j8 >/dev/null
mkdir -p ../lib/exp ../lib/exp/obj
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt"
/opt/rocm/bin/hipcc $FLAGS "$@" -c $src.hip -o ../lib/exp/obj/$name.o
objs=$(ls ../lib/obj/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../lib/exp/$name.so $objs ../lib/exp/obj/$name.o
echo built ../lib/exp/$name.so
