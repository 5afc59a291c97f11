#!/bin/bash
# A/B of tree-kernel builds on ONE box: tools/tree_ab.sh <parity name or -> <names...>   (variants from tools/build_variant.sh <name> mcts_kernels)
set -u
mkdir -p gpurun_out/r2
par=$1; shift
if [ "$par" != "-" ]; then
  echo "== parity $par"
  timeout -k 10 600 env MUZERO_AMD_LIB=$PWD/muzero_amd/lib/exp/$par.so python -m pytest tests/test_gpu_mcts.py tests/test_gpu_search.py tests/test_gpu_selfplay.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -3 || exit 1
fi
for pass in 1 2; do
  for n in "$@"; do
    timeout -k 10 200 env MUZERO_AMD_LIB=$PWD/muzero_amd/lib/exp/$n.so python tools/tree_bench.py 4096 50 2>&1 | grep -v amdgpu.ids | sed "s/^/$n pass $pass: /" || exit 1
  done
done
