#!/bin/bash
# A/B of experiment builds (muzero_amd/lib/exp/<name>.so) on ONE box: tools/tower_ab.sh <parity names, comma separated or -> <names...>
# parity test (tower + heads vs torch) for the first list, then back-to-back timing (two passes) and phase stamps for all.
set -u
mkdir -p gpurun_out/r2
par=$1; shift
if [ "$par" != "-" ]; then
  for n in ${par//,/ }; do
    echo "== parity $n"
    timeout -k 10 300 env MUZERO_AMD_LIB=$PWD/muzero_amd/lib/exp/$n.so python -m pytest tests/test_gpu_net.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3 || exit 1
  done
fi
for pass in 1 2; do
  for n in "$@"; do
    timeout -k 10 120 env MUZERO_AMD_LIB=$PWD/muzero_amd/lib/exp/$n.so python tools/tower_bench.py 4096 hip 2>&1 | grep -v amdgpu.ids | sed "s/^/$n pass $pass: /" || exit 1
  done
done
for n in "$@"; do
  echo "== stamps $n"
  timeout -k 10 120 env MUZERO_AMD_LIB=$PWD/muzero_amd/lib/exp/$n.so python tools/tower_stamps.py 4096 2>&1 | grep -v "amdgpu.ids\|s_memtime\|mid [1-4]" || exit 1
done
