#!/bin/bash
# phase stamps of every muzero_amd/lib/exp/<name>.so given (no parity check: diagnostic builds compute garbage on purpose)
mkdir -p gpurun_out/r2
for n in "$@"; do
  echo "== $n"
  timeout -k 10 120 env MUZERO_AMD_LIB=$PWD/muzero_amd/lib/exp/$n.so python tools/tower_stamps.py 4096 2>&1 | grep -v "amdgpu.ids\|K loop\|s_memtime" || exit 1
done
