#!/bin/bash
# steady-state kernels of one training step: tools/train_steady.sh fp32|bf16   (two profiled runs of 12 and 32 steps, differenced)
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r2/trainprof; mkdir -p $O
for n in 12 32; do
  rm -rf $O/run$n
  timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/run$n -o t -- python3 tools/train_prof.py $1 $n > $O/out.txt 2> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
done
python3 tools/stats_diff.py $(find $O/run12 -name "t_kernel_stats.csv" | head -1) $(find $O/run32 -name "t_kernel_stats.csv" | head -1) 20 30
rm -rf $O/run12 $O/run32
