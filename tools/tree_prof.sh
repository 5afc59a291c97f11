#!/bin/bash
# GPU-side durations of k_backup_select (rocprofv3 kernel trace) for library builds and tree counts: tools/tree_prof.sh "<names>" "<counts>"
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r2/treeprof; mkdir -p $O
for n in $2; do for v in $1; do
  export MUZERO_AMD_LIB=$PWD/muzero_amd/lib/exp/$v.so
  rm -rf $O/run
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/run -o k -- python3 tools/tree_bench.py $n 50 > $O/out.txt 2> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
  f=$(find $O/run -name "k_kernel_stats.csv" | head -1)
  echo "$v n=$n: $(python3 tools/kstat.py $f k_backup_select k_tower)"
done; done
