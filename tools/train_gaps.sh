#!/bin/bash
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03/trainprof; mkdir -p $O; rm -rf $O/run
timeout -s KILL 400 rocprofv3 --kernel-trace --output-format csv -d $O/run -o t -- python3 tools/train_prof.py $1 40 > $O/out.txt 2> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
python3 tools/trace_gaps.py $(find $O/run -name "t_kernel_trace.csv" | head -1)
rm -rf $O/run
