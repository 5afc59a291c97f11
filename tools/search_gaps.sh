#!/bin/bash
# GPU idle time inside the timed searches of bench.py (steady state: the last 100 ms of the trace)
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r2/searchgaps; mkdir -p $O; rm -rf $O/run
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $O/run -o t -- python3 bench.py --steps 20 --warmup 3 --skip-extras > $O/out.txt 2> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
python3 tools/trace_gaps.py $(find $O/run -name "t_kernel_trace.csv" | head -1) 100
rm -rf $O/run
