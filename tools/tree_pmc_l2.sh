#!/bin/bash
# L2 counters of k_backup_select at 4096 trees (product library): hits / misses / fabric requests per launch
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r2/treepmc; mkdir -p $O; rm -rf $O/run
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $O/run -o p -- python3 tools/tree_bench.py 4096 50 > $O/out.txt 2> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
python3 tools/pmc_summary.py $(find $O/run -name "p_counter_collection.csv" | head -1) k_backup_select
rm -rf $O/run
timeout -s KILL 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum --kernel-trace --output-format csv -d $O/run -o p -- python3 tools/tree_bench.py 4096 50 > $O/out.txt 2> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
python3 tools/pmc_summary.py $(find $O/run -name "p_counter_collection.csv" | head -1) k_backup_select
rm -rf $O/run
