#!/bin/bash
# SQ counters of k_backup_select at 4096 trees for one library build: tools/tree_pmc.sh <name>
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r2/treepmc; mkdir -p $O; rm -rf $O/run
export MUZERO_AMD_LIB=$PWD/muzero_amd/lib/exp/$1.so
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/run -o p -- python3 tools/tree_bench.py 4096 50 > $O/out.txt 2> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
python3 tools/pmc_summary.py $(find $O/run -name "p_counter_collection.csv" | head -1) k_backup_select
rm -rf $O/run
timeout -s KILL 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $O/run -o p -- python3 tools/tree_bench.py 4096 50 > $O/out.txt 2> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
python3 tools/pmc_summary.py $(find $O/run -name "p_counter_collection.csv" | head -1) k_backup_select
rm -rf $O/run
