#!/bin/bash
# rocprofv3 evidence of one round, on the GPU box:  bash tools/profile_round.sh r02
# kernel-trace stats of the bench's timed leg, HBM traffic by two separate PMC passes, SQ counters and in-kernel phase stamps of
# the inference kernel, then the un-profiled bench line.  Every pass is its own run with a hard limit; the program sits directly
# behind `--`.  Summaries land in gpurun_out/<round>/profile/ (copy what is to be judged into profiles/<round>/).
set -u
R=${1:-r02}
O=gpurun_out/$R/profile
mkdir -p $O
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
step() { echo "== $*"; }
step kernel stats
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 bench.py --steps 5 --warmup 2 --skip-extras > $O/bench_under_rocprof.json 2> $O/stats.err || exit 1
cp $(find $O/stats -name "b_kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv || exit 1
B="python3 bench.py --steps 1 --warmup 1 --selfplay-games 0 --cpu-trees 8 --fp32-steps 0 --train-batch 0 --config4-searches 0"
step pmc fetch
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- $B > /dev/null 2> $O/pmc_f.err || exit 1
step pmc write
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- $B > /dev/null 2> $O/pmc_w.err || exit 1
python3 tools/pmc_hbm.py $(find $O/pmc_f -name "f_counter_collection.csv" | head -1) $(find $O/pmc_w -name "w_counter_collection.csv" | head -1) $O/pmc_summary.json \
  k_tower k_backup_select k_env_step@1048576=k_env_step k_env_step@16777216=k_env_step_hbm k_env_random_playout > /dev/null || exit 1
step tower sq counters
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $O/sq -o t -- python3 tools/tower_bench.py 4096 hip > $O/tower_bench_under_pmc.txt 2> $O/sq.err || exit 1
python3 tools/pmc_summary.py $(find $O/sq -name "t_counter_collection.csv" | head -1) k_tower > $O/tower_pmc_sq.txt || exit 1
step stamps
timeout -k 10 200 python3 tools/tower_stamps.py 4096 2>&1 | grep -v amdgpu.ids > $O/tower_phase_stamps.txt || exit 1
timeout -k 10 200 python3 tools/tower_bench.py 4096 2>&1 | grep -v amdgpu.ids > $O/tower_vs_miopen.txt || exit 1
step split-precision tower kernel
X="python3 tools/x3_engine_prof.py 4096 20"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/x3s -o x -- $X > $O/x3_engine_under_rocprof.txt 2> $O/x3s.err || exit 1
cp $(find $O/x3s -name "x_kernel_stats.csv" | head -1) $O/x3_kernel_stats.csv || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/x3f -o f -- $X > /dev/null 2> $O/x3f.err || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/x3w -o w -- $X > /dev/null 2> $O/x3w.err || exit 1
python3 tools/pmc_hbm.py $(find $O/x3f -name "f_counter_collection.csv" | head -1) $(find $O/x3w -name "w_counter_collection.csv" | head -1) $O/x3_pmc_summary.json "ResArgs)=k_restower_x3" > /dev/null || exit 1
timeout -k 10 100 python3 tools/restower_stamps.py 4096 2>&1 | grep -v amdgpu.ids > $O/restower_x3_phase_stamps.txt || exit 1
timeout -k 10 100 python3 tools/restower_bench.py 4096 2>&1 | grep -v amdgpu.ids > $O/restower_x3_vs_chain_vs_miopen.txt || exit 1
rm -rf $O/x3s $O/x3f $O/x3w
# the bench line reads its `traffic` fields from profiles/<round>/: give it THIS call's PMC summaries
mkdir -p profiles/$R && cp $O/pmc_summary.json $O/x3_pmc_summary.json profiles/$R/ || exit 1
step bench
step training step kernel stats
bash tools/train_gaps.sh hip > $O/train_hip_step_kernels.txt 2>&1 || exit 1   # steady-state window of the kernel trace: busy time per kernel
timeout -k 10 120 python3 tools/convgrad_bench.py 2>&1 | grep -v amdgpu.ids > $O/train_kernels_vs_miopen.txt || exit 1
timeout -k 10 600 python3 bench.py > $O/bench_final.json 2> $O/bench_final.err || exit 1
rm -rf $O/stats $O/pmc_f $O/pmc_w $O/sq
echo done
