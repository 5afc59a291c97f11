#!/bin/bash
# top kernels of one training configuration: tools/train_top.sh fp32|bf16
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03/trainprof; mkdir -p $O; rm -rf $O/run
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/run -o t -- python3 tools/train_prof.py $1 12 > $O/out.txt 2> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
python3 - $(find $O/run -name "t_kernel_stats.csv" | head -1) <<'PY'
import csv, sys
csv.field_size_limit(1 << 30)
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU time %.1f ms over 12 steps = %.2f ms per step; %d kernel names, %d launches per step" % (tot / 1e6, tot / 12e6, len(rows), sum(int(r["Calls"]) for r in rows) / 12))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print("%5.1f %%  %7.3f ms/step  calls/step %6.1f  avg %8.1f us  %s" % (100 * float(r["TotalDurationNs"]) / tot, float(r["TotalDurationNs"]) / 12e6, int(r["Calls"]) / 12, float(r["AverageNs"]) / 1e3, r["Name"][:110]))
PY
