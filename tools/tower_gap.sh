#!/bin/bash
# launch begin/end overhead of k_tower: rocprofv3's duration of the stamped launch against the in-kernel span (first tile start -> last tile end)
# tools/tower_gap.sh <names...>
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r2/gap; mkdir -p $O
for v in "$@"; do
  export MUZERO_AMD_LIB=$PWD/muzero_amd/lib/exp/$v.so
  rm -rf $O/run
  timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $O/run -o g -- python3 tools/tower_stamps.py 4096 > $O/out_$v.txt 2> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
  f=$(find $O/run -name "g_kernel_trace.csv" | head -1)
  python3 - "$f" "$v" "$O/out_$v.txt" <<'PY'
import csv, sys, re
csv.field_size_limit(1 << 30)
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_tower" in r["Kernel_Name"]]
last = rows[-1]
dur = (int(last["End_Timestamp"]) - int(last["Start_Timestamp"])) / 1e3
span = float(re.search(r"kernel span first start -> last end: ([0-9.]+) us", open(sys.argv[3]).read()).group(1))
med = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[-200:])[100]
print("%s: stamped launch %.1f us by rocprofv3, in-kernel span %.1f us, begin/end overhead %.1f us (median of the last 200 launches %.1f us)" % (sys.argv[2], dur, span, dur - span, med))
PY
done
