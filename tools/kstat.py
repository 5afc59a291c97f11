"""python tools/kstat.py <kernel_stats.csv> <substring> ...: calls / avg / min / max ns of the kernels whose name holds a substring"""
import csv, sys
csv.field_size_limit(1 << 30)
rows = list(csv.DictReader(open(sys.argv[1])))
for sub in sys.argv[2:]:
    for r in rows:
        if sub in r["Name"]:
            print("%s: calls %s avg %.0f ns min %s max %s" % (sub, r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"]), end=" | ")
print()
