"""mean PMC counter values per kernel from a rocprofv3 --pmc counter_collection.csv:
   python tools/pmc_summary.py <counter_collection.csv> <kernel substring>"""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
acc = collections.defaultdict(list)
for r in rows:
    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-32s n=%d mean=%.4g" % (k, len(v), sum(v) / len(v)))
