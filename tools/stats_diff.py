"""steady-state per-step kernel times = (stats of a long run - stats of a short run) / extra steps.
python tools/stats_diff.py short_kernel_stats.csv long_kernel_stats.csv extra_steps [top]"""
import csv, sys
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        d[r["Name"]] = (int(r["Calls"]), int(r["TotalDurationNs"]))
    return d
a, b, n = load(sys.argv[1]), load(sys.argv[2]), int(sys.argv[3])
top = int(sys.argv[4]) if len(sys.argv) > 4 else 25
rows = []
for k, (c, t) in b.items():
    c0, t0 = a.get(k, (0, 0))
    if c - c0 > 0:
        rows.append(((t - t0) / n / 1e3, (c - c0) / n, k))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("steady state: %.1f us of kernels per step, %.0f launches per step" % (tot, sum(r[1] for r in rows)))
for us, calls, k in rows[:top]:
    print("%9.1f us %7.1f calls  %s" % (us, calls, k[:110]))
