"""idle time between consecutive kernels of a rocprofv3 kernel trace (the last window_ms of the run = steady state):
   python tools/trace_gaps.py <kernel_trace.csv> [window_ms]"""
import csv, sys, collections
csv.field_size_limit(1 << 30)
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))), key=lambda r: r[0])
ms = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
t_end = max(r[1] for r in rows)
w = [r for r in rows if r[0] >= t_end - ms * 1e6]     # the last `ms` milliseconds of the run = steady state
span = w[-1][1] - w[0][0]
busy = sum(e - s for s, e, _ in w)
gaps = collections.defaultdict(lambda: [0, 0])
end = w[0][1]
for s, e, k in w[1:]:
    g = max(0, s - end)
    key = k[:60]
    gaps[key][0] += g
    gaps[key][1] += 1
    end = max(end, e)
idle = sum(v[0] for v in gaps.values())
print("window: %d kernels, span %.1f ms, kernels busy %.1f ms (%.0f %%), idle before kernels %.1f ms; mean gap %.1f us" %
      (len(w), span / 1e6, busy / 1e6, 100.0 * busy / span, idle / 1e6, idle / 1e3 / len(w)))
for k, (g, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:14]:
    print("%8.2f ms idle in front of %6d launches (%.1f us each)  %s" % (g / 1e6, c, g / 1e3 / c, k))
busy_by = collections.defaultdict(lambda: [0, 0])
for s_, e, k in w:
    busy_by[k[:90]][0] += e - s_
    busy_by[k[:90]][1] += 1
print("busy time by kernel:")
for k, (b, c) in sorted(busy_by.items(), key=lambda kv: -kv[1][0])[:16]:
    print("%8.2f ms in %6d launches (%.1f us each)  %s" % (b / 1e6, c, b / 1e3 / c, k))
