"""per-simulation-step kernel breakdown from a rocprofv3 kernel trace of bench.py:
   python tools/step_breakdown.py <kernel_trace.csv> [out.txt]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_backup_select' in r['Kernel_Name']]
a, b = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
seg = rows[a + 1:b + 1]
tot = {}
for r in seg:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    k = r['Kernel_Name'][:110]
    tot.setdefault(k, [0, 0.0]); tot[k][0] += 1; tot[k][1] += d
span = (int(rows[b]['End_Timestamp']) - int(rows[a]['End_Timestamp'])) / 1e3
out = ['One simulation step (graph replay) of bench.py from the rocprofv3 kernel trace',
       '(kernels between two consecutive k_backup_select launches in the timed region)',
       'span %.1f us, sum of kernel durations %.1f us, %d kernels' % (span, sum(v[1] for v in tot.values()), len(seg)), '']
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    out.append('%8.1f us x%2d  %s' % (v[1], v[0], k))
txt = '\n'.join(out) + '\n'
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], 'w').write(txt)
