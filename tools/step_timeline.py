#!/usr/bin/env python3
"""One step's launches from a rocprofv3 kernel trace, in time order: start, duration, gap before it, kernel.
step_timeline.py <kernel_trace.csv> [min_us=25] [step_from_end=2] [marker=k_frame_intake] [gaps]
marker: a kernel that runs exactly once per step (the window is marker .. next marker); "gaps": list only the idle periods
>= 15 us with the kernel before and after each."""
import csv
import sys

f = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 25.0
back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marker = sys.argv[4] if len(sys.argv) > 4 else 'k_frame_intake'
idx = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
a, b = idx[-back - 1], idx[-back]
t0 = int(rows[a]['Start_Timestamp'])
prev_end = t0
out = []
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    out.append(((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r['Kernel_Name'][:90]))
    prev_end = max(prev_end, e)
if len(sys.argv) > 5 and sys.argv[5] == "gaps":
    for i, o in enumerate(out):
        if o[2] >= 15:
            print("%9.1f idle %7.1f us   after %-50s before %s" % (o[0] - o[2], o[2], out[i - 1][3][:50] if i else "-", o[3][:50]))
else:
    for o in out:
        if o[1] >= min_us or o[2] >= 15:
            print("%9.1f dur %8.1f gap %6.1f  %s" % o)
small = [o for o in out if o[1] < min_us]
print("window %.1f us;" % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3), "launches", len(out), "kernel time %.1f us" % sum(o[1] for o in out), "gaps %.1f us" % sum(max(o[2], 0) for o in out),
      "| launches under %g us: %d, %.1f us" % (min_us, len(small), sum(o[1] for o in small)))
