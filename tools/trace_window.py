#!/usr/bin/env python3
"""Print one step's kernel timeline from a rocprofv3 kernel trace: usage trace_window.py <dir> [min_dur_us] [min_gap_us]"""
import csv, glob, sys
d = sys.argv[1]; md = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0; mg = float(sys.argv[3]) if len(sys.argv) > 3 else 15.0
f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
cuts = [i for i, n in enumerate(names) if n.startswith("k_frame_intake(") or n.startswith("k_frame_intake")]
cuts = [c for c in cuts if "reduce" not in names[c]]
# the last complete step that contains rANS kernels
best = None
for a, b in zip(cuts[:-1], cuts[1:]):
    if any("k_rans_decode" in n for n in names[a:b]) and int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"]) < 40e6:
        best = (a, b)
a, b = best
t0 = int(rows[a]["Start_Timestamp"]); prev = None
print(f"step of {b - a} launches, span {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    if (e - s) / 1e3 >= md or gap >= mg:
        print(f"{(s - t0) / 1e3:9.1f} dur {(e - s) / 1e3:8.1f} gap {gap:7.1f}  {r['Kernel_Name'].split('(')[0][:56]}")
    prev = max(e, prev or 0)
