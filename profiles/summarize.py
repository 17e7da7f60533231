#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats run (csv) into a per-kernel table.
usage: python profiles/summarize.py <dir with *_kernel_stats.csv> [steps_in_run]"""
import csv
import glob
import sys

d = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = glob.glob(d + "/**/*_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# {f}\n# total kernel time {tot/1e6:.3f} ms over {steps:g} steps = {tot/1e6/steps:.3f} ms/step")
print(f"{'kernel':88s} {'calls':>6s} {'total_ms':>10s} {'ms/step':>9s} {'avg_us':>10s} {'pct':>6s}")
for r in rows[:40]:
    t = float(r["TotalDurationNs"])
    print(f"{r['Name'][:88]:88s} {r['Calls']:>6s} {t/1e6:10.3f} {t/1e6/steps:9.3f} {float(r['AverageNs'])/1e3:10.1f} {100*t/tot:6.2f}")
