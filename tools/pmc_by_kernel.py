#!/usr/bin/env python3
"""Per-kernel HBM traffic and achieved bandwidth from the two rocprofv3 PMC passes of tools/pmc_traffic.py
(FETCH_SIZE / WRITE_SIZE, KiB; FETCH doubled per the gfx950 note).  Kernel durations are taken from the same passes
(counter collection serialises kernels, so they are slightly longer than in an untraced run).
usage: python tools/pmc_by_kernel.py gpurun_out/pmc18 > profiles/r01_pmc_fetch_write_by_kernel.txt"""
import collections
import csv
import glob
import os
import re
import sys

out = sys.argv[1]
agg = collections.defaultdict(lambda: {"n": 0, "us": 0.0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(os.path.join(out, ctr, "**", "*counter_collection.csv"), recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != ctr:
            continue
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:60]
        a = agg[name]
        a[ctr] += float(r["Counter_Value"])
        if ctr == "FETCH_SIZE":
            a["n"] += 1
            a["us"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print(f"# {out}: bench.py --steps 2 --warmup 1 --coder symbols (plus the accounting step); per kernel, summed over the run")
print(f"{'kernel':60s} {'launches':>8s} {'ms':>9s} {'read GB':>9s} {'write GB':>9s} {'GB/s':>8s} {'of 8 TB/s':>9s}")
rows = sorted(agg.items(), key=lambda kv: -kv[1]["us"])
for name, a in rows[:32]:
    rd, wr = 2.0 * a["FETCH_SIZE"] * 1024 / 1e9, a["WRITE_SIZE"] * 1024 / 1e9
    ms = a["us"] / 1e3
    bw = (rd + wr) / (ms * 1e-3) if ms > 0 else 0.0
    print(f"{name:60s} {a['n']:8d} {ms:9.3f} {rd:9.2f} {wr:9.2f} {bw:8.0f} {bw / 8000:9.2f}")
