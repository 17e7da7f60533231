#!/usr/bin/env python3
"""Per-dispatch HBM read / write bytes and duration of the kernels matching a pattern, from the two passes of
tools/pmc_traffic.py collect:  python tools/pmc_dispatches.py gpurun_out/pmc k_convt_gather_csr [last_n]"""
import csv
import glob
import os
import re
import sys

out, pat = sys.argv[1], sys.argv[2]
last = int(sys.argv[3]) if len(sys.argv) > 3 else 8
cols = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(os.path.join(out, ctr, "**", "*counter_collection.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == ctr and re.search(pat, r["Kernel_Name"])]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    cols[ctr] = rows
n = min(len(cols["FETCH_SIZE"]), len(cols["WRITE_SIZE"]))
print(f"# {pat}: {n} dispatches; FETCH_SIZE x2 (gfx950), KiB*1024")
for fr, wr in list(zip(cols["FETCH_SIZE"][:n], cols["WRITE_SIZE"][:n]))[-last:]:
    ms = (int(fr["End_Timestamp"]) - int(fr["Start_Timestamp"])) / 1e6
    rd, wrb = 2 * float(fr["Counter_Value"]) * 1024 / 1e9, float(wr["Counter_Value"]) * 1024 / 1e9
    print(f"grid {fr.get('Grid_Size', fr.get('Grid_Size_X', '?')):>10s}  {ms:7.3f} ms  read {rd:7.3f} GB  write {wrb:7.3f} GB  -> {(rd + wrb) / ms:6.2f} TB/s")
