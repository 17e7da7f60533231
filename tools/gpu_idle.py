#!/usr/bin/env python3
"""GPU idle time and bookkeeping share of one step, from a rocprofv3 --kernel-trace csv of `bench.py`.

usage: python tools/gpu_idle.py <dir with *_kernel_trace.csv> [marker kernel] [steps to analyse] [first step]

The trace is cut into steps at every launch of the marker kernel (default: `k_coords_bounds`, the first kernel of
`UnifiedModel.compress`); the last `steps` complete steps are analysed.  Reports per step: wall span, kernel-busy time,
idle time (gaps between consecutive kernels), launches, copyBuffer / fillBuffer launches, and the time in kernels grouped
as matrix (MFMA) / gather-sum / heads / entropy coder / bookkeeping (coordinate sets, maps, scans, select, prune).
"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "k_coords_bounds"
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
first = int(sys.argv[4]) if len(sys.argv) > 4 else None     # index of the first analysed step (default: the last ones)
f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
cuts = [i for i, r in enumerate(rows) if marker in r[2]]
if len(cuts) < nsteps + 1:
    sys.exit(f"marker {marker} seen {len(cuts)} times: not enough steps")
cuts = cuts[-(nsteps + 1):] if first is None else cuts[first:first + nsteps + 1]

GROUPS = (
    ("matrix", ("k_gemm_h2", "k_pair_h2", "k_conv_mfma", "k_gemm_bf2", "k_conv_wave16", "k_hyper", "k_gemm_ws")),
    ("gather-sum", ("k_convt_gather", "k_pair_reduce", "k_csr_reduce", "k_splitk_reduce", "k_presence")),
    ("heads 2nd conv", ("k_thin", "k_head2")),
    ("entropy coder", ("k_rans", "k_gauss", "k_eb_")),
    ("splits/gdn", ("k_feat_split", "k_gdn")),
    ("copy/fill", ("__amd_rocclr", "at::native")),
)


def group(name):
    for g, pre in GROUPS:
        if any(p in name for p in pre):
            return g
    return "bookkeeping"


tot = collections.Counter()
per_kernel = collections.Counter()
per_kernel_n = collections.Counter()
gaps = []
for a, b in zip(cuts[:-1], cuts[1:]):
    seg = rows[a:b]
    span = seg[-1][1] - seg[0][0]
    busy = 0
    end = seg[0][0]
    for s, e, nme in seg:
        if s > end:
            gaps.append((s - end, prev))
        busy += max(0, e - max(s, end))
        end = max(end, e)
        prev = nme
        tot[group(nme)] += e - s
        per_kernel[nme.split("(")[0][:60]] += e - s
        per_kernel_n[nme.split("(")[0][:60]] += 1
    tot["_span"] += span
    tot["_busy"] += busy
    tot["_launches"] += len(seg)
    tot["_copy"] += sum(1 for r in seg if "copyBuffer" in r[2])
    tot["_fill"] += sum(1 for r in seg if "fillBuffer" in r[2])
n = float(nsteps)
print(f"# {f}: last {nsteps} steps cut at {marker}")
print(f"span {tot['_span'] / n / 1e6:.3f} ms/step  busy {tot['_busy'] / n / 1e6:.3f}  idle {(tot['_span'] - tot['_busy']) / n / 1e6:.3f}  "
      f"launches {tot['_launches'] / n:.0f}  copyBuffer {tot['_copy'] / n:.0f}  fillBuffer {tot['_fill'] / n:.0f}")
for g in [g for g, _ in GROUPS] + ["bookkeeping"]:
    print(f"  {g:16s} {tot[g] / n / 1e6:7.3f} ms/step")
big = sorted(gaps, reverse=True)
print(f"gaps > 20 us: {sum(1 for g in gaps if g[0] > 20000) / n:.1f} per step, "
      f"{sum(g[0] for g in gaps if g[0] > 20000) / n / 1e6:.3f} ms/step; > 5 us: {sum(1 for g in gaps if g[0] > 5000) / n:.1f} per step, "
      f"{sum(g[0] for g in gaps if g[0] > 5000) / n / 1e6:.3f} ms/step; all gaps {sum(g[0] for g in gaps) / n / 1e6:.3f}")
after = collections.Counter()
for g, nme in gaps:
    if g > 20000:
        after[nme.split("(")[0][:50]] += g
print("idle after (gaps > 20 us, ms/step):")
for k, v in after.most_common(14):
    print(f"  {v / n / 1e6:7.3f}  {k}")
print("bookkeeping kernels (ms/step, launches/step):")
for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1]):
    if group(k) == "bookkeeping":
        print(f"  {v / n / 1e6:7.3f} {per_kernel_n[k] / n:6.1f}  {k}")
