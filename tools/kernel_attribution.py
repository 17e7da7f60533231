#!/usr/bin/env python3
"""Launch -> kernel instantiation -> algorithmic GFLOP -> kernel ms -> TFLOP/s for the convolutions of one step.

Joins the call list of tools/layer_report.py (K, channels, rows, pairs of every convolution call, in call order) with
the rocprofv3 kernel trace of the SAME run: every call launches exactly one kernel of the MFMA family
(k_conv_mfma*, k_conv_wave16*), in order, so the last len(calls) such dispatches of the trace are the calls' kernels.

  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && \
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/attr -- python3 tools/layer_report.py 10 gpurun_out/attr_calls.json
  python tools/kernel_attribution.py gpurun_out/attr gpurun_out/attr_calls.json > profiles/r02_kernel_attribution.txt
"""
import csv
import glob
import json
import re
import sys

trace = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
calls = json.load(open(sys.argv[2]))
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
# convolution launches of the family: MODE_CONV instantiations only (the GDN / IGDN modes of the same kernel are not calls of
# the list), and only the calls that take the MFMA path (thin heads with <= 4 output channels run VALU kernels)
fam = [r for r in rows if re.search(r"k_conv_mfma<\d+, \d+, \d+, \d+, 0, (true|false)>|k_conv_mfma_bf<\d+, \d+, \d+, \d+, 0, \d+>|k_conv_in4_bf<|"
                                 r"k_conv_wave16|k_gemm_bf<\d+, \d+, \d+, \d+, 0>|k_gemm_bf2<\d+>|k_gemm_h2<\d+(, \d+)?>|k_pair_h2<\d+>",
                                 r["Kernel_Name"])]
calls = [c for c in calls if c["cout"] > 4 and (c["cin"] in (4, 8, 16) or c["cin"] % 32 == 0)]
assert len(fam) >= len(calls), (len(fam), len(calls))
fam = fam[-len(calls):]
print("# one encode+decode step of bench.py's frame (tools/layer_report.py under rocprofv3 --kernel-trace); K < 0: generative")
print("# transposed convolution (GEMM half; its gather-sum is not an MFMA kernel).  peak: fp32-input MFMA 157.3 TFLOP/s; split")
print("# path (k_conv_mfma_bf / k_gemm_bf2: 6 bf16 MFMA terms per fp32 product) 2500 / 6 = 416.7 TFLOP/s of algorithmic FLOPs; dense")
print("# products in scaled fp16 pairs (k_gemm_h2: 3 MFMA terms) 2500 / 3 = 833.3.  The chip holds ~1.3-1.5 GHz of its 2.4 GHz on")
print("# these kernels (DESIGN.md section 8), so 0.55-0.6 of these peaks is a saturated matrix pipe.")
print(f"{'#':>2s} {'K':>4s} {'cin':>4s} {'cout':>4s} {'n_out':>9s} {'pairs':>10s} {'GFLOP':>8s} {'kernel ms':>9s} {'TFLOP/s':>8s} {'of peak':>7s}  kernel")
tot_f = tot_t = 0.0
by = {}
for i, (c, r) in enumerate(zip(calls, fam)):
    ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    peak = 833.3 if "_h2" in name else (416.7 if "_bf" in name else 157.3)
    tf = c["gflop"] / ms
    assert tf / peak < 1.0, f"call {i} joined to the wrong launch ({name}: {tf:.0f} TFLOP/s): the family regex does not match the build's kernels"
    print(f"{i:2d} {c['K']:4d} {c['cin']:4d} {c['cout']:4d} {c['n_out']:9d} {c['pairs']:10d} {c['gflop']:8.1f} {ms:9.3f} {tf:8.1f} {tf / peak:7.2f}  {name}")
    tot_f += c["gflop"]
    tot_t += ms
    a = by.setdefault(name, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += c["gflop"]
    a[2] += ms
print(f"# total {tot_f:.1f} GFLOP in {tot_t:.3f} ms of MFMA-family kernels = {tot_f / tot_t:.1f} TFLOP/s")
print("# by instantiation: launches, GFLOP, ms, TFLOP/s")
for name, (n, gf, ms) in sorted(by.items(), key=lambda kv: -kv[1][2]):
    print(f"#   {name:48s} {n:3d} {gf:9.1f} {ms:8.3f} {gf / ms:8.1f}")
