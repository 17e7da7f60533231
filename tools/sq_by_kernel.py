#!/usr/bin/env python3
"""MFMA utilisation, effective clock and LDS bank conflicts of the step's matrix / gather kernels from rocprofv3 SQ passes over a
short bench run (`profiles/r04_sq_mfma_busy_by_kernel.txt`).

  python tools/sq_by_kernel.py collect gpurun_out/sq      # separate --pmc passes (with --kernel-trace only), program after "--"
  python tools/sq_by_kernel.py report  gpurun_out/sq > profiles/r04_sq_mfma_busy_by_kernel.txt

Derived per kernel instantiation (averages per dispatch over the run's launches of the timed steps):
  cycles       = GRBM_GUI_ACTIVE / 8                (the counter sums the 8 XCDs: MI355X_MICROARCH.md 'DVFS give-back')
  clock        = cycles / duration
  MFMA busy    = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles)   share of the chip's SIMD-cycles with the matrix pipe occupied
                 (the counter adds, per SIMD, the pipe cycles of every MFMA issued: 32 for a 32x32x16 16-bit MFMA; checked against
                 SQ_INSTS_MFMA)
  x clk/2.4    = MFMA busy x clock / 2.4 GHz        the same share of what the pipe could do at the peak clock
  waves/SIMD   = 4 x SQ_WAVE_CYCLES / (1024 x cycles)               (SQ_WAVE_CYCLES counts in quad-cycle units)
  LDS conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
"""
import collections
import csv
import glob
import os
import re
import subprocess
import sys

PASSES = (("p1", "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA"),
          ("p2", "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU"))
BENCH = ["python3", "bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-aux", "--coder", "symbols"]
KERNELS = r"k_gemm_h2<|k_pair_h2<|k_conv_mfma_bf<|k_convt_gather_csr<|k_gemm_bf2<|k_gdn_bf<|k_conv_in4_bf<|k_thin_project"


def collect(out):
    for name, ctrs in PASSES:
        d = os.path.join(out, name)
        os.makedirs(d, exist_ok=True)
        cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + ctrs.split() + ["--output-format", "csv", "-d", d, "--"] + BENCH
        print(" ".join(cmd), flush=True)
        with open(os.path.join(d, "bench.log"), "w") as f:
            subprocess.run(cmd, stdout=f, stderr=subprocess.STDOUT, check=True, env=dict(os.environ, TMPDIR="/tmp"))


def report(out):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    dur = collections.defaultdict(float)
    for name, _ in PASSES:
        files = glob.glob(os.path.join(out, name, "**", "*counter_collection.csv"), recursive=True)
        assert files, f"no counter csv under {out}/{name}"
        seen = set()
        for r in csv.DictReader(open(files[0])):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
            if not re.search(KERNELS, k):
                continue
            # one instantiation may run very different shapes: split by grid size
            key = (k, r.get("Grid_Size", "?"))
            agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
            if name == "p2" and (key, r["Dispatch_Id"]) not in seen:
                seen.add((key, r["Dispatch_Id"]))
                n[key] += 1
                dur[key] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("# rocprofv3 --kernel-trace --pmc passes of `" + " ".join(BENCH) + "` (counter collection serialises kernels: durations are a")
    print("# little longer than in an untraced run).  One row per (kernel instantiation, grid size); averages per dispatch.")
    print(f"{'kernel':44s} {'grid':>10s} {'n':>3s} {'us':>8s} {'clock GHz':>9s} {'MFMA busy':>9s} {'x clk/2.4':>9s} {'waves/SIMD':>10s} {'MFMA insts':>10s} {'LDS confl':>9s}")
    rows = sorted(agg.items(), key=lambda kv: -dur[kv[0]])
    for key, c in rows:
        if n[key] == 0:
            continue
        us = dur[key] / n[key]
        if us < 20:
            continue
        cycles = c.get("GRBM_GUI_ACTIVE", 0.0) / n[key] / 8          # per dispatch
        clock = cycles / (us * 1e-6) / 1e9 if us > 0 else 0.0
        # p1 counters are sums over the same number of dispatches (same run shape): per-dispatch averages via n[key]
        mfma = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / n[key] / (1024.0 * cycles) if cycles else 0.0
        waves = 4.0 * c.get("SQ_WAVE_CYCLES", 0.0) / n[key] / (1024.0 * cycles) if cycles else 0.0
        lds = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"] if c.get("SQ_LDS_IDX_ACTIVE") else 0.0
        print(f"{key[0][:44]:44s} {key[1]:>10s} {n[key]:3d} {us:8.1f} {clock:9.2f} {mfma:9.2f} {mfma * clock / 2.4:9.2f} {waves:10.2f} "
              f"{c.get('SQ_INSTS_MFMA', 0.0) / max(n[key], 1):10.3g} {lds:9.3f}")


if __name__ == "__main__":
    (collect if sys.argv[1] == "collect" else report)(sys.argv[2])
