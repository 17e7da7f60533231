#!/usr/bin/env python3
"""HBM traffic of the MFMA convolution kernels from rocprofv3 PMC passes (MI355X_MICROARCH.md 'HBM' section):
FETCH_SIZE and WRITE_SIZE need separate passes (TCC slots); both count KiB; on gfx950 FETCH_SIZE reports exactly half
of the bytes of a wide coalesced streaming read, so the read side is doubled (an upper-bound correction for the
gather-heavy kernels, whose 16-byte row pieces are served at sector granularity).

usage (on the GPU box, from the repo root):
  python tools/pmc_traffic.py collect gpurun_out/pmc        # runs two rocprofv3 passes of bench.py
  python tools/pmc_traffic.py parse   gpurun_out/pmc profiles/pmc_traffic.json
The record carries the hash of the kernel sources it was measured on (`bench.csrc_sha`); bench.py reports the traffic
only while that hash matches the code it runs.
"""
import csv
import glob
import json
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_sha  # noqa: E402

KERNELS = ("k_conv_mfma", "k_conv_wave16", "k_gemm_bf2", "k_gemm_h2", "k_pair_h2", "k_convt_gather_csr")
BENCH = ["python3", "bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-aux", "--coder", "symbols"]


def collect(out):
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(out, ctr)
        os.makedirs(d, exist_ok=True)
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "--"] + BENCH
        print(" ".join(cmd), flush=True)
        with open(os.path.join(d, "bench.log"), "w") as f:
            subprocess.run(cmd, stdout=f, stderr=subprocess.STDOUT, check=True, env=dict(os.environ, TMPDIR="/tmp"))


def parse(out, dst):
    res = {}
    per = {}                                  # base kernel name -> {counter: [KiB total, launches]}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        files = glob.glob(os.path.join(out, ctr, "**", "*counter_collection.csv"), recursive=True)
        assert files, f"no counter_collection csv under {out}/{ctr}"
        tot, launches = 0.0, 0
        for r in csv.DictReader(open(files[0])):
            name = r["Kernel_Name"]
            if r["Counter_Name"] == ctr:
                base = re.sub(r"^void ", "", name).split("(")[0].split("<")[0]
                e = per.setdefault(base, {}).setdefault(ctr, [0.0, 0])
                e[0] += float(r["Counter_Value"])
                e[1] += 1
            conv = ("k_conv_wave16" in name or re.search(r"k_conv_mfma<\d+, \d+, \d+, \d+, 0, (true|false)>", name)
                    or re.search(r"k_conv_mfma_bf<\d+, \d+, \d+, \d+, 0, \d+>", name) or "k_conv_in4_bf<" in name or re.search(r"k_gemm_(bf2|h2)<\d+(, \d+)?>|k_pair_h2<\d+>", name))                                   # MODE_CONV only
            if r["Counter_Name"] == ctr and conv:
                tot += float(r["Counter_Value"])
                launches += 1
        res[ctr] = {"kib_total": tot, "launches": launches}
    n = res["FETCH_SIZE"]["launches"]
    assert n == res["WRITE_SIZE"]["launches"] and n > 0
    read = 2.0 * res["FETCH_SIZE"]["kib_total"] * 1024 / n      # gfx950: FETCH_SIZE reads half of wide streaming reads
    write = res["WRITE_SIZE"]["kib_total"] * 1024 / n
    rec = {"kernel": "k_gemm_h2 / k_pair_h2 / k_gemm_bf2 + k_conv_mfma_bf<*,MODE_CONV> + k_conv_mfma<*,MODE_CONV> + k_conv_wave16* (the event-timed MFMA launches)", "launches": n,
           "hbm_read_bytes_per_launch": read, "hbm_write_bytes_per_launch": write,
           "hbm_bytes_per_launch": read + write,
           # the same, kernel by kernel (bench.py quotes the dominant kernel's entry): bytes per launch
           "by_kernel": {k: {"launches": v["WRITE_SIZE"][1],
                             "read": 2.0 * v["FETCH_SIZE"][0] * 1024 / max(v["FETCH_SIZE"][1], 1),
                             "write": v["WRITE_SIZE"][0] * 1024 / max(v["WRITE_SIZE"][1], 1)}
                         for k, v in per.items() if "FETCH_SIZE" in v and "WRITE_SIZE" in v and k.startswith("k_")},
           "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes), KiB*1024, FETCH x2 (gfx950)",
           "command": " ".join(BENCH), "csrc_sha": csrc_sha()}
    json.dump(rec, open(dst, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    if sys.argv[1] == "collect":
        collect(sys.argv[2])
    else:
        parse(sys.argv[2], sys.argv[3])
