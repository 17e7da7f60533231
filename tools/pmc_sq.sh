#!/bin/bash
# SQ counter passes for one convolution: tools/pmc_sq.sh OUTDIR CIN COUT LEVEL
# (separate --pmc passes, no tracing domains; program directly after "--")
set -e
out=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS -d $R/gpurun_out/$out/p1 -o c --output-format csv -- python3 $R/tools/one_conv.py "$@" > $R/gpurun_out/$out/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD -d $R/gpurun_out/$out/p2 -o c --output-format csv -- python3 $R/tools/one_conv.py "$@" > $R/gpurun_out/$out/p2.log 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INSTS_SMEM SQ_WAVES -d $R/gpurun_out/$out/p3 -o c --output-format csv -- python3 $R/tools/one_conv.py "$@" > $R/gpurun_out/$out/p3.log 2>&1
cd $R
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for p in ("p1", "p2", "p3"):
    for f in glob.glob(f"gpurun_out/{out}/{p}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            if "k_conv" not in k:
                continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
        for k, d in agg.items():
            print(p, k)
            for c, v in d.items():
                print(f"   {c:32s} {v / cnt[(k, c)]:16.0f}  per dispatch ({cnt[(k, c)]} dispatches)")
PY
