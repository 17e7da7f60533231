#!/bin/bash
# Counter passes for the dense-GEMM probe: tools/pmc_gemm.sh OUTDIR N CIN NCOL   (separate --pmc passes; program directly after "--")
out=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$out
i=0
# (WRITE_SIZE and FETCH_SIZE in one pass exceed the counter hardware: rocprofv3 aborts and then hangs -- one derived metric per pass)
for set in "WRITE_SIZE" "FETCH_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_INSTS_VALU"; do
  i=$((i+1))
  echo "pass $i: $set" >> $R/gpurun_out/$out/progress.txt
  timeout -k 5 120 rocprofv3 --pmc $set -d $R/gpurun_out/$out/p$i -o c --output-format csv -- python3 $R/tools/gemm_probe.py "$@" 2 > $R/gpurun_out/$out/p$i.log 2>&1 || echo "pass $i failed"
done
cd $R
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(f"gpurun_out/{out}/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:50]
        if "k_gemm" not in k and "k_conv_mfma" not in k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
    for k, d in agg.items():
        for c, v in d.items():
            print(f"{k:50s} {c:28s} {v / cnt[(k, c)]:18.0f}  per dispatch ({cnt[(k, c)]})")
PY
