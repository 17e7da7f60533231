#!/bin/bash
# Counter passes over one short bench run, printed per dispatch for the kernels matching a regex:
#   tools/pmc_kernel.sh OUTDIR 'k_convt_gather_csr' "SET1 counters" "SET2 counters" ...
# (separate --pmc passes, --kernel-trace only; program directly after "--"; one derived metric such as FETCH_SIZE per pass)
out=$1; pat=$2; shift 2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$out
cd $R
i=0
for set in "$@"; do
  i=$((i+1))
  echo "pass $i: $set" >> gpurun_out/$out/progress.txt
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/$out/p$i -o c --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-aux --coder symbols > gpurun_out/$out/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - "$out" "$pat" <<'PY'
import csv, glob, sys, collections, re
out, pat = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(f"gpurun_out/{out}/p*/**/*counter_collection.csv", recursive=True)):
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if not re.search(pat, r["Kernel_Name"]):
            continue
        d = disp.setdefault(r["Dispatch_Id"], {"grid": r.get("Grid_Size", "?"), "ms": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, "name": r["Kernel_Name"][:40]})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for k, d in list(disp.items())[-4:]:
        print("  ".join(f"{a}={b:.4g}" if isinstance(b, float) else f"{a}={b}" for a, b in d.items()))
    print()
PY
