#!/bin/bash
# prof_quick.sh <outdir> [env assignments...]: rocprofv3 kernel trace of a short bench run -> one step's timeline + per-kernel totals.
O=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $O
for kv in "$@"; do export "$kv"; done
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-aux > $O/bench.json 2> $O/err.txt
T=$(ls $O/ks/*/*_kernel_trace.csv | head -1)
python3 tools/step_timeline.py $T 25 > $O/timeline.txt 2>&1
python3 profiles/summarize.py $O/ks 15 > $O/kernel_stats.txt 2>&1
tail -3 $O/timeline.txt
