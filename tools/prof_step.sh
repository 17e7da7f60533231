#!/bin/bash
# usage: prof_step.sh <out dir under gpurun_out> [ENV=VAL ...]   kernel stats + idle / bookkeeping breakdown of bench steps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; shift
mkdir -p $O
for e in "$@"; do export "$e"; done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-aux > $O/ks_bench.json 2> $O/ks.err && python profiles/summarize.py $O/ks 18 > $O/kernel_stats.txt 2>&1 && python tools/gpu_idle.py $O/ks k_eb_encode 4 3 > $O/gpu_idle.txt 2>&1
tail -n +1 $O/gpu_idle.txt | head -12
