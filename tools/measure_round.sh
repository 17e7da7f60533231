#!/bin/bash
# The round's measurement batch on the GPU box (run from the repo root): kernel stats, launch attribution, layer report,
# PMC traffic, SQ counters, host reads / timeline, the train step, the frame sweep, the bench line.
# Outputs under gpurun_out/meas/; tools/publish_profiles.sh copies what is to be judged into profiles/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/meas; rm -rf $O; mkdir -p $O
echo "kernel stats" > $O/progress.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-aux > $O/ks_bench.json 2> $O/ks.err && python profiles/summarize.py $O/ks 18 > $O/kernel_stats.txt 2>&1 && python tools/gpu_idle.py $O/ks k_eb_encode 4 3 > $O/gpu_idle.txt 2>&1
echo "attribution" >> $O/progress.txt
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/attr -- python3 tools/layer_report.py 10 $O/attr_calls.json > $O/layer_report_profiled.txt 2>&1 && python tools/kernel_attribution.py $O/attr $O/attr_calls.json > $O/kernel_attribution.txt 2>&1
echo "layer report" >> $O/progress.txt
timeout -k 10 200 python tools/layer_report.py 10 > $O/layer_report.txt 2>&1
echo "pmc" >> $O/progress.txt
timeout -k 10 400 python tools/pmc_traffic.py collect $O/pmc > $O/pmc.log 2>&1 && python tools/pmc_traffic.py parse $O/pmc $O/pmc_traffic.json >> $O/pmc.log 2>&1 && python tools/pmc_by_kernel.py $O/pmc > $O/pmc_by_kernel.txt 2>&1
cp $O/pmc_traffic.json profiles/pmc_traffic.json   # (so that the bench line below carries the traffic of THIS code: bench.py reads the stamped file)
echo "sq" >> $O/progress.txt
timeout -k 10 400 python tools/sq_by_kernel.py collect $O/sq > $O/sq.log 2>&1 && python tools/sq_by_kernel.py report $O/sq > $O/sq_by_kernel.txt 2>&1
echo "host" >> $O/progress.txt
timeout -k 10 200 python tools/sync_sites.py > $O/host_reads.txt 2>&1
timeout -k 10 200 python tools/host_timeline.py $O/host_timeline.txt > $O/host_timeline.log 2>&1
echo "train step" >> $O/progress.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -- python3 tools/train_profile.py 10 > $O/train_profile.txt 2>&1 && python profiles/summarize.py $O/train 48 > $O/train_kernel_stats.txt 2>&1 && sed -i '2a # (tools/train_profile.py 10 = bench.train_step_ms: 6 + 10 steps each of the full step, the step without the bottleneck optimiser half, and the full step with fused Adam -- averaged together)' $O/train_kernel_stats.txt
timeout -k 10 200 python3 tools/step_timeline.py $O/train/*/*_kernel_trace.csv 25 2 "k_wgrad_thin<4>" gaps > $O/train_gaps_all.txt 2>&1; (tail -n 1 $O/train_gaps_all.txt; grep idle $O/train_gaps_all.txt | sort -k3 -n -r | head -20) > $O/train_gpu_gaps.txt
timeout -k 10 300 python3 tools/train_attribution.py $O/train_attribution.txt > $O/train_attribution.log 2>&1
timeout -k 10 300 python3 tools/train_backward_cpu.py > $O/train_host_by_node.txt 2>&1
echo "eval frames" >> $O/progress.txt
timeout -k 10 300 python tools/eval_frames.py --bits 9 10 11 9 > $O/eval_frames.txt 2>&1
echo "bench" >> $O/progress.txt
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "done" >> $O/progress.txt
tail -c 600 $O/bench.json
