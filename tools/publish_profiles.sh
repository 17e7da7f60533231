#!/bin/bash
# Copy the summaries of the last tools/measure_round.sh batch (gpurun_out/meas/) into profiles/ under the round's names.
# usage: tools/publish_profiles.sh r03
R=${1:-r04}; M=gpurun_out/meas; P=profiles
cp $M/kernel_stats.txt $P/${R}_kernel_stats_bench_steps5_warm2.txt
cp $M/gpu_idle.txt $P/${R}_gpu_idle_and_bookkeeping.txt
cp $M/kernel_attribution.txt $P/${R}_kernel_attribution.txt
cp $M/layer_report.txt $P/${R}_layer_report.txt
cp $M/pmc_by_kernel.txt $P/${R}_pmc_fetch_write_by_kernel.txt
cp $M/pmc_traffic.json $P/${R}_pmc_traffic.json
cp $M/pmc_traffic.json $P/pmc_traffic.json
tail -n 1 $M/bench.json > $P/${R}_bench_line_steps20_warm5.json
cp $M/train_kernel_stats.txt $P/${R}_train_step_kernel_stats.txt
grep -v amdgpu.ids $M/eval_frames.txt > $P/${R}_eval_frames_blocks_1gpu.txt
[ -f $M/host_timeline.txt ] && cp $M/host_timeline.txt $P/${R}_host_timeline.txt
[ -f $M/host_reads.txt ] && grep -v amdgpu.ids $M/host_reads.txt > $P/${R}_host_reads_per_step.txt
[ -f $M/sq_by_kernel.txt ] && cp $M/sq_by_kernel.txt $P/${R}_sq_mfma_busy_by_kernel.txt
[ -f $M/train_gpu_gaps.txt ] && cp $M/train_gpu_gaps.txt $P/${R}_train_step_gpu_gaps.txt
[ -f $M/train_attribution.txt ] && grep -v amdgpu.ids $M/train_attribution.txt > $P/${R}_train_step_by_layer.txt
[ -f $M/train_host_by_node.txt ] && grep -v "amdgpu.ids\|Warn\|warn" $M/train_host_by_node.txt > $P/${R}_train_step_host_by_node.txt
ls -la $P | grep ${R}_ | wc -l
