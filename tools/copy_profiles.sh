#!/bin/bash
# Copy the judged summaries of one tools/gpu_profile.sh run (gpurun_out/<tag>/) into profiles/r06_*.   usage: tools/copy_profiles.sh [tag=round6]
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/${1:-round6}; P=$R/profiles
set -e
cp $O/bench_f32.json $P/r06_bench_f32.json
cp $O/bench_f32_2ranks_one_device.json $P/r06_bench_f32_2ranks_one_device_emulation.json
cp $O/bench_f32_shard8.json $P/r06_bench_f32_shard_of_8.json
cp $O/bench_f64_4k.json $P/r06_bench_f64_4k.json
cp $O/hbm_traffic.json $P/r06_hbm_traffic.json
cp $O/pmc_summary.json $P/r06_pmc_summary.json
cp $O/kernel_resources.txt $P/r06_kernel_resources.txt
cp $O/valu_budget.txt $P/r06_valu_budget.txt
cp $O/valu_budget_200spp.txt $P/r06_valu_budget_200spp.txt
cp $O/small_frames.json $P/r06_small_frames.json
cp $O/trace_small/*kernel_stats.csv $P/r06_kernel_stats_small_frames.csv
for p in "f32:f32_1080p_1000spp" "f32_cull:f32_1080p_1000spp_group_cull" "f32_valu:f32_1080p_1000spp_scan_valu" "f64:f64_4k_1000spp"; do
  cp $O/trace_${p%%:*}/*kernel_stats.csv $P/r06_kernel_stats_${p#*:}.csv
done
echo "copied $O -> $P/r06_*"
