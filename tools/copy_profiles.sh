#!/bin/bash
# Copy the judged summaries of one tools/gpu_profile.sh run (gpurun_out/<tag>/) into profiles/r05_*.   usage: tools/copy_profiles.sh [tag=round5]
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/${1:-round5}; P=$R/profiles
set -e
cp $O/bench_f32.json $P/r05_bench_f32.json
cp $O/bench_f32_2ranks_one_device.json $P/r05_bench_f32_2ranks_one_device_emulation.json
cp $O/bench_f32_shard8.json $P/r05_bench_f32_shard_of_8.json
cp $O/bench_f64_4k.json $P/r05_bench_f64_4k.json
cp $O/hbm_traffic.json $P/r05_hbm_traffic.json
cp $O/pmc_summary.json $P/r05_pmc_summary.json
cp $O/kernel_resources.txt $P/r05_kernel_resources.txt
cp $O/pool_stage_profile.txt $P/r05_pool_stage_profile.txt
cp $O/probe_phases.txt $P/r05_probe_phases.txt
cp $O/probe_phases_f64.txt $P/r05_probe_phases_f64.txt
for p in "f32:f32_1080p_1000spp" "f32_cull:f32_1080p_1000spp_group_cull" "f32_pool:f32_1080p_1000spp_ray_pool" "f32_valu:f32_1080p_1000spp_scan_valu" "f64:f64_4k_1000spp"; do
  cp $O/trace_${p%%:*}/*kernel_stats.csv $P/r05_kernel_stats_${p#*:}.csv
done
echo "copied $O -> $P/r05_*"
