#!/bin/bash
# The round's whole measurement set in ONE call on the GPU box (same box, same binary): what the driver runs at round end
# (smoke, pytest -m gpu, bench) plus the rocprofv3 summaries that are committed under profiles/.
#   usage: tools/gpu_profile.sh [tag=round5] [notests]
# Every rocprofv3 pass is its own run: --kernel-trace --stats only, or --pmc only with --kernel-trace (FETCH_SIZE and WRITE_SIZE
# do not fit one pass; MI355X_MICROARCH.md, HBM section).  Output: gpurun_out/<tag>/ -- kernel_stats CSVs, pmc_summary.json,
# hbm_traffic.json (FETCH_SIZE x 2 + WRITE_SIZE per launch, KiB -> bytes), bench_*.json.  Copy what is to be judged into profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; TAG=${1:-round5}; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
[ "$2" = "notests" ] || timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
B32="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
B64="python $R/bench.py --dtype f64 --width 3840 --steps 1 --warmup 0 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_f32 -o t -- $B32 > $O/trace_f32.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_f32_cull -o t -- $B32 --group-cull > $O/trace_f32_cull.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_f32_valu -o t -- $B32 --scan-valu > $O/trace_f32_valu.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_f32_pool -o t -- $B32 --ray-pool > $O/trace_f32_pool.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_f64 -o t -- $B64 > $O/trace_f64.log 2>&1
for f in $O/trace_*/*kernel_stats.csv; do echo "== $f"; cat $f; done
pmc() { tag=$1; shift; cmd=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$tag -o p -- $cmd > $O/pmc_$tag.log 2>&1; }
P32="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras"
pmc f32_fetch "$P32" GRBM_GUI_ACTIVE FETCH_SIZE
pmc f32_write "$P32" GRBM_GUI_ACTIVE WRITE_SIZE
pmc f32_cull_fetch "$P32 --group-cull" GRBM_GUI_ACTIVE FETCH_SIZE
pmc f32_cull_write "$P32 --group-cull" GRBM_GUI_ACTIVE WRITE_SIZE
pmc f64_fetch "$B64" GRBM_GUI_ACTIVE FETCH_SIZE
pmc f64_write "$B64" GRBM_GUI_ACTIVE WRITE_SIZE
pmc f32_sqA "$P32" GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU
pmc f32_sqB "$P32" SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_FLAT
pmc f32_mfma "$P32" SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16
pmc f32_cull_sqA "$P32 --group-cull" GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU
pmc f32_cull_mfma "$P32 --group-cull" SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM
pmc f32_valu_sqA "$P32 --scan-valu" GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU
pmc f32_pool_sqA "$P32 --ray-pool" GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU
pmc f32_pool_sqB "$P32 --ray-pool" SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_FLAT
pmc f32_pool_mfma "$P32 --ray-pool" SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pmc f64_sqA "$B64" GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU
pmc f64_mfma "$B64" SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16
python3 $R/tools/pmc_summarise.py $O
# the bench lines read the HBM traffic of THIS run (box-local copy; the merged gpurun_out/<tag>/hbm_traffic.json is what gets
# committed as profiles/r05_hbm_traffic.json (likewise r05_pmc_summary.json: `roofline.issue_busy`) -- `traffic_static` in the line says that the figure is not measured by bench.py itself)
cp $O/hbm_traffic.json $R/profiles/r05_hbm_traffic.json; cp $O/pmc_summary.json $R/profiles/r05_pmc_summary.json
cd $R
python bench.py > $O/bench_f32.json 2> $O/bench_f32.err; cut -c1-400 $O/bench_f32.json
python bench.py --dtype f64 --width 3840 --steps 2 --warmup 1 --no-extras > $O/bench_f64_4k.json 2> $O/bench_f64_4k.err; cut -c1-300 $O/bench_f64_4k.json
python bench.py --emulate-shard-of 8 --steps 3 --no-cpu-baseline > $O/bench_f32_shard8.json 2>/dev/null; cut -c1-300 $O/bench_f32_shard8.json
RTW_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_f32_2ranks_one_device.json 2>/dev/null; cut -c1-300 $O/bench_f32_2ranks_one_device.json
python tools/kernel_resources.py > $O/kernel_resources.txt 2>&1; tail -12 $O/kernel_resources.txt
# calibration of WRITE_SIZE / HW_REG_XCC_ID and the price of each phase (built on the box)
mkdir -p $R/build
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 tools/ubench_write_size.hip -o build/ubench_write_size 2>/dev/null
(cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/ws -o w -- $R/build/ubench_write_size > $O/ws.log 2>&1)
python3 - > $O/ubench_write_size.txt <<PY
import csv, glob
print(open("$O/ws.log").read().split("tool finalization")[-1].split("\n", 1)[-1] if "tool finalization" in open("$O/ws.log").read() else open("$O/ws.log").read())
for f in glob.glob("$O/ws/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "WRITE_SIZE" and ("k_" in r["Kernel_Name"]): print("%-24s WRITE_SIZE %10.1f KiB = %.2f x the 24.9 MB written" % (r["Kernel_Name"][:24], float(r["Counter_Value"]), float(r["Counter_Value"]) / 24300))
PY
cat $O/ubench_write_size.txt
bash tools/gpu_probe_phases.sh base mfma mfma2 nomfma eval extract resolve reject noskip rejcap3 operands noaccum fastdiv 2>&1 | grep -v amdgpu.ids > $O/probe_phases.txt; cat $O/probe_phases.txt
# the same table for the Float64 kernel at the reference's published configuration (1920x1080, depth 16) and at 4K
DT=f64 DEPTH=16 SPP=200 bash tools/gpu_probe_phases.sh base mfma nomfma eval extract resolve reject operands noaccum fastdiv 2>&1 | grep -v amdgpu.ids > $O/probe_phases_f64.txt; cat $O/probe_phases_f64.txt
DT=f64 WIDTH=3840 SPP=50 bash tools/gpu_probe_phases.sh base mfma resolve noaccum fastdiv 2>&1 | grep -v amdgpu.ids >> $O/probe_phases_f64.txt; tail -6 $O/probe_phases_f64.txt
# the ray-pool kernel's own stage profile (batches, fill, wave-cycles per stage) next to the lane loop's phase profile
(RTW_ENABLE_TEST_AIDS=1 RTW_PHASE_PROFILE=1 python tools/gpu_quick.py f32 1920 1000 50 pool 1; RTW_ENABLE_TEST_AIDS=1 RTW_PHASE_PROFILE=1 python tools/gpu_quick.py f32 1920 1000 50 plain 1) 2>&1 | grep -E "profile\]|kernel" > $O/pool_stage_profile.txt; cat $O/pool_stage_profile.txt
