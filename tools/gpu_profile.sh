#!/bin/bash
# The round's whole measurement set in ONE call on the GPU box (same box, same binary): what the driver runs at round end
# (smoke, pytest -m gpu, bench) plus the rocprofv3 summaries that are committed under profiles/.
#   usage: tools/gpu_profile.sh [tag=round6] [notests]
# Every rocprofv3 pass is its own run: --kernel-trace --stats only, or --pmc only with --kernel-trace (FETCH_SIZE and WRITE_SIZE
# do not fit one pass; MI355X_MICROARCH.md, HBM section).  Output: gpurun_out/<tag>/ -- kernel_stats CSVs, pmc_summary.json,
# hbm_traffic.json (FETCH_SIZE x 2 + WRITE_SIZE per launch, KiB -> bytes), bench_*.json.  Copy what is to be judged into profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; TAG=${1:-round6}; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
[ "$2" = "notests" ] || timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
B32="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
B64="python $R/bench.py --dtype f64 --width 3840 --steps 1 --warmup 0 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_f32 -o t -- $B32 > $O/trace_f32.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_f32_cull -o t -- $B32 --group-cull > $O/trace_f32_cull.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_f32_valu -o t -- $B32 --scan-valu > $O/trace_f32_valu.log 2>&1
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
pmc f64_sqA "$B64" GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU
pmc f64_mfma "$B64" SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16
python3 $R/tools/pmc_summarise.py $O
# the bench lines read the HBM traffic of THIS run (box-local copy; the merged gpurun_out/<tag>/hbm_traffic.json is what gets
# committed as profiles/r06_hbm_traffic.json (likewise r06_pmc_summary.json: `roofline.issue_busy`) -- `traffic_static` in the line says that the figure is not measured by bench.py itself)
cp $O/hbm_traffic.json $R/profiles/r06_hbm_traffic.json; cp $O/pmc_summary.json $R/profiles/r06_pmc_summary.json
cd $R
python bench.py > $O/bench_f32.json 2> $O/bench_f32.err; cut -c1-400 $O/bench_f32.json
python bench.py --dtype f64 --width 3840 --steps 2 --warmup 1 --no-extras > $O/bench_f64_4k.json 2> $O/bench_f64_4k.err; cut -c1-300 $O/bench_f64_4k.json
python bench.py --emulate-shard-of 8 --steps 3 --no-cpu-baseline > $O/bench_f32_shard8.json 2>/dev/null; cut -c1-300 $O/bench_f32_shard8.json
RTW_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | grep "^{" > $O/bench_f32_2ranks_one_device.json; cut -c1-300 $O/bench_f32_2ranks_one_device.json
python tools/kernel_resources.py > $O/kernel_resources.txt 2>&1; tail -12 $O/kernel_resources.txt
# the small-frame regime: the bench leg's figures are in bench_f32.json; here the kernel trace of the same calls
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_small -o t -- python $R/bench.py --small-frames 100 > $O/small_frames.json 2> $O/trace_small.log); cat $O/trace_small/*kernel_stats.csv
# the VALU budget of the headline kernel (static ISA counts x event counts of the phase-profile build vs SQ_INSTS_VALU)
bash tools/gpu_valu_budget.sh 1000 $O/valu_budget.txt | tail -30
bash tools/gpu_valu_budget.sh 200 $O/valu_budget_200spp.txt | tail -3
