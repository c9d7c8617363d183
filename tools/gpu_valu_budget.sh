#!/bin/bash
# The VALU budget of the headline kernel (tools/valu_budget.py): event counts of the phase-profile build + SQ_INSTS_VALU of the product kernel,
# same workload (1080p x SPP spp, depth 50, Float32, plain scan).   usage: tools/gpu_valu_budget.sh [spp=200] [out=gpurun_out/valu_budget.txt]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; SPP=${1:-200}; OUT=${2:-$R/gpurun_out/valu_budget.txt}; MODE=${3:-plain}     # plain | cull
CF=0; CA=""; [ "$MODE" = cull ] && { CF=1; CA="--cull"; }
RTW_ENABLE_TEST_AIDS=1 RTW_PHASE_PROFILE=1 python tools/gpu_small_probe.py random f32 1920 $SPP 50 0 0 $CF 1 2>&1 | grep -E "phase counts|phase profile" | tail -6 > /tmp/phase_counts.txt
M=$(bash tools/gpu_pmc_quick.sh f32 $SPP $MODE | tee /tmp/pmcq.txt | sed 's/.*VALU \([0-9]*\) .*/\1/')
{ echo "# $(date -u +%F) 1920x1080 x $SPP spp, depth 50, Float32, $MODE scan, reference numerics"; cat /tmp/pmcq.txt; cat /tmp/phase_counts.txt; python tools/valu_budget.py $CA /tmp/phase_counts.txt $M; } > $OUT 2>&1
cat $OUT
