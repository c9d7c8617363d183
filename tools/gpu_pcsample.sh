#!/bin/bash
# PC sampling of the trace kernel (rocprofv3 beta): where the waves' program counters are, by source line.
#   usage: tools/gpu_pcsample.sh [method=host_trap|stochastic] [spp=200] [extra gpu_quick args...]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; M=${1:-host_trap}; SPP=${2:-200}; O=$R/gpurun_out/pcs_$M; rm -rf $O; mkdir -p $O
[ -f /tmp/librtw_g.so ] || make -s -j8 -C raytracingweekend.jl_amd/csrc -B OUT=/tmp/librtw_g.so EXTRA="-gline-tables-only" 2>&1 | grep -E "error"
if [ "$M" = stochastic ]; then UNIT=cycles; IV=${PCS_INTERVAL:-1048576}; else UNIT=time; IV=${PCS_INTERVAL:-1000}; fi
cd /tmp && export TMPDIR=/tmp
RTW_HIP_LIB=/tmp/librtw_g.so ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1 timeout 180 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $M --pc-sampling-unit $UNIT --pc-sampling-interval $IV \
    --kernel-trace --output-format csv -d $O -o p -- python $R/tools/gpu_quick.py f32 1920 $SPP 50 ${3:-plain} 1 > $O/log.txt 2>&1
echo "rc=$?"; tail -5 $O/log.txt; ls -la $O | head; for f in $O/*pc_sampling*.csv; do [ -f $f ] && { wc -l $f; head -5 $f; }; done
