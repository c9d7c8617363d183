#!/bin/bash
# A/B on ONE box (box-to-box spread is ~5 %): baseline library vs the current build, same workload
# usage: gpu_ab2.sh [dtype] [width] [spp] [libs...]   (libs default: build/variants/old.so and the product .so)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
DT=${1:-f32}; W=${2:-1920}; SPP=${3:-1000}; shift 3
LIBS=${@:-"build/variants/old.so raytracingweekend.jl_amd/lib/librtw_hip.so"}
for lib in $LIBS; do
  echo "== $lib"; RTW_HIP_LIB=$R/$lib timeout 300 python tools/gpu_quick.py $DT $W $SPP 50 plain 3 2>&1 | grep -E "Msamples|checksum|rror" | tail -3
done
