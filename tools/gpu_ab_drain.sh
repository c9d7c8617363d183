#!/bin/bash
# A/B library variants: kernel ms + end-of-queue drain.  usage: tools/gpu_ab_drain.sh "<name>=<EXTRA flags>" ...
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for spec in "$@"; do name=${spec%%=*}; flags=${spec#*=}; make -s -C raytracingweekend.jl_amd/csrc -B OUT=/tmp/librtw_$name.so EXTRA="$flags" 2>&1 | grep -E "error"; done
for round in 1 2; do for spec in "$@"; do name=${spec%%=*}
  echo "$name: $(RTW_HIP_LIB=/tmp/librtw_$name.so RTW_DRAIN_PROFILE=1 python tools/gpu_quick.py ${DT:-f32} 1920 ${SPP:-1000} 50 plain 2 2>&1 | grep -E "drain profile\] [0-9]+ waves" | tail -1 | sed 's/.*kernel span/span/')"
done; done
