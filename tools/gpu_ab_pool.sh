#!/bin/bash
# A/B of pool-kernel build variants on the GPU box: kernel ms at 1080p x SPP spp (+ the stage profile with PROF=1).
#   usage: tools/gpu_ab_pool.sh "<name>=<EXTRA compiler flags>" ...     env: SPP (default 300), PROF=1, MODE (pool|plain)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  make -s -j8 -C raytracingweekend.jl_amd/csrc -B POOL=1 OUT=/tmp/librtw_$name.so EXTRA="$flags" 2>&1 | grep -E "error"
done
for spec in "$@"; do
  name=${spec%%=*}
  if [ -n "$PROF" ]; then
    echo "== $name"; RTW_ENABLE_TEST_AIDS=1 RTW_PHASE_PROFILE=1 RTW_HIP_LIB=/tmp/librtw_$name.so timeout 200 python tools/gpu_quick.py f32 1920 ${SPP:-300} 50 ${MODE:-pool} 1 2>&1 | grep -E "pool profile|kernel"
  else
    echo "$name: $(RTW_HIP_LIB=/tmp/librtw_$name.so timeout 200 python tools/gpu_quick.py f32 1920 ${SPP:-300} 50 ${MODE:-pool} 2 2>&1 | grep kernel | tail -1 | sed 's/.*cull False: //; s/segs.*//')"
  fi
done
