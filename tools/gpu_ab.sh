#!/bin/bash
# A/B kernel build variants on the GPU box: tools/gpu_ab.sh "<name>=<EXTRA flags>" ...   (BENCH_ARGS for bench.py)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  make -s -C raytracingweekend.jl_amd/csrc -B OUT=/tmp/librtw_$name.so EXTRA="$flags" 2>&1 | grep -E "error" 
done
for round in 1 2; do
  for spec in "$@"; do
    name=${spec%%=*}
    echo -n "$name: "
    RTW_HIP_LIB=/tmp/librtw_$name.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline $BENCH_ARGS | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'Msamples/s  frac', d['roofline']['frac'], 'kernel_ms', d['roofline']['kernel_ms'])"
  done
done
