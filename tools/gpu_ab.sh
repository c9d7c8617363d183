#!/bin/bash
# A/B library variants built on the GPU box: kernel time (bench.py) and end-of-queue drain of each, two rounds.
#   usage: tools/gpu_ab.sh "<name>=<EXTRA compiler flags>" ...      env: BENCH_ARGS (extra bench.py flags), DRAIN=1 (drain profile instead)
#   e.g.   tools/gpu_ab.sh "base=" "claim1=-DRTW_JOB_CLAIM=1u" "w6=-DRTW_TRACE_WAVES_MFMA_F32=6"
# Build-time switches worth knowing: RTW_TRACE_WAVES_MFMA_F32/_F64 (occupancy), RTW_SCAN_GROUP (1|4), RTW_SCAN_SKIP (0|1),
# RTW_JOB_CLAIM (1..16: most positions per claim), RTW_CLAIM_TAIL (guided-claim divisor); run-time: RTW_JOB_PIXELS (1|4|8|16), RTW_SCAN=valu.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  make -s -C raytracingweekend.jl_amd/csrc -B OUT=/tmp/librtw_$name.so EXTRA="$flags" 2>&1 | grep -E "error"
done
for round in 1 2; do
  for spec in "$@"; do
    name=${spec%%=*}
    if [ -n "$DRAIN" ]; then
      echo "$name: $(RTW_HIP_LIB=/tmp/librtw_$name.so RTW_ENABLE_TEST_AIDS=1 RTW_DRAIN_PROFILE=1 timeout ${AB_TIMEOUT:-300} python tools/gpu_quick.py ${DT:-f32} 1920 ${SPP:-1000} 50 plain 2 2>&1 | grep -E "drain profile\] [0-9]+ waves" | tail -1 | sed 's/.*kernel span/span/')"
    else
      echo "$name: $(RTW_HIP_LIB=/tmp/librtw_$name.so timeout ${AB_TIMEOUT:-300} python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras $BENCH_ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'Msamples/s  frac', d['roofline']['frac'], 'kernel_ms', d['roofline']['kernel_ms'])")"
    fi
  done
done
