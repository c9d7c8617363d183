#!/bin/bash
# The ray-pool kernel is a build option since round 6 (`make POOL=1`): build such a library into /tmp on the GPU box, run ITS tests
# (tests/test_gpu_pool.py, skipped in the default `-m gpu` run) and the timing comparison against the lane-loop kernel.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
make -s -j8 -C raytracingweekend.jl_amd/csrc -B POOL=1 OUT=/tmp/librtw_pool.so 2>&1 | grep -E "error"
export RTW_HIP_LIB=/tmp/librtw_pool.so RTW_TEST_POOL=1
timeout 900 python -m pytest tests/test_gpu_pool.py -m gpu -q -x 2>&1 | tail -2
python tools/gpu_pool_check.py 320,64,16 1920,100,50 2>&1 | tail -8
python bench.py --ray-pool --steps 2 --warmup 1 --no-cpu-baseline --no-extras | cut -c1-200
