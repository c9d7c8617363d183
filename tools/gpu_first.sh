#!/bin/bash
# first GPU pass: parity tests, inner-loop micro-benchmark, short bench, kernel trace
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/smi.log
nproc > gpurun_out/host.log; grep -m1 "model name" /proc/cpuinfo >> gpurun_out/host.log
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > gpurun_out/pytest.log
echo "pytest exit: $?" >> gpurun_out/pytest.log
timeout 300 ./build/ubench_scan 16 > gpurun_out/ubench.log 2>&1
timeout 600 python bench.py --steps 2 --warmup 1 --spp 200 --no-cpu-baseline > gpurun_out/bench_spp100.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o first -- python $R/bench.py --steps 1 --warmup 0 --spp 50 --no-cpu-baseline > $R/gpurun_out/rocprof.log 2>&1
ls -R $R/gpurun_out/prof | head -30 >> $R/gpurun_out/rocprof.log
tail -5 $R/gpurun_out/pytest.log; cat $R/gpurun_out/ubench.log; cat $R/gpurun_out/bench_spp100.log
