#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r2
rocm-smi --showclocks 2>/dev/null | head -20 > gpurun_out/r2/smi.txt
nproc > gpurun_out/r2/host.txt; lscpu | head -20 >> gpurun_out/r2/host.txt
./build/ubench_issue > gpurun_out/r2/ubench_issue.txt 2>&1
python tools/gpu_quick.py f32 1920 100 50 > gpurun_out/r2/quick.txt 2>&1
python tools/gpu_quick.py f64 1920 40 50 >> gpurun_out/r2/quick.txt 2>&1
python tools/gpu_quick.py f64 3840 20 50 >> gpurun_out/r2/quick.txt 2>&1
python tools/gpu_quick.py f64 3840 20 50 cull >> gpurun_out/r2/quick.txt 2>&1
RTW_PHASE_PROFILE=1 python tools/gpu_quick.py f32 1920 50 50 >> gpurun_out/r2/quick.txt 2>&1
RTW_PHASE_PROFILE=1 python tools/gpu_quick.py f64 1920 20 50 >> gpurun_out/r2/quick.txt 2>&1
cat gpurun_out/r2/ubench_issue.txt gpurun_out/r2/quick.txt
