#!/bin/bash
# Scheduler changes (slot protocol, hand-out rounds, static claims): parity first (under a timeout: a protocol bug is a hang), then the
# workloads whose lane utilisation the profile build reports.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_round2.py tests/test_gpu_round3.py -m gpu -x -q 2>&1 | tail -3
P="timeout 120 python tools/gpu_small_probe.py"
for a in "random f32 1920 200 50 0 0 0 3" "random f32 1920 200 50 200 0 0 3" "random f32 1920 64 50 0 0 0 4" "random f32 1920 64 50 64 0 0 4" "random f32 320 64 16 0 0 0 30" "random f32 320 64 16 64 0 0 30" "random f32 320 64 16 32 0 0 30" "random f64 1920 200 16 0 0 0 3" "random f64 1920 200 16 200 0 0 3" "two f32 96 16 4 0 0 0 30" "two f32 96 16 4 0 16 0 30" "two f64 96 16 16 0 0 0 30" "random f64 200 32 16 0 0 0 30" "random f32 1920 1000 50 0 0 0 2" "random f32 1920 1000 50 0 0 1 2"; do $P $a 2>&1 | tail -1; done
export RTW_ENABLE_TEST_AIDS=1 RTW_PHASE_PROFILE=1
for a in "random f32 1920 1000 50 0 0 0 1" "random f32 1920 200 50 0 0 0 1" "random f32 1920 64 50 0 0 0 1" "random f32 320 64 16 0 0 0 1" "random f64 1920 200 16 200 0 0 1"; do $P $a 2>&1 | grep -E "lane loop|kernel" | tail -2; done
