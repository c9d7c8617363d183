#!/bin/bash
# GPU parity tests + a quick timing; every step under its own timeout (a protocol bug must not hang the box)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r2
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout ${1:-900} python -m pytest tests -m gpu -x -q 2>&1 | tail -15
timeout 120 python tools/gpu_quick.py f32 1920 100 50 2>&1 | tail -3
timeout 120 python tools/gpu_quick.py f64 1920 40 50 2>&1 | tail -3
