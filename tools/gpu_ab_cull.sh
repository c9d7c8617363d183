#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for lib in "$@"; do echo "== $lib"; RTW_HIP_LIB=$R/$lib timeout 300 python tools/gpu_quick.py f32 1920 1000 50 cull 3 2>&1 | grep -E "Msamples|checksum|rror" | tail -3; done
