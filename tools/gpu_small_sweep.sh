#!/bin/bash
# Job size x chunk count for the small-frame / small-shard regime (the rule in rtw_launch.hip), and where a tiny frame's time goes.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
P="timeout 120 python tools/gpu_small_probe.py"
export RTW_ENABLE_TEST_AIDS=1
echo "== configs[1] 320x180x64 d16 f32: chunks x job pixels"
for nch in 16 32 64; do for jp in 1 4 8 16; do $P random f32 320 64 16 $nch $jp 2>&1 | tail -1; done; done
echo "== 200x112x32 d16 f64"
for nch in 16 32; do for jp in 1 4 8 16; do $P random f64 200 32 16 $nch $jp 2>&1 | tail -1; done; done
echo "== 1/8 shard of 1080p x 1000 spp"
for jp in 1 4 16; do $P random f32 1920 1000 50 0 $jp 0 3 8 2>&1 | tail -1; done
echo "== 1/8 shard of 1080p x 64 spp (64 chunks)"
for jp in 1 4 16; do $P random f32 1920 64 50 64 $jp 0 5 8 2>&1 | tail -1; done
echo "== tiny 96x54x16 d4 f32"
for jp in 1 4 8 16; do $P two f32 96 16 4 0 $jp 2>&1 | tail -1; done
for g in 32 64 128; do RTW_GRID_BLOCKS=$g $P two f32 96 16 4 0 0 2>&1 | tail -1; done
RTW_DRAIN_PROFILE=1 $P two f32 96 16 4 0 0 0 2 2>&1 | grep -E "drain profile\] [0-9]+ waves|kernel" | tail -2
RTW_DRAIN_PROFILE=1 $P random f32 320 64 16 64 4 0 2 2>&1 | grep -E "drain profile\] [0-9]+ waves|kernel" | tail -2
