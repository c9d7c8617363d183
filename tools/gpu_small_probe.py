#!/usr/bin/env python3
"""Small-frame kernel time under the knobs that shape its tail (device-resident, median of REPS launches):
   python tools/gpu_small_probe.py <scene: random|two> <f32|f64> <W> <spp> <depth> [n_chunks=0] [job_pixels=0] [cull=0] [reps=30] [shards=1]
   env (with RTW_ENABLE_TEST_AIDS=1): RTW_GRID_BLOCKS, RTW_PHASE_PROFILE, RTW_DRAIN_PROFILE"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rtw_amd as R

sc, dt, W, spp, depth = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
nch = int(sys.argv[6]) if len(sys.argv) > 6 else 0
jp = int(sys.argv[7]) if len(sys.argv) > 7 else 0
cull = len(sys.argv) > 8 and sys.argv[8] == "1"
reps = int(sys.argv[9]) if len(sys.argv) > 9 else 30
shards = int(sys.argv[10]) if len(sys.argv) > 10 else 1          # render shard 0 of that many (what one rank of a multi-GPU render does)
T = np.float64 if dt == "f64" else np.float32
H = R.image_height(W)
R.reseed()
scene = R.scene_random_spheres(elem_type=T) if sc == "random" else R.scene_2_spheres(elem_type=T)
cam = R.t_cam1(elem_type=T) if sc == "random" else R.t_default_cam(elem_type=T)
rd = R.DeviceRenderer(scene, cam, device=0)
fb = torch.empty(H * W * 3, dtype=torch.float64 if dt == "f64" else torch.float32, device="cuda:0")
st = torch.cuda.current_stream()
ks = []
for rep in range(reps + 3):
    rd.render_into(fb.data_ptr(), W, spp, depth=depth, seed=1, stream=st.cuda_stream, group_cull=cull, n_chunks=nch, job_pixels=jp, shard_index=0, shard_count=shards)
    s = rd.stats()
    if rep >= 3:
        ks.append(s["kernel_ms"] * 1e3)
ks.sort()
n = W * H * spp // shards
print(f"{sc} {dt} {W}x{H} spp {spp} d{depth} chunks {s['n_chunks']} job_px {jp} cull {int(cull)} shard 1/{shards} grid {s['grid_blocks']} "
      f"(env grid {os.environ.get('RTW_GRID_BLOCKS', '-')}): kernel median {statistics.median(ks):8.1f} us  min {ks[0]:8.1f} us  "
      f"{n / statistics.median(ks):8.1f} Msamples/s  segs/sample {s['segments'] / n:.3f}", flush=True)
