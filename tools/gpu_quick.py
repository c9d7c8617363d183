#!/usr/bin/env python3
"""Quick device-resident render timing: python tools/gpu_quick.py f64 3840 8 [depth] [plain|cull|pool] [reps] -> kernel ms, Msamples/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rtw_amd as R

dt, W, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
depth = int(sys.argv[4]) if len(sys.argv) > 4 else 50
cull = len(sys.argv) > 5 and sys.argv[5] == "cull"
pool = len(sys.argv) > 5 and sys.argv[5] == "pool"
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 2
nch = int(os.environ.get("RTW_QUICK_CHUNKS", "0"))           # sample chunks per pixel (0 = the library's default rule)
T = np.float64 if dt == "f64" else np.float32
H = R.image_height(W)
R.reseed()
scene = R.scene_random_spheres(elem_type=T)
cam = R.t_cam1(elem_type=T)
rd = R.DeviceRenderer(scene, cam, device=0)
fb = torch.empty(H * W * 3, dtype=torch.float64 if dt == "f64" else torch.float32, device="cuda:0")
st = torch.cuda.current_stream()
for rep in range(reps):
    rd.render_into(fb.data_ptr(), W, spp, depth=depth, seed=1, stream=st.cuda_stream, group_cull=cull, ray_pool=pool, n_chunks=nch)
    s = rd.stats()
    print(f"{dt} {W}x{H} spp {spp} depth {depth} cull {cull}: kernel {s['kernel_ms']:.2f} ms total {s['total_ms']:.2f} ms  "
          f"{W*H*spp/s['kernel_ms']/1e3:.1f} Msamples/s  segs/sample {s['segments']/(W*H*spp):.4f}  tests {s['sphere_tests']:.4g} "
          f"grid {s['grid_blocks']} chunks {s['n_chunks']}", flush=True)
print("checksum", float(fb.double().sum()))
