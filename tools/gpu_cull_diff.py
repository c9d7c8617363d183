#!/usr/bin/env python3
"""Find pixels where the group-cull image differs from the plain image (they must not): python tools/gpu_cull_diff.py f32 1920 1000"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rtw_amd as R
dt, W, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
depth = int(sys.argv[4]) if len(sys.argv) > 4 else 50
T = np.float64 if dt == "f64" else np.float32
H = R.image_height(W)
R.reseed(); scene = R.scene_random_spheres(elem_type=T); cam = R.t_cam1(elem_type=T)
rd = R.DeviceRenderer(scene, cam, device=0)
tt = torch.float64 if dt == "f64" else torch.float32
a = torch.empty(H * W * 3, dtype=tt, device="cuda:0"); b = torch.empty_like(a)
st = torch.cuda.current_stream()
rd.render_into(a.data_ptr(), W, spp, depth=depth, seed=1, stream=st.cuda_stream, gamma=False); sa = rd.stats()
rd.render_into(b.data_ptr(), W, spp, depth=depth, seed=1, stream=st.cuda_stream, gamma=False, group_cull=True); sb = rd.stats()
print("segments plain", sa["segments"], "cull", sb["segments"])
A = a.cpu().numpy().reshape(W, H, 3).transpose(1, 0, 2); B = b.cpu().numpy().reshape(W, H, 3).transpose(1, 0, 2)
bad = np.argwhere((A != B).any(axis=2))
print("differing pixels:", len(bad))
for i, j in bad[:20]:
    print("pixel i=%d j=%d (0-based)  plain %s  cull %s  diff*spp %s" % (i, j, A[i, j], B[i, j], (A[i, j].astype(np.float64) - B[i, j]) * spp))
