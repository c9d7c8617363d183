#!/usr/bin/env python3
"""many host-buffer and device-resident renders (single and multi-device lists): free device memory must not drift"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rtw_amd as R
T = np.float32
scene = R.scene_4_spheres(elem_type=T); cam = R.t_default_cam(elem_type=T)
def free_mb():
    torch.cuda.synchronize(); f, t = torch.cuda.mem_get_info(); return f / 1e6
R.render(scene, cam, 96, 2)
f0 = free_mb()
for i in range(300):
    R.render(scene, cam, 96, 2, devices=[0, 0, 0] if i % 3 == 0 else None, group_cull=bool(i & 1))
f1 = free_mb()
dr = R.DeviceRenderer(scene, cam, device=0)
fb = torch.empty(96 * 54 * 3, dtype=torch.float32, device="cuda:0")
for i in range(2000):
    dr.render_into(fb.data_ptr(), 96, 2, stream=torch.cuda.current_stream().cuda_stream)
    if i % 100 == 0: dr.stats()
f2 = free_mb()
print(f"free MB: start {f0:.1f}, after 300 host renders {f1:.1f}, after 2000 async device renders {f2:.1f}")
assert abs(f1 - f0) < 64 and abs(f2 - f1) < 64, "device memory drift"
print("no drift")
