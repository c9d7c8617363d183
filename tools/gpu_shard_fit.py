#!/usr/bin/env python3
"""kernel time of shard 0 of N for N = 1,2,4,8,16,32 (job_pixels given): fixed cost vs proportional cost"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
import rtw_amd as R
from rtw_amd import _capi
jp = int(sys.argv[1]) if len(sys.argv) > 1 else 4
nchunks = int(sys.argv[3]) if len(sys.argv) > 3 else 0
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
W, T = 1920, np.float32
H = R.image_height(W)
R.reseed(); scene = R.scene_random_spheres(elem_type=T); cam = R.t_cam1(elem_type=T)
rd = R.DeviceRenderer(scene, cam, device=0)
fb = torch.empty(H * W * 3, dtype=torch.float32, device="cuda:0")
L = _capi.lib()
def run(idx, cnt):
    P = _capi.make_params(W, H, spp, 50, 1, nchunks, idx, cnt, -1, 1, 0, job_pixels=jp)
    _capi.check(L.rtw_render_device_f32(rd.handle, C.byref(rd.cam), C.byref(P), C.c_void_p(fb.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    s = rd.stats()
    return s["kernel_ms"], s["segments"]
run(0, 1)
k1, s1 = run(0, 1)
print(f"job_pixels {jp} spp {spp} n_chunks {nchunks}: full {k1:.2f} ms")
for n in (2, 8, 32):
    k, s = run(0, n); k2, _ = run(0, n)
    ideal = k1 * s / s1
    print(f"  1/{n:<2d}: {k:7.2f} / {k2:7.2f} ms  ideal {ideal:7.2f}  extra {min(k,k2)-ideal:6.2f} ms  eff {ideal/min(k,k2):.3f}")
