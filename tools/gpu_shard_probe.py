#!/usr/bin/env python3
"""Kernel time of one shard of N for each job size: python tools/gpu_shard_probe.py 8 [f32] [1920] [1000]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
import rtw_amd as R
from rtw_amd import _capi
N = int(sys.argv[1]); dt = sys.argv[2] if len(sys.argv) > 2 else "f32"; W = int(sys.argv[3]) if len(sys.argv) > 3 else 1920; spp = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
T = np.float64 if dt == "f64" else np.float32
H = R.image_height(W)
R.reseed(); scene = R.scene_random_spheres(elem_type=T); cam = R.t_cam1(elem_type=T)
rd = R.DeviceRenderer(scene, cam, device=0)
fb = torch.empty(H * W * 3, dtype=torch.float64 if dt == "f64" else torch.float32, device="cuda:0")
L = _capi.lib()
def run(idx, cnt, jp, cull=False):
    P = _capi.make_params(W, H, spp, 50, 1, 0, idx, cnt, -1, 1, 1 if cull else 0, job_pixels=jp)
    fn = L.rtw_render_device_f64 if dt == "f64" else L.rtw_render_device_f32
    torch.cuda.synchronize(); t = time.perf_counter()
    _capi.check(fn(rd.handle, C.byref(rd.cam), C.byref(P), C.c_void_p(fb.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize(); wall = (time.perf_counter() - t) * 1e3
    s = rd.stats()
    return s["kernel_ms"], wall, s["grid_blocks"], s["segments"]
k1, w1, g, seg1 = run(0, 1, 0); k1, w1, g, seg1 = run(0, 1, 0)
print(f"full frame: kernel {k1:.1f} ms wall {w1:.1f} ms grid {g} segments {seg1}")
for jp in (16, 4, 1, 0):
    for idx in (0, N - 1):
        k, w, g, seg = run(idx, N, jp)
        print(f"shard {idx} of {N} job_pixels {jp:2d}: kernel {k:7.2f} ms wall {w:7.2f} ms grid {g} segments share {seg/seg1:.4f} -> efficiency vs segments share {k1*seg/seg1/k:.3f}")
