#!/usr/bin/env python3
"""full-frame kernel time + drain profile for (job_pixels, n_chunks) pairs: python tools/gpu_drain.py 16:125 4:500 ..."""
import os, sys
os.environ["RTW_DRAIN_PROFILE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
import rtw_amd as R
from rtw_amd import _capi
W, T, spp = 1920, np.float32, 1000
H = R.image_height(W)
R.reseed(); scene = R.scene_random_spheres(elem_type=T); cam = R.t_cam1(elem_type=T)
rd = R.DeviceRenderer(scene, cam, device=0)
fb = torch.empty(H * W * 3, dtype=torch.float32, device="cuda:0")
L = _capi.lib()
def run(jp, nch, idx=0, cnt=1):
    P = _capi.make_params(W, H, spp, 50, 1, nch, idx, cnt, -1, 1, 0, job_pixels=jp)
    _capi.check(L.rtw_render_device_f32(rd.handle, C.byref(rd.cam), C.byref(P), C.c_void_p(fb.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return rd.stats()["kernel_ms"]
run(16, 125)
for a in sys.argv[1:]:
    jp, nch = [int(x) for x in a.split(":")]
    print(f"== job_pixels {jp} n_chunks {nch}: full {run(jp, nch):.2f} ms, 1/8 shard {run(jp, nch, 0, 8):.2f} ms", file=sys.stderr)
