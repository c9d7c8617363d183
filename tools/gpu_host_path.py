#!/usr/bin/env python3
"""Host-buffer entry point (rtw_render_*): wall time minus kernel time per call, first call vs context reuse, one device and
the in-library multi-device path emulated with a repeated ordinal.  usage: python tools/gpu_host_path.py [spp=100]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rtw_amd as R
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for T in (np.float32, np.float64):
    R.reseed()
    scene = R.scene_random_spheres(elem_type=T)
    cam = R.t_cam1(elem_type=T)
    ref = None
    for devices in (None, [0] * 2, [0] * 8):
        for k in range(4):
            t = time.perf_counter()
            img = R.render(scene, cam, 1920, spp, depth=50, seed=1, devices=devices)
            wall = (time.perf_counter() - t) * 1e3
            st = R.last_stats()
            print(f"{T.__name__} devices={devices and len(devices)} call {k}: wall {wall:8.3f} ms  kernel(max) {st['kernel_ms']:8.3f} ms  overhead {wall - st['kernel_ms']:8.3f} ms", flush=True)
        if ref is None:
            ref = img.copy()
        assert np.array_equal(img, ref)
