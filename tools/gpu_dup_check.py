#!/usr/bin/env python3
"""Job duplication / loss check: the kernel's sample counter must equal W*H*spp for every render (a duplicated job does not change the image).
usage: python tools/gpu_dup_check.py [reps=200]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtw_amd as R
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
T = np.float32
bad = 0
cases = [(R.scene_4_spheres, R.t_default_cam, 200, 6, 8), (R.scene_2_spheres, R.t_default_cam, 96, 16, 4), (R.scene_random_spheres, R.t_cam1, 320, 64, 16),
         (R.scene_4_spheres, R.t_default_cam, 333, 3, 8), (R.scene_4_spheres, R.t_default_cam, 64, 40, 8)]
for mk, cam_fn, W, spp, depth in cases:
    R.reseed()
    scene, cam = mk(elem_type=T), cam_fn(elem_type=T)
    H = R.image_height(W)
    seen = {}
    for k in range(reps):
        R.render(scene, cam, W, spp, depth=depth)
        s = R.last_stats()
        seen[s["samples"]] = seen.get(s["samples"], 0) + 1
    ok = list(seen) == [W * H * spp]
    bad += 0 if ok else 1
    print(f"{mk.__name__} {W}x{H} spp {spp}: expected {W*H*spp}, seen {seen} grid {s['grid_blocks']} {'OK' if ok else 'WRONG'}", flush=True)
sys.exit(1 if bad else 0)
