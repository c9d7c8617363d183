#!/usr/bin/env python3
"""Pool kernel vs lane-loop kernel: bit-identity and timing.  python tools/gpu_pool_check.py [W spp depth] ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rtw_amd as R

def one(W, spp, depth, scene_fn=None, cam_fn=None, reps=2):
    T = np.float32
    H = R.image_height(W)
    R.reseed()
    scene = (scene_fn or R.scene_random_spheres)(elem_type=T)
    cam = (cam_fn or R.t_cam1)(elem_type=T)
    rd = R.DeviceRenderer(scene, cam, device=0)
    st = torch.cuda.current_stream()
    out = {}
    for mode in ("lane", "pool"):
        fb = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda:0")
        for rep in range(reps):
            rd.render_into(fb.data_ptr(), W, spp, depth=depth, seed=1, stream=st.cuda_stream, ray_pool=(mode == "pool"))
            s = rd.stats()
        out[mode] = (fb.clone(), s)
        print(f"  {mode}: kernel {s['kernel_ms']:.2f} ms  {W*H*spp/s['kernel_ms']/1e3:.1f} Msamples/s  segs {s['segments']} samples {s['samples']} grid {s['grid_blocks']}x{s['block_threads']}", flush=True)
    a, b = out["lane"][0], out["pool"][0]
    same = torch.equal(a, b)
    nd = int((a != b).sum())
    print(f"{W}x{H} spp {spp} depth {depth}: identical={same} differing={nd} segs_equal={out['lane'][1]['segments']==out['pool'][1]['segments']}", flush=True)
    return same

if __name__ == "__main__":
    args = sys.argv[1:]
    cases = [(96, 16, 4), (320, 64, 16), (640, 100, 50)] if not args else [tuple(int(x) for x in a.split(",")) for a in args]
    ok = True
    for (W, spp, depth) in cases:
        ok &= one(W, spp, depth)
    sys.exit(0 if ok else 1)
