#!/usr/bin/env python3
"""stress: many concurrent two-stream renders (full + compact shard) and threads, compare every result"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import rtw_amd as R
from conftest import load_golden
T = np.float32
g = load_golden("cfg2_random_320x180_64spp_d16_f32")
R.reseed()
dr = R.DeviceRenderer(R.scene_random_spheres(elem_type=T), R.t_cam1(elem_type=T), device=0)
W, H = 320, 180
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 100
gold = np.ascontiguousarray(g["image"].transpose(1, 0, 2)).reshape(-1)
idx = R.compact_to_frame_index(W, 1, 3)
n_local = R.local_tile_count(W, 1, 3)
exp_comp = np.full((n_local * 64, 3), -7.0, np.float32)
exp_comp[idx >= 0] = gold.reshape(-1, 3)[idx[idx >= 0]]
bad = 0
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for it in range(n_iter):
    full = torch.empty(H * W * 3, dtype=torch.float32, device="cuda:0")
    comp = torch.full((n_local * 64 * 3,), -7.0, dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    dr.render_into(full.data_ptr(), W, 64, depth=16, seed=1, n_chunks=g["n_chunks"], stream=s1.cuda_stream)
    dr.render_into(comp.data_ptr(), W, 64, depth=16, seed=1, n_chunks=g["n_chunks"], shard_index=1, shard_count=3, stream=s2.cuda_stream, compact=True)
    st2 = dr.stats()
    s1.synchronize(); s2.synchronize()
    a = full.cpu().numpy(); c = comp.cpu().numpy().reshape(-1, 3)
    e1 = int((a != gold).sum()); e2 = int((c != exp_comp).sum())
    if e1 or e2 or st2["samples"] != int((idx >= 0).sum()) * 64:
        bad += 1
        w = np.flatnonzero(a != gold)[:6]
        print(f"iter {it}: full frame wrong channels {e1}, compact wrong {e2}, samples {st2['samples']} first idx {w} got {a[w]} want {gold[w]}", flush=True)
        if e2:
            w2 = np.argwhere(c != exp_comp)[:6]
            print("   compact:", w2.tolist(), c[w2[:, 0], w2[:, 1]], exp_comp[w2[:, 0], w2[:, 1]])
print("iterations", n_iter, "bad", bad)
