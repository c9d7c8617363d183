#!/usr/bin/env python3
"""Replay pixels with the oracle's unit functions, collect every ray segment, and check the device scans
(ops 8 plain-global, 10 plain-LDS, 11 group-cull) against the oracle on exactly those rays.
usage: python tools/debug_cull_pixel.py f32 1920 1000 50  i,j [i,j ...]   (0-based pixel coordinates)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import rtw_amd as R, rtw_oracle as O
from test_gpu_units import run_unit

dt, W, spp, depth = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
T = np.float64 if dt == "f64" else np.float32
H = R.image_height(W)
R.reseed(); scene = R.scene_random_spheres(elem_type=T); cam = R.t_cam1(elem_type=T)
flat = R.flatten_scene(scene, T)
nch = O.default_n_chunks(spp); cs = -(-spp // nch); nch = -(-spp // cs)
rays = []
for arg in sys.argv[5:]:
    i0, j0 = [int(x) for x in arg.split(",")]
    u = T(np.float64(j0 + 1) / np.float64(W)); v = T(np.float64(H - (i0 + 1)) / np.float64(H))
    pix = j0 * H + i0
    for ch in range(nch):
        st = O.rng_stream(1, pix, ch)
        for s in range(ch * cs, min(spp, (ch + 1) * cs)):
            du = dv = T(0)
            if s != 0:
                du = O.rng_float(st, T) / T(np.float32(W)); dv = O.rng_float(st, T) / T(np.float32(H))
            ray, st = O.get_ray(cam, u + du, v + dv, st, T)
            o, d = ray[:3].copy(), ray[3:].copy()
            for k in range(depth):
                rays.append((i0, j0, ch, s, k, *o, *d))
                idx, rec = O.hit_world(flat, o, d, T(1e-4), np.inf, T)
                if idx < 0:
                    break
                alb = np.array([flat["ar"][idx], flat["ag"][idx], flat["ab"][idx]], T)
                out, st = O.scatter(int(flat["kind"][idx]), alb, flat["param"][idx], d, rec, st, T)
                o, d = out[:3].astype(T), out[3:6].astype(T)
rays = np.array(rays, np.float64)
print(len(rays), "ray segments collected")
r6 = rays[:, 5:11].astype(T)
ref_idx, ref_t = O.hit_world_batch(flat, r6, T(1e-4), np.inf, T)
x = np.concatenate([r6.astype(np.float64), np.full((len(r6), 1), float(T(1e-4))), np.full((len(r6), 1), np.inf)], 1)
for op in (8, 10, 11):
    y = run_unit(op, x, 9, T, flat=flat)
    bad = (y[:, 0].astype(np.int64) != ref_idx) | ((ref_idx >= 0) & (y[:, 1] != ref_t.astype(np.float64)))
    print("op", op, "mismatches:", int(bad.sum()))
    for q in np.flatnonzero(bad)[:10]:
        i = int(ref_idx[q]); g = int(y[q, 0])
        print("  pixel/chunk/sample/bounce", rays[q, :5].astype(int), "o", repr(r6[q, :3]), "d", repr(r6[q, 3:]))
        print("   oracle idx %d t %r | device idx %d t %r" % (i, ref_t[q], g, y[q, 1]))
        for name, k in (("oracle", i), ("device", g)):
            if k >= 0:
                print("   %s sphere %d: c=(%r, %r, %r) r=%r kind %d" % (name, k, flat["cx"][k], flat["cy"][k], flat["cz"][k], flat["r"][k], flat["kind"][k]))
