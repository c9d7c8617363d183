#!/usr/bin/env python3
"""Generate tests/golden/*.npz: inputs + expected outputs of the hot path, produced by the CPU
oracle (oracle/) in PIXEL_STREAM mode -- the mode the device reproduces.

The reference itself (Julia) cannot be run in this image, and its own tests hold no image, RNG
or intersection vector; its only value assertions are the unit KATs, which
live in tests/test_oracle_kats.py.  These fixtures pin the oracle against regressions and give
the GPU tests committed vectors to match.  Re-run:  python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import rtw_amd as R          # noqa: E402  (host mirror: scene / camera producers)
import rtw_oracle as O       # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
MODES = ("reference", "contract", "reference_fma2")       # the numerics modes of the library (include/rtw_hip.h RTW_FLAG_NUMERICS_*)


def cam_dict(cam):
    d = {"cam_" + k: np.asarray(getattr(cam, k)) for k in O.CAM_FIELDS}
    d["cam_lens_radius"] = np.asarray(cam.lens_radius)
    return d


def case(name, flat, cam, width, spp, depth, T, seed=1, n_chunks=0, both_orders=True):
    height = R.image_height(width)
    nch = n_chunks or O.default_n_chunks(spp)
    d = {"scene_" + k: np.asarray(v) for k, v in flat.items()}
    d.update(cam_dict(cam))
    d.update(width=width, height=height, spp=spp, depth=depth, seed=seed, n_chunks=nch)
    # one expected image + counters per numerics mode of the ray-sphere test (include/rtw_hip.h RTW_FLAG_NUMERICS_*):
    # "reference" (the default) under the plain keys, the others with the mode's name as a suffix
    for mode in MODES:
        suf = "" if mode == "reference" else "_" + mode
        img, st = O.render(flat, cam, width, height, spp, T=T, max_depth=depth, seed=seed, n_chunks=nch,
                           product_order=O.PRODUCT_FORWARD, numerics=mode)
        d["image" + suf] = np.ascontiguousarray(img)
        d["segments" + suf] = st["segments"]
        d["rng_draws" + suf] = st["rng_draws"]
        if mode == "reference":
            img0, st0 = img, st
        if both_orders:
            img_ref, _ = O.render(flat, cam, width, height, spp, T=T, max_depth=depth, seed=seed, n_chunks=nch,
                                  product_order=O.PRODUCT_REFERENCE, numerics=mode)
            d["image_reference_order" + suf] = np.ascontiguousarray(img_ref)
    img, st = img0, st0
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **d)
    print(f"{name}: {img.shape} {img.dtype} segments={st['segments']} -> {os.path.getsize(path)/1024:.0f} KiB")


def main():
    os.makedirs(OUT, exist_ok=True)
    f32, f64 = np.float32, np.float64
    # BASELINE.json configs[0]: scene_2_spheres, 96x54, 16 spp, depth 4, Float32
    case("cfg1_2spheres_96x54_16spp_d4_f32", R.flatten_scene(R.scene_2_spheres(elem_type=f32), f32),
         R.t_default_cam(elem_type=f32), 96, 16, 4, f32)
    # the reference's own smoke render (test/runtests.jl:194): Float64, depth 16
    case("smoke_2spheres_96x54_16spp_d16_f64", R.flatten_scene(R.scene_2_spheres(elem_type=f64), f64),
         R.t_default_cam(elem_type=f64), 96, 16, 16, f64)
    # BASELINE.json configs[1]: scene_random_spheres, 320x180, 64 spp, depth 16, Float32
    R.reseed()
    rs32 = R.flatten_scene(R.scene_random_spheres(elem_type=f32), f32)
    case("cfg2_random_320x180_64spp_d16_f32", rs32, R.t_cam1(elem_type=f32), 320, 64, 16, f32, both_orders=False)
    # small Float64 random-spheres case (configs[4] is the fp64 path), depth 50
    R.reseed()
    rs64 = R.flatten_scene(R.scene_random_spheres(elem_type=f64), f64)
    case("random_64x36_8spp_d50_f64", rs64, R.t_cam1(elem_type=f64), 64, 8, 50, f64)
    # material / camera coverage: hollow glass bubble, fuzzy metal, wide aperture
    case("diel_bubble_96x54_8spp_d16_f32", R.flatten_scene(R.scene_diel_spheres(-0.5, elem_type=f32), f32),
         R.t_cam2(elem_type=f32), 96, 8, 16, f32)
    case("metal4_96x54_8spp_d16_f32", R.flatten_scene(R.scene_4_spheres(elem_type=f32), f32),
         R.t_default_cam(elem_type=f32), 96, 8, 16, f32)
    # the remaining reference scenes (src/scenes.jl:25-47; src/proto/proto.jl:94,104,271): both hollow-glass
    # signs and the Float64-only blue/red scene (R = cos(pi/4) is a Float64 literal there)
    case("diel_plus_96x54_8spp_d16_f32", R.flatten_scene(R.scene_diel_spheres(0.5, elem_type=f32), f32),
         R.t_cam2(elem_type=f32), 96, 8, 16, f32)
    case("blue_red_96x54_8spp_d16_f64", R.flatten_scene(R.scene_blue_red_spheres(elem_type=f64), f64),
         R.t_default_cam(elem_type=f64), 96, 8, 16, f64)
    # provisional RNG / scene vectors (unpinned against Julia: DESIGN.md section 3)
    st = O.rng_seed(1)
    u64 = [O.rng_next(st) for _ in range(16)]
    st = O.rng_seed(1); f32s = [O.rng_float(st, f32) for _ in range(16)]
    st = O.rng_seed(1); f64s = [O.rng_float(st, f64) for _ in range(16)]
    st2 = O.rng_seed(2); u64_2 = [O.rng_next(st2) for _ in range(16)]
    np.savez_compressed(os.path.join(OUT, "rng_provisional.npz"), state_seed1=O.rng_seed(1), state_seed2=O.rng_seed(2),
                        u64_seed1=np.array(u64, np.uint64), u64_seed2=np.array(u64_2, np.uint64),
                        f32_seed1=np.array(f32s, f32), f64_seed1=np.array(f64s, f64),
                        stream_1_0_0=O.rng_stream(1, 0, 0), stream_1_12345_3=O.rng_stream(1, 12345, 3))
    print("rng_provisional written")


if __name__ == "__main__":
    main()
