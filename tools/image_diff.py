#!/usr/bin/env python3
"""image_diff.py A B        -- difference report of two images (.npy float arrays [H,W,3] or P6 .ppm)
image_diff.py t3 [outdir]  -- tier-T3 artefacts for the headline scene (needs the GPU): GPU (PIXEL_STREAM) vs the oracle's REF_SERIAL mode
                              (= the reference's own sampling order with `julia -t H`), scene_random_spheres / t_cam1, 320x180, 1024 spp,
                              depth 16: three PNGs + a JSON report with the tolerances of tests/test_gpu_round2.py::test_t3_...
                              (profiles/r02_t3_* were made this way)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np  # noqa: E402
from rtw_amd import imageio  # noqa: E402


def load(p):
    return np.load(p) if p.endswith(".npy") else imageio.load_ppm(p).astype(np.float64) / 255.0


def t3(argv):
    import rtw_amd as R
    import rtw_oracle as O
    out = argv[0] if argv else os.path.join(ROOT, "gpurun_out", "t3")
    os.makedirs(out, exist_ok=True)
    T = np.float32
    R.reseed(); scene = R.scene_random_spheres(elem_type=T); cam = R.t_cam1(elem_type=T); flat = R.flatten_scene(scene, T)
    W, H, spp, depth = 320, 180, 1024, 16
    report = {"workload": f"scene_random_spheres, t_cam1, {W}x{H}, {spp} spp, depth {depth}, Float32, linear radiance compared, gamma images written"}
    t = time.time(); A = R.render(scene, cam, W, spp, depth=depth, seed=1, gamma=False).astype(np.float64)
    B = R.render(scene, cam, W, spp, depth=depth, seed=2, gamma=False).astype(np.float64); tg = time.time() - t
    t = time.time()
    Rr, _ = O.render(flat, cam, W, H, spp, T=T, max_depth=depth, rng_mode=O.REF_SERIAL, ref_threads=H, product_order=O.PRODUCT_REFERENCE, gamma=False)
    Rr = Rr.astype(np.float64); tr = time.time() - t
    D1, D2 = A - B, A - Rr
    def blocks(x, f):
        hh, ww = (H // 8) * 8, (W // 8) * 8
        return f(x[:hh, :ww].reshape(hh // 8, 8, ww // 8, 8, 3), axis=(1, 3))
    var_hat = np.repeat(np.repeat(blocks(D1 ** 2, np.mean), 8, 0), 8, 1)
    z = np.abs(D2[:var_hat.shape[0], :var_hat.shape[1]]) / np.sqrt(np.maximum(var_hat, 1e-12))
    report.update(gpu_seconds_two_renders=round(tg, 2), oracle_ref_serial_seconds=round(tr, 1),
                  mean_D2=float(D2.mean()), sd_D2_over_sqrtN=float(D2.std() / np.sqrt(D2.size)),
                  variance_ratio_D2_over_D1=float((D2 ** 2).mean() / (D1 ** 2).mean()),
                  frac_channels_within_4p5_sigma=float((z <= 4.5).mean()),
                  max_abs_block_mean_D1=float(np.abs(blocks(D1, np.mean)).max()), max_abs_block_mean_D2=float(np.abs(blocks(D2, np.mean)).max()),
                  gamma_space=imageio.diff_report(np.sqrt(A), np.sqrt(Rr)),
                  gamma_space_gpu_seed1_vs_seed2=imageio.diff_report(np.sqrt(A), np.sqrt(B)),
                  tolerances="|mean(D2)| <= 4 sd/sqrt(N); 0.90 <= variance ratio <= 1.10; >= 99.5 % of channels within 4.5 sigma_hat; max |8x8 block mean| of D2 <= 1.6 x that of D1")
    report["pass"] = bool(abs(report["mean_D2"]) <= 4 * report["sd_D2_over_sqrtN"] and 0.9 <= report["variance_ratio_D2_over_D1"] <= 1.1
                          and report["frac_channels_within_4p5_sigma"] >= 0.995 and report["max_abs_block_mean_D2"] <= 1.6 * report["max_abs_block_mean_D1"])
    imageio.save_png(np.sqrt(A), os.path.join(out, "t3_gpu_pixel_stream_320x180_1024spp_d16.png"))
    imageio.save_png(np.sqrt(Rr), os.path.join(out, "t3_oracle_ref_serial_320x180_1024spp_d16.png"))
    imageio.save_png(np.clip(np.abs(np.sqrt(A) - np.sqrt(Rr)) * 16.0, 0, 1), os.path.join(out, "t3_abs_diff_x16.png"))
    json.dump(report, open(os.path.join(out, "t3_report.json"), "w"), indent=1, default=lambda o: list(o) if isinstance(o, tuple) else str(o))
    print(json.dumps(report, indent=1, default=str))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "t3":
        t3(sys.argv[2:])
    else:
        print(json.dumps(imageio.diff_report(load(sys.argv[1]), load(sys.argv[2])), indent=1))
