#!/usr/bin/env python3
"""Soak test of the matrix-pipe filter: random scans (unit ops 13 and 14 vs the oracle) with fresh seeds for a
given number of seconds, then random small renders in every scan mode vs the oracle; the numerics mode of the ray-sphere test
(reference / contract / reference_fma2) changes with the seed, on both sides.
usage: python tools/gpu_soak.py [seconds=300] [first_seed=1000]   (needs the GPU; prints one summary line per part)
A 60-second slice of the same two loops gates the GPU suite: tests/test_gpu_round3.py::test_soak_slice."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np

NUMERICS = ("reference", "contract", "reference_fma2")      # the numerics mode of a round follows its seed: all three are soaked


def _set_numerics(seed):
    """both sides -- the oracle and the product's Python mirror -- to the mode of this seed; returns (mode, restore())"""
    import rtw_oracle as O
    from rtw_amd import _capi
    mode = NUMERICS[(seed // 2) % 3]
    prev = (O.set_numerics(mode), _capi.set_default_numerics(mode))
    return mode, (lambda: (O.set_numerics(prev[0]), _capi.set_default_numerics(prev[1])))


def scan_rounds(seconds, seed, log=print, max_rounds=None):
    """Random scenes (1 ... 1500 spheres, extents 1e-3 ... 1e6, both precisions) x 131 072 random rays through the matrix-pipe
    scan and its block-culling form vs the oracle.  -> (rounds, rays, mismatches, next seed)"""
    import rtw_oracle as O
    from test_gpu_units import run_unit
    from test_gpu_round2 import _stress_scene, _stress_rays
    t0 = time.time()
    rays_total = bad_total = rounds = 0
    while time.time() - t0 < seconds and (max_rounds is None or rounds < max_rounds):
        rng = np.random.default_rng(seed)
        T = np.float32 if seed % 2 == 0 else np.float64
        mode, restore = _set_numerics(seed)
        n = int(rng.choice([1, 2, 7, 33, 64, 65, 200, 485, 600, 1500]))
        scale = float(rng.choice([1e-3, 0.1, 1, 10, 12, 100, 1e3, 1e4, 1e6]))
        m = 131072
        flat = _stress_scene(rng, n, T, scale)
        rays = _stress_rays(rng, flat, m, T, scale)
        tmin = T(1e-4)
        ref_idx, ref_t = O.hit_world_batch(flat, rays, tmin, np.inf, T)
        x = np.concatenate([rays.astype(np.float64), np.full((m, 1), float(tmin)), np.full((m, 1), np.inf)], 1)
        for op in (13, 14):                      # the matrix-pipe scan and its block-culling form
            y = run_unit(op, x, 9, T, flat=flat)
            bad = (y[:, 0].astype(np.int64) != ref_idx) | ((ref_idx >= 0) & (y[:, 1] != ref_t.astype(np.float64)))
            rays_total += m; bad_total += int(bad.sum())
            if bad.any():
                log(f"MISMATCH op {op} seed {seed} numerics {mode} n {n} scale {scale} T {T.__name__}: {int(bad.sum())} rays, first {np.flatnonzero(bad)[:5]}")
        restore()
        rounds += 1
        seed += 1
    return rounds, rays_total, bad_total, seed


def render_rounds(seconds, seed, log=print, max_rounds=None):
    """Random small scenes / cameras rendered in all four scan modes (RTW_TEST_POOL=1, a `make POOL=1` library: also by the ray-pool kernel) vs the oracle.  -> (images, mismatches, next seed)
    `max_rounds`: stop after that many scenes (a reproducible amount of work for a fixed seed list)"""
    import rtw_oracle as O
    import rtw_amd as R
    from test_gpu_render import gpu_render
    from conftest import load_golden
    g0 = load_golden("cfg1_2spheres_96x54_16spp_d4_f32")
    t0 = time.time()
    imgs = bad_imgs = scenes = 0
    while time.time() - t0 < seconds and (max_rounds is None or scenes < max_rounds):
        rng = np.random.default_rng(seed)
        T = np.float32 if seed % 2 == 0 else np.float64
        mode, restore = _set_numerics(seed)
        n = int(rng.choice([3, 20, 100, 485]))
        scale = float(rng.choice([0.5, 1, 4, 30]))
        flat = dict(n=n, cx=(rng.uniform(-4, 4, n) * scale).astype(T), cy=(rng.uniform(-2, 2, n) * scale).astype(T),
                    cz=(rng.uniform(-9, -2, n) * scale).astype(T), r=(rng.uniform(0.05, 0.6, n) * scale * rng.choice([1, 1, 1, -1], n)).astype(T),
                    kind=rng.integers(0, 3, n).astype(np.int32), ar=rng.uniform(0, 1, n).astype(T), ag=rng.uniform(0, 1, n).astype(T),
                    ab=rng.uniform(0, 1, n).astype(T), param=np.where(rng.random(n) < 0.5, 1.5, rng.uniform(0, 1, n)).astype(T))
        flat["cx"][0], flat["cy"][0], flat["cz"][0], flat["r"][0] = 0, T(-100.5 * scale), T(-1 * scale), T(100 * scale)   # a ground sphere
        cam = R.default_camera((rng.uniform(-3, 3) * scale, rng.uniform(0.2, 3) * scale, rng.uniform(0.5, 6) * scale), (0, 0, -5 * scale), (0, 1, 0),
                               float(rng.uniform(20, 90)), 16 / 9, float(rng.choice([0.0, 0.1 * scale])), 5.0 * scale, elem_type=T)
        camd = {k: np.asarray(getattr(cam, k)) for k in ("origin", "lower_left_corner", "horizontal", "vertical", "u", "v", "w", "lens_radius")}
        g = dict(g0, flat=flat, cam=camd, image=np.zeros((1, 1, 3), T))
        ref, ost = O.render(flat, cam, 64, 36, 6, T=T, max_depth=12, seed=seed, n_chunks=3)
        for flags in (0, 4, 1, 5) + ((8,) if os.environ.get("RTW_TEST_POOL") == "1" else ()):     # matrix-pipe scan, all-VALU scan, cull, all-VALU cull (+ the ray-pool kernel of a POOL=1 build)
            img, st = gpu_render(g, width=64, height=36, spp=6, n_chunks=3, max_depth=12, seed=seed, flags=flags)
            imgs += 1
            if not (np.array_equal(img, ref, equal_nan=True) and st.segments == ost["segments"]):
                bad_imgs += 1
                log(f"IMAGE MISMATCH seed {seed} numerics {mode} n {n} scale {scale} flags {flags}: {(img != ref).sum()} channels")
        restore()
        seed += 1
        scenes += 1
    return imgs, bad_imgs, seed


if __name__ == "__main__":
    import rtw_oracle as O
    if len(sys.argv) > 1 and sys.argv[1] == "--day":      # the rolling slice that was half of tests/test_gpu_round3.py::test_soak_slice until round 5
        budget, seed = (float(sys.argv[2]) if len(sys.argv) > 2 else 40.0), 100000 + 1000 * (int(time.time()) // 86400 % 10000)
        print(f"day seed {seed}", flush=True)
    else:
        budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
        seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    O.build(); O.lib()
    t0 = time.time()
    say = lambda s: print(s, flush=True)
    rounds, rays_total, bad_total, seed = scan_rounds(budget * 0.7, seed, say)
    say(f"scan soak: {rounds} rounds, {rays_total} rays, {bad_total} mismatches, {time.time() - t0:.0f} s")
    imgs, bad_imgs, seed = render_rounds(budget - (time.time() - t0), seed, say)
    say(f"render soak: {imgs} images (all four scan modes), {bad_imgs} mismatches; total {time.time() - t0:.0f} s")
    sys.exit(1 if (bad_total or bad_imgs) else 0)
