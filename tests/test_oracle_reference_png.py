"""The ONLY image the reference holds: /root/reference/img/scene_random_spheres_1920x1080.png -- a 961 x 541 screenshot of
scene_random_spheres through t_cam1 (unknown code version, seed and spp).  SURVEY.md section 8(c): statistical sanity only.  CPU test, build
container only (the reference is not on the GPU box): the oracle's render of the same camera at 960 x 540 (Float64, depth 16) against
region statistics of the screenshot.

What the screenshot PINS (8-bit agreement): the sky gradient (src/ray_color.jl:1-6 + gamma 2, src/vec.jl:22), the camera (src/camera.jl:18-36,
proto.jl:19: the three unit spheres sit where the oracle puts them), the mirror of the big Metal sphere (reflect, src/light.jl:6: its upper
half reflects the sky to 0.001), the Lambertian (0.4, 0.2, 0.1) sphere and the glass sphere's mean.
What it CANNOT pin: the small spheres.  Their layout is not the layout of the current src/scenes.jl:49-84 under reseed!() (other positions
and colours), and its small Metal spheres are visibly mirror-like -- today's `fuzz = random_between(0, 5)` (src/scenes.jl:70) makes four
fifths of them matte (fuzz > 1): the screenshot predates that line.  The ground, lit by colour bleeding from those spheres, agrees only
loosely.  (Measured 2026-09: sky 0.001, metal 0.001, brown 0.011, glass 0.011, ground 0.03 - 0.08, whole frame 0.015.)"""
import os

import numpy as np
import pytest

PNG = "/root/reference/img/scene_random_spheres_1920x1080.png"
PIL = pytest.importorskip("PIL.Image")
pytestmark = pytest.mark.skipif(not os.path.exists(PNG), reason="the reference checkout is not on this box (build container only)")

#: name -> ((row0, row1, col0, col1) in the 540 x 960 frame, tolerance on the per-channel mean)
REGIONS = {
    "sky, top left": ((5, 40, 20, 250), 0.01),
    "sky, top right": ((5, 40, 800, 940), 0.01),
    "sky, above the horizon": ((90, 120, 20, 280), 0.01),
    "Metal (0.7, 0.6, 0.5) fuzz 0 at (4, 1, 0): upper half, mirrors the sky": ((80, 150, 600, 700), 0.01),
    "the same sphere around its equator": ((150, 190, 560, 740), 0.02),
    "Lambertian (0.4, 0.2, 0.1) at (-4, 1, 0)": ((75, 170, 330, 365), 0.04),
    "Dielectric 1.5 at (0, 1, 0)": ((60, 110, 420, 470), 0.04),
    "whole frame": ((0, 540, 0, 960), 0.04),
    "lower half (ground + small spheres: another random layout)": ((270, 540, 0, 960), 0.08),
    "ground in front (colour bleeding from other small spheres)": ((500, 538, 300, 660), 0.15),
}


@pytest.fixture(scope="module")
def frames(oracle, rtw):
    ref = np.asarray(PIL.open(PNG).convert("RGB")).astype(np.float64) / 255.0
    assert ref.shape == (541, 961, 3)
    T = np.float64
    rtw.reseed()                                                  # src/proto/proto.jl:198-199
    scene, cam = rtw.scene_random_spheres(elem_type=T), rtw.t_cam1(elem_type=T)
    img, _ = oracle.render(rtw.flatten_scene(scene, T), cam, 960, 540, 16, T=T, max_depth=16, seed=1)
    return ref[:540, :960], np.clip(np.asarray(img, dtype=np.float64), 0.0, 1.0)      # (the screenshot's last row / column: window edge)


@pytest.mark.parametrize("name", list(REGIONS))
def test_region_means_agree_with_the_screenshot(frames, name):
    ref, img = frames
    (y0, y1, x0, x1), tol = REGIONS[name]
    a, b = ref[y0:y1, x0:x1].mean(axis=(0, 1)), img[y0:y1, x0:x1].mean(axis=(0, 1))
    assert np.abs(a - b).max() <= tol, (name, a.round(3), b.round(3))


def test_sky_gradient_row_by_row(frames):
    """rows 2 .. 110 of the left margin (row 0 of the screenshot is a window edge): (0.5, 0.7, 1.0) -> white with the reference's gamma, row by row"""
    ref, img = frames
    a, b = ref[2:110, 10:200].mean(axis=1), img[2:110, 10:200].mean(axis=1)
    assert np.abs(a - b).max() <= 0.012, float(np.abs(a - b).max())
    assert a[0, 0] < a[-1, 0] and b[0, 0] < b[-1, 0]             # red rises towards the horizon: (1 - t) white + t (0.5, 0.7, 1.0)


def test_the_screenshots_small_spheres_are_not_todays_scene(frames):
    """Documented finding, kept honest by a check: under the same camera the small-sphere band of the screenshot is NOT today's
    scene_random_spheres under reseed!() -- structurally different although its mean colour is close (same generator of colours)."""
    ref, img = frames
    band_r, band_i = ref[300:540:4, 0:960:4], img[300:540:4, 0:960:4]                 # (4 x 4 decimation: well above the oracle's 16-spp noise)
    corr = np.corrcoef(band_r.mean(axis=2).ravel(), band_i.mean(axis=2).ravel())[0, 1]
    big = np.corrcoef(ref[30:250:4, 300:760:4].mean(axis=2).ravel(), img[30:250:4, 300:760:4].mean(axis=2).ravel())[0, 1]      # (control: the three unit spheres against the sky DO correlate)
    assert big > 0.9 and corr < 0.6, (big, corr)
