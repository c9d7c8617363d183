"""Round-2 GPU parity cases: the Float64 path at BASELINE configs[4] scale, the in-library
multi-device render (rtw_params.n_devices / device_ids), re-entrancy, the compact tile layout,
the LDS / group-cull scan instantiations as unit ops, a randomized scan stress, the exact
pixel accumulation, the remaining reference scenes, and the statistical tier T3 against the
oracle's REF_SERIAL mode (the closest stand-in for the reference's own render loop).

Everything goes through the C ABI (ctypes).  Bit-exact unless a tolerance is written out.
"""
import ctypes as C
import threading

import numpy as np
import pytest

from conftest import CamObj, all_numerics, load_golden
from test_gpu_render import gpu_render
from test_gpu_units import run_unit

pytestmark = pytest.mark.gpu


def _cam_dict(cam, oracle):
    return {k: getattr(cam, k) for k in oracle.CAM_FIELDS + ("lens_radius",)}


def _random_spheres_case(rtw, oracle, T, width, spp, depth=50, n_chunks=0):
    rtw.reseed()
    flat = rtw.flatten_scene(rtw.scene_random_spheres(elem_type=T), T)
    cam = rtw.t_cam1(elem_type=T)
    h = rtw.image_height(width)
    return dict(flat=flat, cam=_cam_dict(cam, oracle), image=np.zeros(1, T), width=width, height=h, spp=spp,
                depth=depth, seed=1, n_chunks=n_chunks or oracle.default_n_chunks(spp)), cam


# ---- Float64 at BASELINE configs[4] geometry -------------------------------------------------------
@all_numerics
def test_full_size_f64_4k_properties(oracle, rtw):
    """3840x2160, scene_random_spheres, depth 50, Float64 (configs[4], one GPU's view) at a sample
    count the oracle finishes in seconds: bit-exact vs the live oracle, exact segment count,
    shard-sum invariance, group-cull identity."""
    T = np.float64
    g, cam = _random_spheres_case(rtw, oracle, T, 3840, 1)
    img, st = gpu_render(g)
    ref, ost = oracle.render(g["flat"], cam, 3840, 2160, 1, T=T, max_depth=50, seed=1, n_chunks=1)
    assert img.shape == (2160, 3840, 3) and img.dtype == T
    assert np.array_equal(img, ref) and st.segments == ost["segments"]
    assert st.samples == 3840 * 2160
    a, _ = gpu_render(g, shard_index=0, shard_count=3)
    b, _ = gpu_render(g, shard_index=1, shard_count=3)
    c, _ = gpu_render(g, shard_index=2, shard_count=3)
    assert np.array_equal(a + b + c, img)
    fast, st1 = gpu_render(g, flags=1)
    assert np.array_equal(fast, img) and st1.segments == st.segments
    assert img[:400].mean() > 0.7 and 0.2 < img[1800:].mean() < 0.8


@all_numerics
def test_f64_more_chunks_than_a_batch(oracle, rtw):
    """Float64, 37 spp in 37 chunks (10 batches per job, the last one ragged), odd image size"""
    T = np.float64
    g, cam = _random_spheres_case(rtw, oracle, T, 100, 37, depth=16, n_chunks=37)
    img, st = gpu_render(g)
    ref, ost = oracle.render(g["flat"], cam, 100, g["height"], 37, T=T, max_depth=16, seed=1, n_chunks=37)
    assert np.array_equal(img, ref) and st.segments == ost["segments"]


# ---- multi-device behind the C ABI (n_devices / device_ids; Julia `devices=:all`) -------
@all_numerics
@pytest.mark.parametrize("T", [np.float32, np.float64])
def test_multi_device_render_is_identical(rtw, T):
    import torch
    scene = rtw.scene_4_spheres(elem_type=T)
    cam = rtw.t_default_cam(elem_type=T)
    one = rtw.render(scene, cam, 200, 6, depth=8)
    n_dev = torch.cuda.device_count()
    every = rtw.render(scene, cam, 200, 6, depth=8, devices="all")            # every visible device
    assert np.array_equal(one, every)
    st = rtw.last_stats()
    assert st["samples"] == 200 * 112 * 6
    # an explicit list; an ordinal may repeat: the three shards then run concurrently on device 0,
    # each on its own host thread, stream and scene copy (this is also a re-entrancy test)
    ids = [0, 0, 0] if n_dev == 1 else [d % n_dev for d in range(max(3, n_dev))]
    listed = rtw.render(scene, cam, 200, 6, depth=8, devices=ids)
    assert np.array_equal(one, listed)
    assert rtw.last_stats()["samples"] == 200 * 112 * 6 and rtw.last_stats()["segments"] == st["segments"]
    assert np.array_equal(one, rtw.render(scene, cam, 200, 6, depth=8, devices=[0], group_cull=True))


def test_multi_device_errors(rtw):
    from rtw_amd._capi import RtwError
    scene = rtw.scene_2_spheres(elem_type=np.float32)
    cam = rtw.t_default_cam()
    with pytest.raises(RtwError, match="out of range|device"):
        rtw.render(scene, cam, 96, 1, devices=[0, 4096])
    with pytest.raises(ValueError):
        rtw.render(scene, cam, 96, 1, devices="some")


@all_numerics
def test_reentrant_renders_from_two_threads(rtw, oracle):
    """Two host threads render different scenes on the same device at the same time (each call owns
    its counters; there is no shared device workspace): both images and both stats are right."""
    T = np.float32
    jobs = [(rtw.scene_4_spheres(elem_type=T), rtw.t_default_cam(elem_type=T), 160, 12),
            (rtw.scene_diel_spheres(-0.5, elem_type=T), rtw.t_cam2(elem_type=T), 128, 9)]
    expect = []
    for sc, cam, w, spp in jobs:
        ref, ost = oracle.render(rtw.flatten_scene(sc, T), cam, w, rtw.image_height(w), spp, T=T, max_depth=16, seed=1)
        expect.append((ref, ost["segments"]))
    for _ in range(3):
        out = [None, None]

        def work(k):
            sc, cam, w, spp = jobs[k]
            img = rtw.render(sc, cam, w, spp, depth=16, device=0)
            out[k] = (img, rtw.last_stats()["segments"])         # rtw_stats is per calling thread
        th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
        [t.start() for t in th]
        [t.join() for t in th]
        for k in range(2):
            assert np.array_equal(out[k][0], expect[k][0]) and out[k][1] == expect[k][1], k


@all_numerics
def test_two_streams_in_flight_and_compact_layout(rtw):
    """device-resident path: two renders enqueued back to back on two streams without any host wait;
    the compact tile-major shard layout holds exactly the owned tiles of the full frame"""
    import torch
    T = np.float32
    g = load_golden("cfg2_random_320x180_64spp_d16_f32")
    rtw.reseed()
    dr = rtw.DeviceRenderer(rtw.scene_random_spheres(elem_type=T), rtw.t_cam1(elem_type=T), device=0)
    W, H = 320, 180
    full = torch.empty(H * W * 3, dtype=torch.float32, device="cuda:0")
    tiles_i, tiles_j = (H + 7) // 8, (W + 7) // 8
    count, index = 3, 1
    n_local = (tiles_i * tiles_j - index + count - 1) // count
    comp = torch.full((n_local * 64 * 3,), -7.0, dtype=torch.float32, device="cuda:0")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())   # the fill of `comp` runs on the default stream
    dr.render_into(full.data_ptr(), W, 64, depth=16, seed=1, n_chunks=g["n_chunks"], stream=s1.cuda_stream)
    dr.render_into(comp.data_ptr(), W, 64, depth=16, seed=1, n_chunks=g["n_chunks"], shard_index=index, shard_count=count,
                   stream=s2.cuda_stream, compact=True)
    st2 = dr.stats()                                           # the second call's counters
    s1.synchronize(); s2.synchronize()
    img = full.cpu().numpy().reshape(W, H, 3).transpose(1, 0, 2)
    assert np.array_equal(img, g["image"])
    c = comp.cpu().numpy().reshape(n_local, 8, 8, 3)           # [tile k][j mod 8][i mod 8][channel]
    mask = rtw.owned_pixel_mask(W, index, count)
    seen = 0
    for k in range(n_local):
        t = k * count + index
        tj, ti = divmod(t, tiles_i)
        for jj in range(8):
            for ii in range(8):
                i0, j0 = ti * 8 + ii, tj * 8 + jj
                if i0 < H and j0 < W:
                    assert mask[i0, j0] and np.array_equal(c[k, jj, ii], img[i0, j0]), (k, ii, jj)
                    seen += 1
                else:
                    assert np.all(c[k, jj, ii] == -7.0)       # outside the image: never written
    assert seen == int(mask.sum()) and st2["samples"] == seen * 64
    dr.close()


@all_numerics
def test_gather_mode_assembles_the_frame(rtw):
    """shard.render_sharded(mode="gather") on one rank: compact render + indexed copy = the golden frame"""
    import torch
    T = np.float32
    g = load_golden("metal4_96x54_8spp_d16_f32")
    dr = rtw.DeviceRenderer(rtw.scene_4_spheres(elem_type=T), rtw.t_default_cam(elem_type=T), device=0)
    buf = torch.empty(96 * 54 * 3 + 192 * 8, dtype=torch.float32, device="cuda:0")     # larger than needed is fine

    def shard(idx, cnt):
        dr.render_into(buf.data_ptr(), 96, g["spp"], depth=g["depth"], seed=g["seed"], n_chunks=g["n_chunks"], shard_index=idx,
                       shard_count=cnt, stream=torch.cuda.current_stream().cuda_stream, compact=True)
        return buf
    frame = rtw.render_sharded(shard, 96, mode="gather")
    assert np.array_equal(frame.cpu().numpy().reshape(96, 54, 3).transpose(1, 0, 2), g["image"])
    dr.close()


def test_c_host_example_renders_the_same_image(rtw, tmp_path):
    """examples/render_c.c (plain C99 over include/rtw_hip.h, n_devices = -1) against the Python mirror"""
    import os
    import subprocess
    from conftest import ROOT
    from rtw_amd import _capi
    lib_dir = os.path.dirname(_capi.LIB_PATH)
    exe = str(tmp_path / "render_c")
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "render_c.c"),
                           "-L", lib_dir, "-lrtw_hip", f"-Wl,-rpath,{lib_dir}", "-lm", "-o", exe])
    r = subprocess.run([exe, "96", "8"], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0 and "samples" in r.stderr, r.stderr
    got = rtw.imageio.load_ppm(str(tmp_path / "render_c.ppm"))
    T = np.float32
    img = rtw.render(rtw.scene_2_spheres(elem_type=T), rtw.t_default_cam(elem_type=T), 96, 8, depth=16, seed=1)
    assert np.array_equal(got, rtw.imageio.to_u8(img))


def test_stats_after_shutdown_reports_no_render(rtw):
    from rtw_amd import _capi
    L = _capi.lib()
    rtw.render(rtw.scene_2_spheres(elem_type=np.float32), rtw.t_default_cam(), 96, 1)
    assert L.rtw_shutdown() == 0
    st = _capi.Stats()
    assert L.rtw_stats(C.byref(st)) == -6 and b"no render" in L.rtw_last_error()
    img = rtw.render(rtw.scene_2_spheres(elem_type=np.float32), rtw.t_default_cam(), 96, 1)   # and it still works
    assert img.shape == (54, 96, 3) and rtw.last_stats()["samples"] == 96 * 54


def test_non_finite_scene_is_rejected(rtw):
    from rtw_amd._capi import RtwError
    scene = rtw.scene_2_spheres(elem_type=np.float32)
    scene[0].center = np.array([np.nan, 0, -1], np.float32)
    with pytest.raises(RtwError, match="not finite"):
        rtw.render(scene, rtw.t_default_cam(), 96, 1)


# ---- exact pixel accumulation ---------------------------------------------------------------------
def test_exact_accumulation_unit(oracle):
    """op 12: fx_from_double -> 128-bit adds -> fx_to_double, vs the oracle's __int128 version
    (itself checked against rational arithmetic in tests/test_oracle_kats.py)"""
    rng = np.random.default_rng(5)
    n = 4096
    x = rng.uniform(0, 4, (n, 8)) * rng.choice([1, 1, 1, 1e-3, 1e-9, 2.0 ** -40, -1, 1e6], (n, 8))
    x[0] = [0.1] * 8
    x[1] = [2.0 ** 31, 1, 1, 1, 1, 1, 1, 1]                     # poison: magnitude >= 2^31
    x[2] = [np.nan, 1, 1, 1, 1, 1, 1, 1]
    x[3] = [np.inf, 1, 1, 1, 1, 1, 1, 1]
    x[4] = [2.0 ** 31 - 1, 2.0 ** 31 - 1, 0.5, 2.0 ** -64, 2.0 ** -65, -2.0 ** -64, 0, 0]
    x[5] = [1.0, 2.0 ** -53, 2.0 ** -54, 0, 0, 0, 0, 0]        # a tie and a sticky bit
    x[6] = [-0.3, 0.3, 1e-20, -1e-20, 0, 0, 0, 0]
    y = run_unit(12, x, 2, np.float64)
    for i in range(n):
        s, bad = oracle.fx_sum(x[i])
        assert y[i, 1] == bad, i
        assert (np.isnan(y[i, 0]) and np.isnan(s)) or y[i, 0] == s, (i, y[i, 0], s)


@all_numerics
@pytest.mark.parametrize("name", ["cfg2_random_320x180_64spp_d16_f32", "random_64x36_8spp_d50_f64", "metal4_96x54_8spp_d16_f32"])
def test_job_size_does_not_change_the_image(name):
    """rtw_params.job_pixels (1, 4, 8 or 16 pixels per work-queue job: column strips of 1x1, 4x1, 8x1, 8x2 rows x columns;
    0 = automatic) is scheduling granularity only: same image, same segment count, also for a shard and in group-cull mode"""
    g = load_golden(name)
    for jp in (1, 4, 8, 16):
        img, st = gpu_render(g, job_pixels=jp)
        assert np.array_equal(img, g["image"]) and st.segments == g["segments"], jp
    a, _ = gpu_render(g, job_pixels=1, shard_index=1, shard_count=5, flags=1)
    b, _ = gpu_render(g, job_pixels=16, shard_index=1, shard_count=5)
    assert np.array_equal(a, b)
    # ragged size, fewer chunks than a one-pixel job's batch holds
    w = 50
    x, _ = gpu_render(g, width=w, height=(w * 9) // 16, spp=5, n_chunks=5, job_pixels=1)
    y, _ = gpu_render(g, width=w, height=(w * 9) // 16, spp=5, n_chunks=5, job_pixels=16)
    z, _ = gpu_render(g, width=w, height=(w * 9) // 16, spp=5, n_chunks=5, job_pixels=8)
    assert np.array_equal(x, y) and np.array_equal(x, z)


@all_numerics
def test_image_is_invariant_under_chunk_order_and_slots(oracle, rtw):
    """the pixel sum is an exact integer sum: sharding, group-cull and the LDS job-slot schedule
    cannot change it; and a pixel that receives a huge radiance is NaN, as documented"""
    T = np.float32
    g = load_golden("metal4_96x54_8spp_d16_f32")
    a, _ = gpu_render(g, spp=300, n_chunks=0)
    b, _ = gpu_render(g, spp=300, n_chunks=0, flags=1)
    assert np.array_equal(a, b)
    flat = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in g["flat"].items()}
    flat["ar"][:] = 3e9; flat["ag"][:] = 3e9; flat["ab"][:] = 3e9     # "albedo" 3e9: radiance >= 2^31 after one bounce
    img, _ = gpu_render(dict(g, flat=flat), spp=4, n_chunks=4)
    ref, _ = oracle.render(flat, g["cam"], g["width"], g["height"], 4, T=T, max_depth=g["depth"], seed=g["seed"], n_chunks=4)
    assert np.isnan(ref).any() and np.array_equal(np.isnan(img), np.isnan(ref))
    assert np.array_equal(img[~np.isnan(ref)], ref[~np.isnan(ref)])


# ---- the scan instantiations the trace kernel runs, as unit ops ------------------------------------
def _stress_scene(rng, n, T, scale):
    cx, cy, cz = [rng.uniform(-scale, scale, n) for _ in range(3)]
    r = 10.0 ** rng.uniform(-3, np.log10(max(scale, 1.0)) - 0.3, n) * rng.choice([1, 1, 1, -1], n)
    if n >= 4:
        cx[1], cy[1], cz[1], r[1] = cx[0], cy[0], cz[0], r[0]         # coincident spheres: exact ties
        r[2] = 1000.0 * scale / 10                                      # one huge sphere (the "ground")
    flat = dict(n=n, cx=cx.astype(T), cy=cy.astype(T), cz=cz.astype(T), r=r.astype(T),
                kind=rng.integers(0, 3, n).astype(np.int32), ar=np.ones(n, T), ag=np.ones(n, T), ab=np.ones(n, T),
                param=np.full(n, 1.5, T))
    return flat


def _stress_rays(rng, flat, m, T, scale):
    n = flat["n"]
    o = rng.uniform(-scale, scale, (m, 3))
    far = rng.random(m) < 0.2
    o[far] *= 50.0                                                      # far origins
    d = rng.normal(size=(m, 3))
    if n > 0:
        # a third of the rays start ON a sphere (the self-intersection regime after a bounce), a third aim at a centre
        k = rng.integers(0, n, m)
        c = np.stack([flat["cx"][k], flat["cy"][k], flat["cz"][k]], 1).astype(np.float64)
        u = rng.normal(size=(m, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
        on = rng.random(m) < 0.33
        o[on] = c[on] + u[on] * np.abs(flat["r"][k][on, None].astype(np.float64))
        aim = (~on) & (rng.random(m) < 0.5)
        d[aim] = c[aim] - o[aim] + rng.normal(size=(int(aim.sum()), 3)) * np.abs(flat["r"][k][aim, None]) * 0.7
    o = o.astype(T)
    d = d.astype(T)
    nrm = np.sqrt((d.astype(T) ** 2).sum(1, dtype=T)).astype(T)
    nrm[nrm == 0] = 1
    d = (d / nrm[:, None]).astype(T)
    if n > 0:
        # grazing rays: the line passes at distance r (1 + eps) from a centre, eps from -1e-3 to +1e-3 through 0 in
        # both precisions' rounding range -- the Float64 scan's binary32 pass-1 filter must never lose a candidate
        tang = rng.random(m) < 0.2
        nt = int(tang.sum())
        k2 = rng.integers(0, n, nt)
        c2 = np.stack([flat["cx"][k2], flat["cy"][k2], flat["cz"][k2]], 1).astype(np.float64)
        dd = rng.normal(size=(nt, 3)); dd /= np.linalg.norm(dd, axis=1, keepdims=True)
        perp = np.cross(dd, rng.normal(size=(nt, 3))); perp /= np.linalg.norm(perp, axis=1, keepdims=True)
        eps = rng.choice([0.0, 1e-15, -1e-15, 1e-12, -1e-12, 1e-9, -1e-9, 1e-7, -1e-7, 1e-5, -1e-5, 1e-3, -1e-3], nt)
        rr = np.abs(flat["r"][k2].astype(np.float64))
        back = rng.uniform(0.5, 40.0, nt) * np.maximum(rr, 1.0)
        o[tang] = (c2 + perp * (rr * (1.0 + eps))[:, None] - dd * back[:, None]).astype(T)
        d[tang] = dd.astype(T)
        nrm = np.sqrt((d.astype(T) ** 2).sum(1, dtype=T)).astype(T)
        nrm[nrm == 0] = 1
        d = (d / nrm[:, None]).astype(T)
    # the reference does not renormalise dielectric reflections (src/material.jl:48): directions drift away from
    # unit length along internal-reflection chains; the scans must agree with the oracle for those rays as well
    off = rng.random(m) < 0.15
    d[off] = (d[off] * rng.choice([1 + 3e-7, 1 - 3e-7, 1 + 2e-6, 1.0009, 0.97, 1.0097, 1.08, 3.0, 20.8], (int(off.sum()), 1))).astype(T)
    return np.concatenate([o, d], 1)


@all_numerics
@pytest.mark.parametrize("T", [np.float32, np.float64])
def test_scan_stress_plain_lds_and_cull_agree_with_oracle(oracle, T):
    """~10^6 random rays over random scenes (coordinates up to 1e4, |r| from 1e-3 to 1e3, negative
    radii, coincident spheres, origins on surfaces / far away): hit_world from global memory
    (op 8), from LDS (op 10), hit_world_cull (op 11), hit_world_mfma (op 13: pass 1 on the matrix pipe, the
    trace kernel's plain scan) and its block-culling form (op 14: RTW_FLAG_GROUP_CULL) must return the oracle's sphere
    index and t exactly."""
    rng = np.random.default_rng(2024)
    total = 0
    for case, (n, scale, m) in enumerate([(0, 1, 256), (1, 1, 4096), (2, 10, 4096), (7, 10, 65536), (33, 1, 65536),
                                          (200, 10, 262144), (485, 12, 262144), (600, 1e4, 262144), (100, 1e3, 131072)]):
        flat = _stress_scene(rng, n, T, scale)
        rays = _stress_rays(rng, flat, m, T, scale)
        tmin = T(1e-4)
        ref_idx, ref_t = oracle.hit_world_batch(flat, rays, tmin, np.inf, T)
        x = np.concatenate([rays.astype(np.float64), np.full((m, 1), float(tmin)), np.full((m, 1), np.inf)], 1)
        for op in (8, 10, 11, 13, 14):
            if op in (13, 14) and n == 0:
                continue                                   # (no matrix-pipe operands for an empty scene)
            y = run_unit(op, x, 9, T, flat=flat)
            bad = (y[:, 0].astype(np.int64) != ref_idx) | ((ref_idx >= 0) & (y[:, 1] != ref_t.astype(np.float64)))
            assert not bad.any(), (case, op, int(bad.sum()), np.flatnonzero(bad)[:5])
        total += m
        assert n == 0 or (ref_idx >= 0).mean() > 0.02
    assert total > 1_000_000


# (the full-scale comparison of the scan modes -- matrix-pipe filter, all-VALU scan, group cull -- is
#  tests/test_gpu_round3.py::test_three_scan_modes_identical_at_headline_scale)


# ---- tier T3: statistical parity with the reference's own sampling order ---------------------------
@all_numerics
def test_t3_statistical_parity_with_ref_serial(oracle, rtw):
    """GPU (PIXEL_STREAM streams) vs the oracle in REF_SERIAL mode -- one Xoroshiro128+ per Julia
    thread, static row blocks, serial consumption, reference product order: exactly what
    src/render.jl:19-23 does with `julia -t 180` -- on the headline scene at 320x180, 1024 spp,
    depth 16 (the reference's depth), linear radiance (gamma off).

    Both are unbiased estimators of the same image, so with A = GPU seed 1, B = GPU seed 2 and
    R = REF_SERIAL:  D1 = A - B and D2 = A - R have the same distribution per channel (variance
    2 v / spp).  Tolerances (stated, per channel on linear radiance):
      * |mean(D2)| <= 4 * sd(D2) / sqrt(N)                       (no global bias)
      * 0.90 <= mean(D2^2) / mean(D1^2) <= 1.10                  (same noise level)
      * >= 99.5 % of channels: |D2| <= 4.5 * sigma_hat, sigma_hat^2 = local mean of D1^2 (8x8 blocks)
      * no structured residual: the largest |8x8 block mean| of D2 is <= 1.6 x that of D1
    """
    T = np.float32
    W, H, spp, depth = 320, 180, 1024, 16          # (artefacts of the same comparison: profiles/r02_t3_*.png / .json)
    g, cam = _random_spheres_case(rtw, oracle, T, W, spp, depth=depth)
    A, _ = gpu_render(g, gamma=0)
    B, _ = gpu_render(g, gamma=0, seed=2)
    Rr, _ = oracle.render(g["flat"], cam, W, H, spp, T=T, max_depth=depth, rng_mode=oracle.REF_SERIAL, ref_threads=H,
                          product_order=oracle.PRODUCT_REFERENCE, gamma=False)
    A, B, Rr = A.astype(np.float64), B.astype(np.float64), Rr.astype(np.float64)
    D1, D2 = A - B, A - Rr
    N = D2.size
    assert abs(D2.mean()) <= 4 * D2.std() / np.sqrt(N), (D2.mean(), D2.std())
    ratio = (D2 ** 2).mean() / (D1 ** 2).mean()
    assert 0.90 <= ratio <= 1.10, ratio

    def blocks(x, f):
        hh, ww = (H // 8) * 8, (W // 8) * 8
        return f(x[:hh, :ww].reshape(hh // 8, 8, ww // 8, 8, 3), axis=(1, 3))
    var_hat = np.repeat(np.repeat(blocks(D1 ** 2, np.mean), 8, 0), 8, 1)             # per 8x8 block and channel
    hh, ww = var_hat.shape[:2]
    z = np.abs(D2[:hh, :ww]) / np.sqrt(np.maximum(var_hat, 1e-12))
    assert (z <= 4.5).mean() >= 0.995, (z <= 4.5).mean()
    m1, m2 = np.abs(blocks(D1, np.mean)).max(), np.abs(blocks(D2, np.mean)).max()
    assert m2 <= 1.6 * m1, (m1, m2)
    # and the gamma-space images agree to the Monte-Carlo noise level: mean abs diff below 1.5 %
    assert np.abs(np.sqrt(A) - np.sqrt(Rr)).mean() < 0.015
