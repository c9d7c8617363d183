"""Round-4 GPU tests (all through the C ABI):
  * the ray-pool kernel (RTW_FLAG_RAY_POOL, rtw_pool.hpp) -- a second, independently scheduled implementation of the whole path:
    every golden, ragged frames, every job shape, shards and the headline frame must come out bit-identical to the default
    lane-loop kernel and to the oracle, with equal segment counters;
  * Float64 at full scale: configs[4]'s geometry (3840x2160, depth 50) in all three scan modes, and against the live oracle;
  * the soak slice on a FIXED seed list (reproducible) next to the rolling day seed."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN_CASES, ROOT, load_golden
from test_gpu_render import gpu_render
from test_gpu_round2 import _random_spheres_case

pytestmark = pytest.mark.gpu

FLAG_CULL, FLAG_COMPACT, FLAG_VALU, FLAG_POOL = 1, 2, 4, 8


# ---- the ray-pool kernel -------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_ray_pool_matches_golden_bit_exact(name):
    """RTW_FLAG_RAY_POOL: rays parked in LDS between the stages, every stage on full waves of one kind (Float32; a Float64
    render ignores the flag).  Same image, same counters as the golden vectors -- and the launch geometry says which kernel ran."""
    g = load_golden(name)
    img, st = gpu_render(g, flags=FLAG_POOL)
    assert np.array_equal(img, g["image"]), int((img != g["image"]).sum())
    assert st.segments == g["segments"] and st.samples == g["width"] * g["height"] * g["spp"]
    assert st.block_threads == (1024 if g["image"].dtype == np.float32 else 256)


@pytest.mark.parametrize("W,H", [(8, 2048), (24, 1000), (2048, 8), (72, 9), (1, 1), (9, 7)])
def test_ray_pool_ragged_frames(oracle, rtw, W, H):
    """frames that are not whole tiles, one-pixel frames, frames with fewer items than the pool has slots"""
    T = np.float32
    g, cam = _random_spheres_case(rtw, oracle, T, 64, 5, depth=12)
    g = dict(g, width=W, height=H)
    ref, ost = oracle.render(g["flat"], cam, W, H, 5, T=T, max_depth=12, seed=1)
    img, st = gpu_render(g, width=W, height=H, flags=FLAG_POOL)
    assert st.block_threads == 1024
    assert np.array_equal(img, ref) and st.segments == ost["segments"]


@pytest.mark.parametrize("job_pixels", [1, 4, 8, 16])
@pytest.mark.parametrize("spp,n_chunks", [(3, 0), (40, 0), (97, 0), (64, 1), (200, 200)])
def test_ray_pool_job_shapes_and_chunkings(job_pixels, spp, n_chunks):
    """the pool's item dispenser over every job size and over chunkings that leave padding items, one chunk per pixel, one sample
    per chunk: pool == lane loop, bit for bit (the lane loop is pinned on the oracle for the same parameters elsewhere)"""
    g = load_golden("cfg2_random_320x180_64spp_d16_f32")
    g = dict(g, width=160, height=90)
    a, sa = gpu_render(g, width=160, height=90, spp=spp, n_chunks=n_chunks, job_pixels=job_pixels)
    b, sb = gpu_render(g, width=160, height=90, spp=spp, n_chunks=n_chunks, job_pixels=job_pixels, flags=FLAG_POOL)
    assert sb.block_threads == 1024 and sa.block_threads == 256
    assert np.array_equal(a, b) and sa.segments == sb.segments and sa.samples == sb.samples == 160 * 90 * spp


def test_ray_pool_depth_zero_and_one(oracle, rtw):
    """max_depth 0: every path is over before its first scan (the draws of the camera ray are still consumed); 1: one bounce"""
    T = np.float32
    g, cam = _random_spheres_case(rtw, oracle, T, 96, 6, depth=1)
    for depth in (0, 1, 2):
        ref, ost = oracle.render(g["flat"], cam, 96, 54, 6, T=T, max_depth=depth, seed=1)
        img, st = gpu_render(g, max_depth=depth, flags=FLAG_POOL)
        assert np.array_equal(img, ref) and st.segments == ost["segments"], depth


def test_ray_pool_shards_and_compact_tiles(rtw):
    """3 shards, full-frame and compact: the pool kernel's shards sum / scatter to the unsharded lane-loop frame"""
    import torch
    T = np.float32
    rtw.reseed()
    dr = rtw.DeviceRenderer(rtw.scene_random_spheres(elem_type=T), rtw.t_cam1(elem_type=T), device=0)
    W, H, spp = 200, 112, 24
    s = torch.cuda.current_stream()
    full = torch.empty(H * W * 3, dtype=torch.float32, device="cuda:0")
    dr.render_into(full.data_ptr(), W, spp, depth=16, seed=3, stream=s.cuda_stream)
    acc = torch.zeros_like(full)
    for r in range(3):
        part = torch.empty_like(full)
        dr.render_into(part.data_ptr(), W, spp, depth=16, seed=3, stream=s.cuda_stream, shard_index=r, shard_count=3, ray_pool=True)
        assert dr.stats()["block_threads"] == 1024
        acc += part
    assert bool(torch.equal(acc, full))
    frame = torch.zeros(H * W, 3, dtype=torch.float32, device="cuda:0")
    for r in range(3):
        ne = rtw.compact_elems(W, r, 3)
        comp = torch.empty(ne, dtype=torch.float32, device="cuda:0")
        dr.render_into(comp.data_ptr(), W, spp, depth=16, seed=3, stream=s.cuda_stream, shard_index=r, shard_count=3, compact=True, ray_pool=True, n_elems=ne)
        dest = rtw.compact_to_frame_index(W, r, 3)
        src = np.flatnonzero(dest >= 0)
        frame.index_copy_(0, torch.from_numpy(dest[src]).to("cuda:0"), comp.reshape(-1, 3).index_select(0, torch.from_numpy(src).to("cuda:0")))
    assert bool(torch.equal(frame.reshape(-1), full))
    dr.close()


def test_ray_pool_identical_at_1080p(rtw):
    """1920x1080 x 100 spp, depth 50 (8.2e8 segments): pool kernel == lane-loop kernel, image and counters"""
    import torch
    T = np.float32
    rtw.reseed()
    dr = rtw.DeviceRenderer(rtw.scene_random_spheres(elem_type=T), rtw.t_cam1(elem_type=T), device=0)
    a = torch.empty(1080 * 1920 * 3, dtype=torch.float32, device="cuda:0")
    b = torch.empty_like(a)
    s = torch.cuda.current_stream()
    dr.render_into(a.data_ptr(), 1920, 100, depth=50, seed=1, stream=s.cuda_stream)
    sa = dr.stats()
    dr.render_into(b.data_ptr(), 1920, 100, depth=50, seed=1, stream=s.cuda_stream, ray_pool=True)
    sb = dr.stats()
    assert sa["block_threads"] == 256 and sb["block_threads"] == 1024
    assert sa["segments"] == sb["segments"] and sa["samples"] == sb["samples"] == 1920 * 1080 * 100
    assert bool(torch.equal(a, b)), int((a != b).sum())
    dr.close()


# ---- Float64 at full scale --------------------------------------------------------------------------------------------------
def test_three_scan_modes_identical_f64_4k(rtw):
    """BASELINE configs[4]'s geometry -- 3840x2160, Float64, depth 50 -- at 100 spp (3.3e9 ray segments) in all three scan modes.  The
    Float64 kernel feeds the SAME binary32 / f16 matrix-pipe filter with inputs ROUNDED from binary64 (an extra 1.5 S term of its error
    budget, rtw_device.hpp): a candidate lost to that rounding shows against the all-VALU leg (its own conservative binary32 filter, a
    different derivation) and the cull leg; all three share the exact binary64 contract test of the candidates."""
    import torch
    T = np.float64
    rtw.reseed()
    dr = rtw.DeviceRenderer(rtw.scene_random_spheres(elem_type=T), rtw.t_cam1(elem_type=T), device=0)
    a = torch.empty(2160 * 3840 * 3, dtype=torch.float64, device="cuda:0")
    b, c = torch.empty_like(a), torch.empty_like(a)
    s = torch.cuda.current_stream()
    dr.render_into(a.data_ptr(), 3840, 100, depth=50, seed=1, stream=s.cuda_stream)
    sa = dr.stats()
    dr.render_into(b.data_ptr(), 3840, 100, depth=50, seed=1, stream=s.cuda_stream, scan_valu=True)
    sb = dr.stats()
    dr.render_into(c.data_ptr(), 3840, 100, depth=50, seed=1, stream=s.cuda_stream, group_cull=True)
    sc = dr.stats()
    assert sa["segments"] == sb["segments"] == sc["segments"] and sa["samples"] == 3840 * 2160 * 100
    assert bool(torch.equal(a, b)), int((a != b).sum())
    assert bool(torch.equal(a, c)), int((a != c).sum())
    dr.close()


def test_f64_4k_8spp_against_the_live_oracle(oracle, rtw):
    """3840x2160 x 8 spp, depth 50, Float64 (6.6e7 samples, 1.8e8 segments) against the oracle rendered here: bit-exact image and
    segment count, plain and cull mode (round 2 compared this geometry at 1 spp)."""
    T = np.float64
    g, cam = _random_spheres_case(rtw, oracle, T, 3840, 8, depth=50)
    ref, ost = oracle.render(g["flat"], cam, 3840, 2160, 8, T=T, max_depth=50, seed=1)
    for flags in (0, FLAG_CULL):
        img, st = gpu_render(g, flags=flags)
        assert st.segments == ost["segments"]
        assert np.array_equal(img, ref), int((img != ref).sum())


def test_f64_published_configuration_modes_identical(rtw):
    """the reference's published configuration (Float64, 1920x1080, depth 16; README.md:86) at 200 spp: three scan modes, one image"""
    import torch
    T = np.float64
    rtw.reseed()
    dr = rtw.DeviceRenderer(rtw.scene_random_spheres(elem_type=T), rtw.t_cam1(elem_type=T), device=0)
    a = torch.empty(1080 * 1920 * 3, dtype=torch.float64, device="cuda:0")
    b, c = torch.empty_like(a), torch.empty_like(a)
    s = torch.cuda.current_stream()
    dr.render_into(a.data_ptr(), 1920, 200, depth=16, seed=1, stream=s.cuda_stream)
    sa = dr.stats()
    dr.render_into(b.data_ptr(), 1920, 200, depth=16, seed=1, stream=s.cuda_stream, scan_valu=True)
    sb = dr.stats()
    dr.render_into(c.data_ptr(), 1920, 200, depth=16, seed=1, stream=s.cuda_stream, group_cull=True)
    sc = dr.stats()
    assert sa["segments"] == sb["segments"] == sc["segments"]
    assert bool(torch.equal(a, b)) and bool(torch.equal(a, c))
    dr.close()
