"""The ray-pool kernel (RTW_FLAG_RAY_POOL, rtw_pool.hpp) -- a second, independently scheduled implementation of the whole path, a measured
16 - 19 % LOSS on MI355X and since round 6 a BUILD OPTION (`make POOL=1`), not part of the default library.  These tests run only with
RTW_TEST_POOL=1 against such a build (tools/gpu_pool_check.sh builds one into /tmp and points RTW_HIP_LIB at it): every golden, ragged
frames, every job shape, shards and the headline frame must come out bit-identical to the default lane-loop kernel and to the oracle,
with equal segment counters.  Without RTW_TEST_POOL=1 the module checks one thing: the default library REFUSES the flag, loudly."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_CASES, all_numerics, load_golden
from test_gpu_render import gpu_render
from test_gpu_round2 import _random_spheres_case

pytestmark = pytest.mark.gpu
POOL_BUILD = os.environ.get("RTW_TEST_POOL") == "1"
needs_pool = pytest.mark.skipif(not POOL_BUILD, reason="the ray-pool kernel is a `make POOL=1` build option (tools/gpu_pool_check.sh; RTW_TEST_POOL=1)")

FLAG_CULL, FLAG_COMPACT, FLAG_VALU, FLAG_POOL = 1, 2, 4, 8


@pytest.mark.skipif(POOL_BUILD, reason="a POOL=1 build accepts the flag")
def test_default_library_refuses_the_ray_pool_flag():
    from rtw_amd import _capi
    g = load_golden("cfg1_2spheres_96x54_16spp_d4_f32")
    with pytest.raises(_capi.RtwError) as e:
        gpu_render(g, flags=FLAG_POOL)
    assert e.value.code == -7 and "POOL=1" in str(e.value)


@needs_pool
@all_numerics
@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_ray_pool_matches_golden_bit_exact(name):
    """RTW_FLAG_RAY_POOL: rays parked in LDS between the stages, every stage on full waves of one kind (Float32; a Float64
    render ignores the flag).  Same image, same counters as the golden vectors -- and the launch geometry says which kernel ran."""
    g = load_golden(name)
    img, st = gpu_render(g, flags=FLAG_POOL)
    assert np.array_equal(img, g["image"]), int((img != g["image"]).sum())
    assert st.segments == g["segments"] and st.samples == g["width"] * g["height"] * g["spp"]
    assert st.block_threads == (1024 if g["image"].dtype == np.float32 else 256)


@needs_pool
@all_numerics
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


@needs_pool
@all_numerics
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


@needs_pool
@all_numerics
def test_ray_pool_depth_zero_and_one(oracle, rtw):
    """max_depth 0: every path is over before its first scan (the draws of the camera ray are still consumed); 1: one bounce"""
    T = np.float32
    g, cam = _random_spheres_case(rtw, oracle, T, 96, 6, depth=1)
    for depth in (0, 1, 2):
        ref, ost = oracle.render(g["flat"], cam, 96, 54, 6, T=T, max_depth=depth, seed=1)
        img, st = gpu_render(g, max_depth=depth, flags=FLAG_POOL)
        assert np.array_equal(img, ref) and st.segments == ost["segments"], depth


@needs_pool
@all_numerics
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


@needs_pool
@all_numerics
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
