"""Tier T1 on the GPU: whole images through the C ABI (rtw_render_f32/_f64 and the
device-resident variant) against the committed golden vectors and the live oracle.

Tolerance: NONE.  The device and the oracle share one arithmetic (IEEE ops, one
rounding per written operation, the ray-sphere discriminant in the selected numerics mode -- the
reference's own un-fused order by default --, binary64 colour math), the
same per-(pixel, chunk) Xoroshiro128+ streams and the same exact (order-free) accumulation, so every
stored channel must be bit-identical (np.array_equal), for Float32 and Float64.  The segment
counter must match the oracle's exactly as well (every path took the same branches).
"""
import numpy as np
import pytest

from conftest import GOLDEN_CASES, CamObj, load_golden

# every test once per numerics mode of the ray-sphere test (conftest.numerics): oracle, goldens and device switch together
pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("numerics")]


def gpu_render(g, **over):
    import ctypes as C
    from rtw_amd import _capi
    T = g["image"].dtype.type
    L = _capi.lib()
    S, keep = _capi.make_scene(g["flat"], T)
    Cm = _capi.make_camera(CamObj(g["cam"]), T)
    kw = dict(width=g["width"], height=g["height"], spp=g["spp"], max_depth=g["depth"], seed=g["seed"],
              n_chunks=g["n_chunks"], flags=0)
    kw.update(over)
    P = _capi.make_params(**kw)
    out = np.empty(kw["width"] * kw["height"] * 3, T)
    fn = L.rtw_render_f64 if T is np.float64 else L.rtw_render_f32
    _capi.check(fn(C.byref(S), C.byref(Cm), C.byref(P), out.ctypes.data_as(C.c_void_p)))
    st = _capi.Stats()
    _capi.check(L.rtw_stats(C.byref(st)))
    return out.reshape(kw["width"], kw["height"], 3).transpose(1, 0, 2), st


@pytest.mark.parametrize("scan", ["matrix", "valu"])
@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_image_matches_golden_bit_exact(name, scan):
    """default: pass 1 of the plain scan on the matrix pipe (a conservative filter, hit_world_mfma);
    RTW_FLAG_SCAN_VALU: the contract discriminant for every sphere on the VALU.  Same image, same counters."""
    g = load_golden(name)
    img, st = gpu_render(g, flags=4 if scan == "valu" else 0)
    assert img.dtype == g["image"].dtype and img.shape == g["image"].shape
    bad = img != g["image"]
    assert not bad.any(), f"{bad.sum()} of {bad.size} channels differ; max abs diff {np.abs(img - g['image']).max()}"
    assert st.segments == g["segments"]
    assert st.samples == g["width"] * g["height"] * g["spp"]
    assert st.sphere_tests == g["segments"] * g["flat"]["n"]


@pytest.mark.parametrize("T", [np.float32, np.float64])
@pytest.mark.parametrize("log2_scale", [-45, -30, 0, 12, 30, 45])
def test_matrix_scan_over_extreme_extents(oracle, rtw, T, log2_scale):
    """the f16-split filter scales lengths by a power of two chosen from the scene's extent; scenes beyond 2^+-40
    fall back to the VALU scan inside the library.  Whole scene and camera scaled by 2^k: GPU == oracle."""
    g = load_golden("metal4_96x54_8spp_d16_f32")
    k = 2.0 ** log2_scale
    flat = {key: (v.astype(T) if isinstance(v, np.ndarray) and v.dtype.kind == "f" else v) for key, v in g["flat"].items()}
    for key in ("cx", "cy", "cz", "r"):
        flat[key] = (flat[key].astype(np.float64) * k).astype(T)
    cam = {key: np.asarray(v, np.float64) * (k if key in ("origin", "lower_left_corner", "horizontal", "vertical", "lens_radius") else 1.0)
           for key, v in g["cam"].items()}
    cam = {key: v.astype(T) for key, v in cam.items()}
    gg = dict(g, flat=flat, cam=cam, image=np.zeros((1, 1, 3), T))
    img, st = gpu_render(gg, spp=4, n_chunks=2)
    ref, ost = oracle.render(flat, cam, g["width"], g["height"], 4, T=T, max_depth=g["depth"], seed=g["seed"], n_chunks=2)
    assert np.array_equal(img, ref, equal_nan=True) and st.segments == ost["segments"]


@pytest.mark.parametrize("T", [np.float32, np.float64])
def test_far_cameras_inside_and_beyond_the_filter_range(oracle, rtw, T):
    """the matrix-pipe filter covers ray origins up to 64 x the scene's extent (here 100.5: the ground sphere); farther
    rays take every sphere as a candidate.  Telephoto cameras at 3 ... 500 x the extent: GPU == oracle."""
    scene = rtw.scene_4_spheres(elem_type=T)
    flat = rtw.flatten_scene(scene, T)
    for dist, vfov in ((300.0, 0.6), (5000.0, 0.04), (20000.0, 0.01), (50000.0, 0.004)):
        cam = rtw.default_camera((0.3, 0.2 * dist, dist), (0, 0, -1), (0, 1, 0), vfov, 16 / 9, 0.0, dist, elem_type=T)
        img = rtw.render(scene, cam, 64, 3, depth=8, n_chunks=3)
        ref, _ = oracle.render(flat, cam, 64, 36, 3, T=T, max_depth=8, seed=1, n_chunks=3)
        assert np.array_equal(img, ref), dist
        assert np.unique(ref.reshape(-1, 3), axis=0).shape[0] > 50       # the spheres are in view


def test_default_chunk_rule_matches_oracle(oracle):
    g = load_golden("cfg1_2spheres_96x54_16spp_d4_f32")
    for spp in (1, 5, 16, 40, 300, 700):
        img, st = gpu_render(g, spp=spp, n_chunks=0)
        ref, ost = oracle.render(g["flat"], g["cam"], g["width"], g["height"], spp, T=np.float32,
                                 max_depth=g["depth"], seed=g["seed"], n_chunks=oracle.default_n_chunks(spp))
        assert np.array_equal(img, ref) and st.segments == ost["segments"], spp
        cs = -(-spp // oracle.default_n_chunks(spp))
        assert st.n_chunks == -(-spp // cs)            # the non-empty chunks of the default rule


def test_ragged_sizes_and_edge_tiles(oracle):
    """widths whose height/width are not multiples of the 8x8 tile; 1-pixel-high image"""
    g = load_golden("metal4_96x54_8spp_d16_f32")
    for width in (100, 33, 17, 2):
        h = (width * 9) // 16
        img, st = gpu_render(g, width=width, height=h, spp=3, n_chunks=2)
        ref, ost = oracle.render(g["flat"], g["cam"], width, h, 3, T=np.float32, max_depth=g["depth"],
                                 seed=g["seed"], n_chunks=2)
        assert img.shape == (h, width, 3)
        assert np.array_equal(img, ref) and st.segments == ost["segments"], width


def test_empty_scene_is_sky(oracle):
    g = load_golden("cfg1_2spheres_96x54_16spp_d4_f32")
    empty = {k: (v[:0] if k != "n" else 0) for k, v in g["flat"].items()}
    img, st = gpu_render(dict(g, flat=empty), spp=2, n_chunks=1)
    ref, _ = oracle.render(empty, g["cam"], g["width"], g["height"], 2, T=np.float32, max_depth=g["depth"],
                           seed=g["seed"], n_chunks=1)
    assert np.array_equal(img, ref) and st.segments == g["width"] * g["height"] * 2


def test_depth_limits(oracle):
    g = load_golden("cfg1_2spheres_96x54_16spp_d4_f32")
    img0, st0 = gpu_render(g, max_depth=0, spp=2)
    assert np.all(img0 == 0) and st0.segments == 0                       # src/ray_color.jl:15-17
    for depth in (1, 2, 50):
        img, st = gpu_render(g, max_depth=depth, spp=4, n_chunks=4)
        ref, ost = oracle.render(g["flat"], g["cam"], g["width"], g["height"], 4, T=np.float32, max_depth=depth,
                                 seed=g["seed"], n_chunks=4)
        assert np.array_equal(img, ref) and st.segments == ost["segments"]


def test_seed_changes_image_and_repeat_is_deterministic():
    g = load_golden("diel_bubble_96x54_8spp_d16_f32")
    a, _ = gpu_render(g)
    b, _ = gpu_render(g)
    c, _ = gpu_render(g, seed=2)
    assert np.array_equal(a, b) and not np.array_equal(a, c)


def test_linear_output_and_gamma(oracle):
    g = load_golden("metal4_96x54_8spp_d16_f32")
    lin, _ = gpu_render(g, gamma=0)
    gam, _ = gpu_render(g, gamma=1)
    ref, _ = oracle.render(g["flat"], g["cam"], g["width"], g["height"], g["spp"], T=np.float32,
                           max_depth=g["depth"], seed=g["seed"], n_chunks=g["n_chunks"], gamma=False)
    assert np.array_equal(lin, ref)
    assert np.allclose(gam.astype(np.float64) ** 2, lin, rtol=3e-7, atol=1e-12)   # rgb_gamma2 = sqrt (src/vec.jl:22)


@pytest.mark.parametrize("count", [2, 3, 8])
def test_shards_sum_to_full_image(count):
    """Multi-GPU partition invariance: the shards are disjoint, zero elsewhere, and their sum is
    bit-identical to the unsharded render (this is what the RCCL reduce computes)."""
    g = load_golden("cfg2_random_320x180_64spp_d16_f32")
    import rtw_amd as R
    full, st_full = gpu_render(g, spp=4, n_chunks=4)
    acc = np.zeros_like(full)
    segs = 0
    for idx in range(count):
        part, st = gpu_render(g, spp=4, n_chunks=4, shard_index=idx, shard_count=count)
        mask = R.owned_pixel_mask(g["width"], idx, count)
        assert np.all(part[~mask] == 0)
        acc += part
        segs += st.segments
    assert np.array_equal(acc, full) and segs == st_full.segments


def test_python_render_api_matches_reference_signature(rtw, oracle):
    """render(scene, cam, image_width, n_samples) -- the drop-in call of src/render.jl:8-9"""
    T = np.float32
    scene = rtw.scene_2_spheres(elem_type=T)
    cam = rtw.t_default_cam(elem_type=T)
    img = rtw.render(scene, cam, 96, 16)                                  # depth 16 = reference default
    assert img.shape == (54, 96, 3) and img.dtype == T
    ref, _ = oracle.render(rtw.flatten_scene(scene, T), cam, 96, 54, 16, T=T, max_depth=16, seed=1, n_chunks=16)
    assert np.array_equal(img, ref)
    img64 = rtw.render(rtw.scene_2_spheres(elem_type=np.float64), rtw.t_default_cam(elem_type=np.float64), 96, 16)
    assert np.array_equal(img64, load_golden("smoke_2spheres_96x54_16spp_d16_f64")["image"])
    assert rtw.render(scene, cam)[..., 0].shape == (225, 400)              # defaults: width 400, 1 sample


def test_device_resident_path_with_torch_stream(rtw):
    import torch
    g = load_golden("cfg1_2spheres_96x54_16spp_d4_f32")
    scene_flat = g["flat"]

    class _S(list):
        pass
    dr = rtw.DeviceRenderer.__new__(rtw.DeviceRenderer)
    # build through the public constructor from real structs instead
    T = np.float32
    dr = rtw.DeviceRenderer(rtw.scene_2_spheres(elem_type=T), rtw.t_default_cam(elem_type=T), device=0)
    fb = torch.empty(96 * 54 * 3, dtype=torch.float32, device="cuda:0")
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        dr.render_into(fb.data_ptr(), 96, 16, depth=4, seed=1, n_chunks=16, stream=stream.cuda_stream)
    stream.synchronize()
    st = dr.stats()
    img = fb.cpu().numpy().reshape(96, 54, 3).transpose(1, 0, 2)
    assert np.array_equal(img, g["image"]) and st["segments"] == g["segments"] and st["kernel_ms"] > 0
    dr.close()


def test_full_size_properties(oracle, rtw):
    """BASELINE configs[2] geometry (1920x1080, scene_random_spheres, depth 50) at a sample count
    the oracle still finishes in seconds: bit-exact vs the live oracle, exact segment count,
    determinism, and shard-sum invariance at full resolution."""
    T = np.float32
    rtw.reseed()
    flat = rtw.flatten_scene(rtw.scene_random_spheres(elem_type=T), T)
    cam = rtw.t_cam1(elem_type=T)
    g = dict(flat=flat, cam={k: getattr(cam, k) for k in oracle.CAM_FIELDS + ("lens_radius",)},
             image=np.zeros(1, T), width=1920, height=1080, spp=2, depth=50, seed=1, n_chunks=2)
    img, st = gpu_render(g)
    ref, ost = oracle.render(flat, cam, 1920, 1080, 2, T=T, max_depth=50, seed=1, n_chunks=2)
    assert np.array_equal(img, ref) and st.segments == ost["segments"]
    a, _ = gpu_render(g, shard_index=0, shard_count=2)
    b, _ = gpu_render(g, shard_index=1, shard_count=2)
    assert np.array_equal(a + b, img)
    # the picture is the right picture: sky gradient on top, three big spheres, grey ground
    assert img[:200].mean() > 0.7 and 0.2 < img[900:, :, :].mean() < 0.8


def test_large_scene_global_gather_path(oracle):
    """2 000 spheres: geom no longer fits the LDS staging budget, pass 2 gathers from global memory"""
    rng = np.random.default_rng(7)
    n = 2000
    T = np.float32
    flat = dict(n=n, cx=rng.uniform(-8, 8, n).astype(T), cy=rng.uniform(-3, 3, n).astype(T),
                cz=rng.uniform(-12, -2, n).astype(T), r=rng.uniform(0.05, 0.3, n).astype(T),
                kind=rng.integers(0, 3, n).astype(np.int32), ar=rng.uniform(0, 1, n).astype(T),
                ag=rng.uniform(0, 1, n).astype(T), ab=rng.uniform(0, 1, n).astype(T),
                param=np.zeros(n, T))
    flat["param"][flat["kind"] == 1] = rng.uniform(0, 1, int((flat["kind"] == 1).sum())).astype(T)
    flat["param"][flat["kind"] == 2] = T(1.5)
    g = load_golden("cfg1_2spheres_96x54_16spp_d4_f32")
    g = dict(g, flat=flat)
    img, st = gpu_render(g, width=64, height=36, spp=3, n_chunks=3, max_depth=8)
    ref, ost = oracle.render(flat, g["cam"], 64, 36, 3, T=T, max_depth=8, seed=g["seed"], n_chunks=3)
    assert np.array_equal(img, ref) and st.segments == ost["segments"]
    assert st.sphere_tests == ost["segments"] * n


def test_too_many_spheres_is_an_error():
    import ctypes as C
    from rtw_amd import _capi
    n = 70000
    z = np.zeros(n, np.float32)
    flat = dict(n=n, cx=z, cy=z, cz=z, r=z + 1, kind=np.zeros(n, np.int32), ar=z, ag=z, ab=z, param=z)
    S, keep = _capi.make_scene(flat, np.float32)
    h = C.c_void_p()
    rc = _capi.lib().rtw_scene_upload_f32(C.byref(S), 0, C.byref(h))
    assert rc == -5 and b"too many spheres" in _capi.lib().rtw_last_error()


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_group_cull_mode_is_bit_identical(name):
    """opt-in accelerated scan (RTW_FLAG_GROUP_CULL), on the matrix pipe (block culling in front of the filter) and in its
    all-VALU form (| RTW_FLAG_SCAN_VALU): same image, same segment count"""
    g = load_golden(name)
    for flags in (1, 5):
        img, st = gpu_render(g, flags=flags)
        assert np.array_equal(img, g["image"]) and st.segments == g["segments"], flags


def test_group_cull_mode_full_size_and_shards(oracle, rtw):
    T = np.float32
    rtw.reseed()
    flat = rtw.flatten_scene(rtw.scene_random_spheres(elem_type=T), T)
    cam = rtw.t_cam1(elem_type=T)
    g = dict(flat=flat, cam={k: getattr(cam, k) for k in oracle.CAM_FIELDS + ("lens_radius",)},
             image=np.zeros(1, T), width=1920, height=1080, spp=2, depth=50, seed=1, n_chunks=2)
    plain, st0 = gpu_render(g)
    fast, st1 = gpu_render(g, flags=1)
    assert np.array_equal(plain, fast) and st0.segments == st1.segments
    a, _ = gpu_render(g, flags=1, shard_index=1, shard_count=3)
    b, _ = gpu_render(g, shard_index=1, shard_count=3)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("T", [np.float32, np.float64])
def test_large_and_degenerate_scenes_in_every_scan_mode(oracle, T):
    """1 ... 2000 spheres (2000 do not fit the LDS copy: the global-memory instantiations) with coincident spheres,
    negative radii: group cull on the matrix pipe (flags 1) and on the VALU (5), matrix-pipe plain scan (0) and all-VALU
    plain scan (4) all equal the oracle"""
    rng = np.random.default_rng(11)
    g0 = load_golden("cfg1_2spheres_96x54_16spp_d4_f32")
    g0 = dict(g0, cam={k: np.asarray(v).astype(T) for k, v in g0["cam"].items()})
    for n in (1, 3, 5, 2000):
        flat = dict(n=n, cx=rng.uniform(-4, 4, n).astype(T), cy=rng.uniform(-2, 2, n).astype(T),
                    cz=rng.uniform(-9, -2, n).astype(T), r=(rng.uniform(0.1, 0.5, n) * rng.choice([1, -1], n)).astype(T),
                    kind=rng.integers(0, 3, n).astype(np.int32), ar=rng.uniform(0, 1, n).astype(T),
                    ag=rng.uniform(0, 1, n).astype(T), ab=rng.uniform(0, 1, n).astype(T), param=np.full(n, 1.5, T))
        flat["cx"][0], flat["cy"][0], flat["cz"][0] = flat["cx"][-1], flat["cy"][-1], flat["cz"][-1]   # coincident spheres: exact ties
        flat["r"][0] = flat["r"][-1]
        g = dict(g0, flat=flat, image=np.zeros((1, 1, 3), T))
        ref, ost = oracle.render(flat, g["cam"], 64, 36, 2, T=T, max_depth=6, seed=g["seed"], n_chunks=2)
        for flags in (1, 5, 0, 4):           # group cull on the matrix pipe / on the VALU, plain scan on the matrix pipe / on the VALU
            img, st = gpu_render(g, width=64, height=36, spp=2, n_chunks=2, max_depth=6, flags=flags)
            assert np.array_equal(img, ref) and st.segments == ost["segments"], (n, flags)
