"""Round-6 GPU tests (all through the C ABI): the scheduler of the lane loop -- out-of-order job slots, the second hand-out round, static
first claims -- has no effect on images by construction (every golden / oracle comparison elsewhere runs through it); what an image cannot
show is a job rendered TWICE (each job stores its own pixels) or counters cleared at the wrong time: the kernel's sample counter can."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_shards", [1, 3, 6])
def test_sample_and_segment_counters_are_exact_on_fresh_records(rtw, oracle, n_shards):
    """N shards on ONE device run as N concurrent kernels, each on a render record of its own -- records that this call may have to
    create.  (Round 6 cleared a new record with a hipMemset on the null stream, which the renders' non-blocking streams do not wait for:
    it could land in the middle of the first kernel, reset the job queues -- jobs rendered twice, the image unchanged -- and the counters.)"""
    T = np.float32
    scene, cam = rtw.scene_4_spheres(elem_type=T), rtw.t_default_cam(elem_type=T)
    W, spp = 200, 6
    H = rtw.image_height(W)
    ref, ost = oracle.render(rtw.flatten_scene(scene, T), cam, W, H, spp, T=T, max_depth=8, seed=1)
    for _ in range(4):
        img = rtw.render(scene, cam, W, spp, depth=8, devices=[0] * n_shards)
        st = rtw.last_stats()
        assert np.array_equal(img, ref)
        assert st["samples"] == W * H * spp and st["segments"] == ost["segments"], (st["samples"], W * H * spp)
        assert len(st["per_device"]) == n_shards


@pytest.mark.parametrize("W,spp,depth", [(8, 1, 4), (16, 3, 4), (40, 17, 8), (96, 16, 4), (200, 32, 16), (320, 64, 16)])
def test_small_frames_count_every_sample_once(rtw, W, spp, depth):
    """frames of a handful of jobs: fewer jobs than workgroups, static claims that cover the whole queue or none of it, one batch per wave"""
    T = np.float32
    rtw.reseed()
    scene, cam = rtw.scene_random_spheres(elem_type=T), rtw.t_cam1(elem_type=T)
    H = rtw.image_height(W)
    seen = set()
    for _ in range(10):
        rtw.render(scene, cam, W, spp, depth=depth)
        seen.add(rtw.last_stats()["samples"])
    assert seen == {W * H * spp}


def test_more_shards_than_tiles_reports_every_shard(rtw):
    """a 16 x 9 frame has 4 tiles: shards 4 and 5 of six own nothing, launch nothing and are still listed (rtw_stats_devices: one entry per shard)"""
    T = np.float32
    scene, cam = rtw.scene_2_spheres(elem_type=T), rtw.t_default_cam(elem_type=T)
    one = rtw.render(scene, cam, 16, 4, depth=4)
    six = rtw.render(scene, cam, 16, 4, depth=4, devices=[0] * 6)
    st = rtw.last_stats()
    assert np.array_equal(one, six) and st["samples"] == 16 * 9 * 4
    assert len(st["per_device"]) == 6 and [ms for _, ms in st["per_device"]][4:] == [0.0, 0.0]


@pytest.mark.parametrize("T", [np.float32, np.float64])
@pytest.mark.parametrize("job_pixels,spp,n_chunks", [(16, 24, 24), (16, 5, 0), (8, 40, 40), (4, 40, 8), (4, 64, 64), (1, 130, 130), (1, 40, 0)])
def test_job_slots_with_stragglers(rtw, oracle, T, job_pixels, spp, n_chunks):
    """Depth 50 in the reference's scene (glass spheres: paths of dozens of bounces next to paths of one), every job size -- 4 slots of 16
    pixels ... 24 of one --, chunkings with one batch per job and with many: slots are recycled out of order around the stragglers; image,
    segment and sample counters against the oracle."""
    rtw.reseed()
    scene, cam = rtw.scene_random_spheres(elem_type=T), rtw.t_cam1(elem_type=T)
    W, H = 112, 63
    flat = rtw.flatten_scene(scene, T)
    nch = n_chunks or oracle.default_n_chunks(spp)
    ref, ost = oracle.render(flat, cam, W, H, spp, T=T, max_depth=50, seed=5, n_chunks=nch)
    from test_gpu_render import gpu_render
    g = dict(flat=flat, cam={k: np.asarray(getattr(cam, k)) for k in oracle.CAM_FIELDS} | {"lens_radius": np.asarray(cam.lens_radius)},
             image=np.zeros(1, T), width=W, height=H, spp=spp, depth=50, seed=5, n_chunks=n_chunks)
    for flags in (0, 1):
        img, st = gpu_render(g, job_pixels=job_pixels, flags=flags)
        assert np.array_equal(img, ref), (flags, int((img != ref).sum()))
        assert st.samples == W * H * spp and st.segments == ost["segments"]


def test_short_sqrt_and_reciprocal_are_correctly_rounded_for_every_binary32():
    """The kernels' square root and reciprocal (rtw_path.hpp t_sqrt / t_rcp: Markstein's FMA sequences on v_rsq_f32 / v_rcp_f32 inside a range
    check, the compiler's IEEE sequence outside) against __builtin_sqrtf / 1.0f / x for ALL 2^32 binary32 bit patterns, on the device: the
    arithmetic of hit(::Sphere)'s root (src/hit.jl:20) and of normalize (src/rand.jl:29, src/camera.jl:46) stays one rounding per operation."""
    from test_gpu_units import run_unit
    n_items, per = 1 << 16, 1 << 16
    x = np.stack([np.arange(n_items, dtype=np.float64) * per, np.full(n_items, per, np.float64)], axis=1)
    y = run_unit(16, x, 4, np.float32)
    bad_s, bad_r = int(y[:, 0].sum()), int(y[:, 1].sum())
    first = [int(v) for v in y[:, 2:].ravel() if v >= 0][:4]
    assert bad_s == 0 and bad_r == 0, (bad_s, bad_r, [hex(v) for v in first])
