"""Round-3 GPU tests (all through the C ABI / bench.py as a user runs it):
  * the N > 1 control flow of bench.py from a cold shell (self-spawned ranks, both collectives, frame identical to the 1-rank
    frame) and its refusal to mislabel; 8-way sharding at 1920x1080;
  * the matrix-pipe filter's full-scale parity: all three scan modes at the headline scale, 1080p x 32 spp against the live
    oracle, a 60-second slice of the soak (tools/gpu_soak.py);
  * the generator core on the DEVICE against the published xoroshiro128+ jump polynomial; near_zero KAT (src/vec.jl:20);
  * tier T3 for Float64 and for the dielectric scene through the wide-aperture camera;
  * the host-buffer entry point's persistent per-device context."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from conftest import ROOT, all_numerics
from test_gpu_render import gpu_render
from test_gpu_round2 import _cam_dict, _random_spheres_case
from test_gpu_units import run_unit

pytestmark = pytest.mark.gpu


# ---- bench.py: N > 1 from a plain command ------------------------------------------------------------------------------
def _bench(args, env_extra=None, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "RTW_BENCH_ONE_DEVICE"):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, (json.loads(lines[-1]) if lines else None)


SMALL = ["--steps", "1", "--warmup", "1", "--spp", "8", "--no-cpu-baseline", "--no-extras"]


def test_bench_gpus2_from_a_plain_command_matches_one_rank():
    """`python bench.py --gpus 2` with no launcher: the script starts its two ranks itself, the process group reports world size
    2, every rank reports a kernel time, and the assembled frame (reduce AND gather) is the 1-rank frame bit for bit.
    One-GPU box: RTW_BENCH_ONE_DEVICE=1 puts both ranks on cuda:0 with the collective over gloo (the control flow, not RCCL)."""
    p1, one = _bench(["--gpus", "1"] + SMALL)
    assert p1.returncode == 0 and one["n_gpus"] == 1 and one["world_size_observed"] == 1, p1.stderr[-2000:]
    for coll in ("reduce", "gather"):
        p2, two = _bench(["--gpus", "2", "--collective", coll] + SMALL, {"RTW_BENCH_ONE_DEVICE": "1"})
        assert p2.returncode == 0, p2.stderr[-3000:]
        assert two["n_gpus"] == 2 and two["world_size_observed"] == 2 and two["backend"] == "gloo" and two["one_device_emulation"] is True
        assert two["launched_by"].startswith("bench.py (self-spawned")
        assert [r["rank"] for r in two["per_rank"]] == [0, 1] and all(r["kernel_ms"] > 0 for r in two["per_rank"])
        assert two["frame_sha256"] == one["frame_sha256"], coll
        assert two["scaling"] == "strong" and f"1 RCCL {coll}" in two["config"]["parallelism"]


def test_bench_refuses_to_time_fewer_gpus_than_asked_for():
    """A bare `--gpus N` with fewer than N visible devices must fail, not print a 1-GPU number labelled N."""
    import torch
    n = torch.cuda.device_count() + 1
    p, line = _bench(["--gpus", str(n)] + SMALL)
    assert p.returncode != 0 and line is None
    assert "refusing" in (p.stderr + p.stdout)
    # and a launcher that starts the wrong number of ranks is caught too
    p, line = _bench(["--gpus", "2"] + SMALL, {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and line is None and "WORLD_SIZE=1" in (p.stderr + p.stdout)


@all_numerics
def test_eight_shards_at_1920x1080_sum_to_the_full_frame(rtw):
    """configs[3]'s partition at its real size: 8 shards of 1920x1080 (zero elsewhere) sum to the unsharded frame bit for bit,
    and the compact tile-major shards reassemble to it as well -- what the 8-rank reduce / gather compute."""
    import torch
    T = np.float32
    W, H, spp = 1920, 1080, 4
    rtw.reseed()
    dr = rtw.DeviceRenderer(rtw.scene_random_spheres(elem_type=T), rtw.t_cam1(elem_type=T), device=0)
    s = torch.cuda.current_stream()
    full = torch.empty(H * W * 3, dtype=torch.float32, device="cuda:0")
    dr.render_into(full.data_ptr(), W, spp, depth=50, seed=1, stream=s.cuda_stream, n_elems=full.numel())
    seg_full = dr.stats()["segments"]
    acc = torch.zeros_like(full)
    part = torch.empty_like(full)
    frame = torch.zeros(H * W, 3, dtype=torch.float32, device="cuda:0")
    segs = 0
    for idx in range(8):
        dr.render_into(part.data_ptr(), W, spp, depth=50, seed=1, stream=s.cuda_stream, shard_index=idx, shard_count=8)
        segs += dr.stats()["segments"]
        mask = torch.from_numpy(np.ascontiguousarray(rtw.owned_pixel_mask(W, idx, 8).T)).to("cuda:0").reshape(-1)   # column-major
        assert bool((part.reshape(-1, 3)[~mask] == 0).all())
        acc += part
        n = rtw.compact_elems(W, idx, 8)
        comp = torch.empty(n, dtype=torch.float32, device="cuda:0")
        dr.render_into(comp.data_ptr(), W, spp, depth=50, seed=1, stream=s.cuda_stream, shard_index=idx, shard_count=8, compact=True, n_elems=n)
        dest = rtw.compact_to_frame_index(W, idx, 8)
        src = np.flatnonzero(dest >= 0)
        frame.index_copy_(0, torch.from_numpy(dest[src]).to("cuda:0"), comp.reshape(-1, 3).index_select(0, torch.from_numpy(src).to("cuda:0")))
    assert segs == seg_full
    assert bool(torch.equal(acc, full)) and bool(torch.equal(frame.reshape(-1), full))
    with pytest.raises(ValueError):
        dr.render_into(part.data_ptr(), W, spp, n_elems=10)
    dr.close()


# ---- the filter at full scale ---------------------------------------------------------------------------------------------
@all_numerics
def test_three_scan_modes_identical_at_headline_scale(rtw):
    """BASELINE configs[2] in full (1920x1080, 1000 spp, depth 50: 8.2e9 ray segments) in all three scan modes: the
    matrix-pipe filter, the INDEPENDENT all-VALU scan (the contract discriminant for every sphere: no filter, no margin
    constants) and the opt-in group cull must give one image bit for bit and one segment count.  A candidate lost by the
    f16 filter would show here against the all-VALU leg (the cull leg shares the filter)."""
    import torch
    T = np.float32
    rtw.reseed()
    dr = rtw.DeviceRenderer(rtw.scene_random_spheres(elem_type=T), rtw.t_cam1(elem_type=T), device=0)
    a = torch.empty(1080 * 1920 * 3, dtype=torch.float32, device="cuda:0")
    b, c = torch.empty_like(a), torch.empty_like(a)
    s = torch.cuda.current_stream()
    dr.render_into(a.data_ptr(), 1920, 1000, depth=50, seed=1, stream=s.cuda_stream)
    sa = dr.stats()
    dr.render_into(b.data_ptr(), 1920, 1000, depth=50, seed=1, stream=s.cuda_stream, scan_valu=True)
    sb = dr.stats()
    dr.render_into(c.data_ptr(), 1920, 1000, depth=50, seed=1, stream=s.cuda_stream, group_cull=True)
    sc = dr.stats()
    assert sa["segments"] == sb["segments"] == sc["segments"] and sa["samples"] == 1920 * 1080 * 1000
    assert bool(torch.equal(a, b)), int((a != b).sum())
    assert bool(torch.equal(a, c)), int((a != c).sum())
    dr.close()


@all_numerics
def test_1080p_32spp_against_the_live_oracle(oracle, rtw):
    """1920x1080 x 32 spp, depth 50, Float32 against the oracle rendered here (6.6e7 samples, 2.6e8 segments): bit-exact image
    and segment count, in the matrix-pipe mode and in the cull mode."""
    T = np.float32
    g, cam = _random_spheres_case(rtw, oracle, T, 1920, 32, depth=50)
    ref, ost = oracle.render(g["flat"], cam, 1920, 1080, 32, T=T, max_depth=50, seed=1)
    for flags in (0, 1):
        img, st = gpu_render(g, flags=flags)
        assert st.segments == ost["segments"]
        assert np.array_equal(img, ref), int((img != ref).sum())


SOAK_FIXED_SEEDS = [100000, 424242, 31337000, 271828183]      # (even -> the first scene is Float32, then alternating)


def test_soak_slice():
    """tools/gpu_soak.py: random scenes x 131 072 random rays through the matrix-pipe scan and its block-culling form, then random small
    renders in all four scan modes -- 0 mismatches with the oracle.  (The filter's margin is derived by hand and rests on measured MFMA
    accumulation behaviour: this is its gate.)  A FIXED seed list with a fixed amount of work per seed: a red run reproduces
    (`python tools/gpu_soak.py <seconds> <seed>`).  The time-budgeted slice on a seed that changes from day to day is NOT a test any more
    (round 6: a gate's inputs do not depend on the calendar or the box's speed): `python tools/gpu_soak.py --day` runs it."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gpu_soak
    import rtw_oracle as O
    O.build(); O.lib()
    for seed0 in SOAK_FIXED_SEEDS:
        msgs = []
        rounds, rays, bad, _ = gpu_soak.scan_rounds(1e9, seed0, msgs.append, max_rounds=24)
        imgs, bad_imgs, _ = gpu_soak.render_rounds(1e9, seed0, msgs.append, max_rounds=16)
        assert rounds == 24 and imgs == 16 * 4
        assert bad == 0 and bad_imgs == 0, (f"fixed seed {seed0}", msgs[:10])


# ---- generator core and near_zero on the device ---------------------------------------------------------------------------
@pytest.mark.parametrize("T", [np.float32, np.float64])
def test_device_generator_is_the_2016_xoroshiro128plus(oracle, T):
    """The device's state transition (unit op 7 = four trand calls) on the 128 basis states: it is GF(2)-linear, its columns
    equal the oracle's, and the authors' published 2016 jump polynomial equals 2^64 steps of its single-step transition
    (recovered as the 4th root: the columns of M^4 determine the check  P_J(M)^4 = M^(4 * 2^64))."""
    from test_oracle_round3 import JUMP_2016, JUMP_2018, M64, _apply
    st = np.zeros((128, 2), np.uint64)
    for i in range(128):
        st[i, i // 64] = np.uint64(1) << np.uint64(i % 64)
    y = run_unit(7, st.view(np.float64), 6, T)
    dev4 = [int(a) | (int(b) << 64) for a, b in y[:, :2].copy().view(np.uint64)]

    def oracle4(x, yy):
        s = np.array([x, yy], np.uint64)
        for _ in range(4):
            oracle.rng_next(s)
        return int(s[0]) | (int(s[1]) << 64)
    assert dev4 == [oracle4(int(a), int(b)) for a, b in st]
    # P_J(M4) over the 4-step map: sum_{k in J} M^(4k) must equal (M^4)^(2^64); holds for the right constants only
    P = dev4
    for _ in range(64):
        P = [_apply(P, c) for c in P]
    for jump, want in ((JUMP_2016, True), (JUMP_2018, False)):
        v = 0x0123456789ABCDEF0FEDCBA987654321
        acc, cur = 0, v
        for k in range(128):
            if (jump >> k) & 1:
                acc ^= cur
            cur = _apply(dev4, cur)
        assert (acc == _apply(P, v)) is want
    # the uniforms are the low 23 / 52 bits of x + y of the pre-step state (src/rand.jl:12; Julia 1.6/1.7 Random)
    s0 = np.array([[0x0123456789ABCDEF, 0x0FEDCBA987654321]], np.uint64)
    u0 = run_unit(7, s0.view(np.float64), 6, T)[0, 2]
    out = (0x0123456789ABCDEF + 0x0FEDCBA987654321) & M64
    assert u0 == ((out & 0x7FFFFF) / 2.0 ** 23 if T is np.float32 else (out & ((1 << 52) - 1)) / 2.0 ** 52)


@pytest.mark.parametrize("T", [np.float32, np.float64])
def test_near_zero_kat_on_the_device(oracle, T):
    """near_zero(v) = squared_length(v) < 1e-5 with the Float64 literal (src/vec.jl:19-20): the reference's own assertion
    `!near_zero(SA[0.4, 0.5, 0.1])` (test/runtests.jl:131), the threshold from both sides in T's arithmetic, and random
    vectors against the oracle."""
    rng = np.random.default_rng(5)
    thr = np.sqrt(1e-5)
    v = [[0.4, 0.5, 0.1], [0, 0, 0], [thr * 0.9999, 0, 0], [thr * 1.0001, 0, 0], [0.003, 0.001, 0.0005], [1e-3, 2e-3, 2.5e-3],
         [-0.0018257, 0.0018257, -0.0018257], [-0.0018258, 0.0018258, -0.0018258], [1e-30, -1e-30, 0], [0, -0.00316227, 0], [0, 0.00316228, 0]]
    v = np.array(v + list(rng.normal(size=(500, 3)) * 10.0 ** rng.uniform(-4, -2, (500, 1))), np.float64).astype(T).astype(np.float64)
    y = run_unit(15, v, 2, T)
    assert y[0, 0] == 0.0                                                  # test/runtests.jl:131
    assert y[1, 0] == 1.0 and y[2, 0] == 1.0 and y[3, 0] == 0.0
    both = set()
    for i in range(len(v)):
        x = v[i].astype(T)
        sl = (x[0] * x[0] + x[1] * x[1]) + x[2] * x[2]                     # StaticArrays dot, in T
        assert y[i, 1] == float(sl)
        want = oracle.near_zero(v[i], T)
        assert bool(y[i, 0]) == bool(want) == (float(sl) < 1e-5)
        both.add(bool(want))
    assert both == {True, False}


# ---- tier T3 beyond Float32 / scene_random_spheres --------------------------------------------------------------------
def _t3(oracle, g, cam, T, W, H, spp, depth, what):
    A, _ = gpu_render(g, gamma=0)
    B, _ = gpu_render(g, gamma=0, seed=2)
    Rr, _ = oracle.render(g["flat"], cam, W, H, spp, T=T, max_depth=depth, rng_mode=oracle.REF_SERIAL, ref_threads=H,
                          product_order=oracle.PRODUCT_REFERENCE, gamma=False)
    A, B, Rr = A.astype(np.float64), B.astype(np.float64), Rr.astype(np.float64)
    D1, D2 = A - B, A - Rr
    N = D2.size
    assert abs(D2.mean()) <= 4 * D2.std() / np.sqrt(N), (what, D2.mean(), D2.std())
    ratio = (D2 ** 2).mean() / (D1 ** 2).mean()
    assert 0.90 <= ratio <= 1.10, (what, ratio)

    def blocks(x, f):
        hh, ww = (H // 8) * 8, (W // 8) * 8
        return f(x[:hh, :ww].reshape(hh // 8, 8, ww // 8, 8, 3), axis=(1, 3))
    var_hat = np.repeat(np.repeat(blocks(D1 ** 2, np.mean), 8, 0), 8, 1)
    hh, ww = var_hat.shape[:2]
    z = np.abs(D2[:hh, :ww]) / np.sqrt(np.maximum(var_hat, 1e-12))
    assert (z <= 4.5).mean() >= 0.995, (what, (z <= 4.5).mean())
    m1, m2 = np.abs(blocks(D1, np.mean)).max(), np.abs(blocks(D2, np.mean)).max()
    assert m2 <= 1.6 * m1, (what, m1, m2)


@all_numerics
def test_t3_float64_random_spheres(oracle, rtw):
    """Tier T3 (tolerances of test_t3_statistical_parity_with_ref_serial) in the reference's own headline precision, Float64:
    GPU PIXEL_STREAM vs the oracle's REF_SERIAL (= `julia -t 180`), scene_random_spheres, 320x180, 1024 spp, depth 16."""
    T = np.float64
    g, cam = _random_spheres_case(rtw, oracle, T, 320, 1024, depth=16)
    _t3(oracle, g, cam, T, 320, 180, 1024, 16, "f64 random spheres")


@all_numerics
def test_t3_dielectric_scene_wide_aperture(oracle, rtw):
    """Tier T3 on the dielectric-heavy scene (scene_diel_spheres: hollow glass, negative radius, total internal reflection)
    through t_cam2 (aperture 2.0: the lens sampler matters), Float32, 320x180, 1024 spp, depth 16."""
    T = np.float32
    flat = rtw.flatten_scene(rtw.scene_diel_spheres(elem_type=T), T)
    cam = rtw.t_cam2(elem_type=T)
    g = dict(flat=flat, cam=_cam_dict(cam, oracle), image=np.zeros(1, T), width=320, height=180, spp=1024, depth=16, seed=1,
             n_chunks=oracle.default_n_chunks(1024))
    _t3(oracle, g, cam, T, 320, 180, 1024, 16, "diel spheres, t_cam2")


# ---- the host-buffer entry point keeps its per-device context ----------------------------------------------------------
def test_host_entry_point_reuses_its_context(rtw):
    """rtw_render_f32 (what the Julia `render()` shim binds) at 1920x1080: from the second call on the scene upload, stream and
    device image are reused and a call costs its kernel + one 24.9 MB D2H: wall - kernel <= 1.5 ms through the C ABI into a
    touched buffer (measured 0.49 ms; round 2: 9 - 14 ms), <= 5 ms through the Python mirror (scene flattening + a fresh numpy
    image included).  A changed scene is noticed (new upload, new image).  The in-library multi-device path -- shards gathered in
    HBM of the first device, one D2H -- gives the same frame with a repeated ordinal standing for 8 devices."""
    import ctypes as C
    from rtw_amd import _capi
    T = np.float32
    rtw.reseed()
    scene = rtw.scene_random_spheres(elem_type=T)
    cam = rtw.t_cam1(elem_type=T)
    L = _capi.lib()
    flat = rtw.flatten_scene(scene, T)
    S, keep = _capi.make_scene(flat, T)
    Cm = _capi.make_camera(cam, T)
    P = _capi.make_params(1920, 1080, 20, 50, 1, 0, 0, 1, -1, 1, 0)
    out = np.zeros(1080 * 1920 * 3, dtype=T)
    over = []
    for _ in range(5):
        t = time.perf_counter()
        _capi.check(L.rtw_render_f32(C.byref(S), C.byref(Cm), C.byref(P), out.ctypes.data_as(C.c_void_p)))
        wall = (time.perf_counter() - t) * 1e3
        st = _capi.Stats()
        _capi.check(L.rtw_stats(C.byref(st)))
        over.append(wall - st.kernel_ms)
    assert min(over[1:]) <= 1.5, over
    ref = out.copy()
    over_py = []
    for _ in range(4):
        t = time.perf_counter()
        img = rtw.render(scene, cam, 1920, 20, depth=50, seed=1)
        over_py.append((time.perf_counter() - t) * 1e3 - rtw.last_stats()["kernel_ms"])
    assert min(over_py) <= 5.0, over_py
    assert np.array_equal(np.ascontiguousarray(img.transpose(1, 0, 2)).reshape(-1), ref)
    # eight "devices" (ordinal 0 repeated): same frame, segments add up, context reused on the second call
    t8 = []
    for _ in range(2):
        t = time.perf_counter()
        img8 = rtw.render(scene, cam, 1920, 20, depth=50, seed=1, devices=[0] * 8)
        t8.append(time.perf_counter() - t)
    assert np.array_equal(img8, img) and rtw.last_stats()["segments"] == st.segments
    # a different scene through the same context
    flat2 = dict(flat)
    flat2["cy"] = flat["cy"].copy()
    flat2["cy"][1] += T(0.25)
    S2, keep2 = _capi.make_scene(flat2, T)
    out2 = np.zeros_like(out)
    _capi.check(L.rtw_render_f32(C.byref(S2), C.byref(Cm), C.byref(P), out2.ctypes.data_as(C.c_void_p)))
    assert not np.array_equal(out2, ref)
    _capi.check(L.rtw_render_f32(C.byref(S), C.byref(Cm), C.byref(P), out2.ctypes.data_as(C.c_void_p)))
    assert np.array_equal(out2, ref)
    del keep, keep2


# ---- job queues: frames that fill the eight queues very unevenly, and the size limit ------------------------------------
@all_numerics
@pytest.mark.parametrize("W,H", [(8, 2048), (24, 1000), (2048, 8), (72, 9)])
def test_narrow_and_flat_frames(oracle, rtw, W, H):
    """Tile column tj belongs to job queue tj mod 8 (claim_job): an 8-pixel-wide frame has ONE non-empty queue, a 24-pixel-wide one
    three -- every workgroup then takes its jobs from another die's queue.  The C ABI takes the height from the caller, so frames
    need not be 16:9.  Bit-exact against the oracle, every pixel sample counted once."""
    from test_gpu_render import gpu_render
    T = np.float32
    rtw.reseed()
    scene = rtw.scene_random_spheres(elem_type=T)
    cam = rtw.t_cam1(elem_type=T)
    g = dict(flat=rtw.flatten_scene(scene, T), cam=_cam_dict(cam, oracle), image=np.zeros(1, T), width=W, height=H, spp=8, depth=16,
             seed=5, n_chunks=2)
    img, st = gpu_render(g)
    ref, ost = oracle.render(g["flat"], g["cam"], W, H, 8, T=T, max_depth=16, seed=5, n_chunks=2)
    assert np.array_equal(img, ref)
    assert st.samples == W * H * 8 and st.segments == ost["segments"]


def test_render_too_large_is_an_error(rtw):
    """claim_job packs a queue position into 28 bits: a frame whose longest queue would hold 2^28 jobs is refused (-5) before
    anything is launched or written (the output pointer is never touched)."""
    import ctypes as C
    import torch
    from rtw_amd import _capi
    T = np.float32
    rd = rtw.DeviceRenderer(rtw.scene_2_spheres(elem_type=T), rtw.default_camera(elem_type=T), device=0)
    dummy = torch.zeros(16, dtype=torch.float32, device="cuda:0")
    L = _capi.lib()
    for w, h, ok in ((8, 1 << 27, False), (1 << 17, 1 << 17, False), (64, 36, True)):
        P = _capi.make_params(w, h, 1, 4, 1, 0, 0, 1, -1, 1, 0)
        P.job_pixels = 4
        out = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda:0") if ok else dummy
        rc = L.rtw_render_device_f32(rd.handle, C.byref(rd.cam), C.byref(P), C.c_void_p(out.data_ptr()), C.c_void_p(0))
        if ok:
            assert rc == 0
            assert rd.stats()["samples"] == w * h
        else:
            assert rc == -5 and b"too large" in L.rtw_last_error(), (w, h, rc, L.rtw_last_error())
    torch.cuda.synchronize()
    assert float(dummy.abs().sum()) == 0.0
    rd.close()
