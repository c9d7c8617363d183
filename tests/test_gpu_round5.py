"""Round-5 GPU tests (all through the C ABI):
  * the numerics modes of the ray-sphere test (include/rtw_hip.h RTW_FLAG_NUMERICS_*): the device follows the oracle in every mode
    (the parity modules run once per mode: conftest.numerics), the modes differ from each other where the oracle says they do, and the
    statistical tier T3 can SEE a bias of that size -- it fails when the contract form is compared with the reference's own order;
  * test aids of the environment are dead without the master switch RTW_ENABLE_TEST_AIDS=1;
  * the in-library device list as bench.py times it."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden
from test_gpu_render import gpu_render
from test_gpu_round2 import _random_spheres_case

pytestmark = pytest.mark.gpu


# ---- numerics modes ---------------------------------------------------------------------------------------------------------
def test_numerics_flags_select_the_oracles_modes(oracle):
    """every mode by its flag bits, against the golden of that mode; both bits at once are refused"""
    from rtw_amd import _capi
    name = "cfg2_random_320x180_64spp_d16_f32"
    seen = {}
    for mode, bits in (("reference", 0), ("contract", _capi.FLAG_NUMERICS_CONTRACT), ("reference_fma2", _capi.FLAG_NUMERICS_REFERENCE_FMA2)):
        g = load_golden(name, numerics=mode)
        for flags in (0, _capi.FLAG_SCAN_VALU, _capi.FLAG_GROUP_CULL, _capi.FLAG_GROUP_CULL | _capi.FLAG_SCAN_VALU):
            img, st = gpu_render(g, flags=flags | bits)
            assert np.array_equal(img, g["image"]) and st.segments == g["segments"], (mode, flags)
        seen[mode] = (g["image"], g["segments"])
    assert seen["reference"][1] > seen["reference_fma2"][1] > seen["contract"][1]               # tmin re-hits of the ground sphere
    assert not np.array_equal(seen["reference"][0], seen["contract"][0])
    with pytest.raises(_capi.RtwError, match="exclude each other"):
        gpu_render(load_golden(name), flags=_capi.FLAG_NUMERICS_CONTRACT | _capi.FLAG_NUMERICS_REFERENCE_FMA2)
    with pytest.raises(_capi.RtwError, match="unknown flags"):           # ABI 3's RTW_FLAG_NUMERICS_REFERENCE_FMA
        gpu_render(load_golden(name), flags=64)


def test_t3_sees_the_bias_between_numerics_modes(oracle, rtw):
    """The statistical tier (test_t3_statistical_parity_with_ref_serial) compares two unbiased estimators of one image.  The three-FMA
    contract form of rounds 1 - 4 and the reference's own evaluation order of src/hit.jl:16-18 are NOT estimators of one image in
    Float32: the contract form re-hits the r = 1000 ground sphere at tmin less often (3.13 against 3.22 segments per sample at depth
    16), and every such re-hit multiplies the path by the ground's albedo 0.5.  Same tolerance, same frame: A = reference seed 1,
    B = reference seed 2, C = contract seed 3 -- |mean(A - C)| must EXCEED 4 sd / sqrt(N) by a wide margin (it did not, T3 could not have
    caught the rounds 1 - 4 choice), while A - B passes; the sign is the oracle's (contract brighter).  In Float64 the modes pass."""
    W, H, spp, depth = 320, 180, 1024, 16
    for T, must_differ in ((np.float32, True), (np.float64, False)):
        g, cam = _random_spheres_case(rtw, oracle, T, W, spp, depth=depth)
        A, sa = gpu_render(g, gamma=0, numerics="reference")
        B, _ = gpu_render(g, gamma=0, seed=2, numerics="reference")
        Cc, sc = gpu_render(g, gamma=0, seed=3, numerics="contract")
        A, B, Cc = A.astype(np.float64), B.astype(np.float64), Cc.astype(np.float64)
        D1, D2 = A - B, A - Cc
        N = D1.size
        z1 = abs(D1.mean()) / (D1.std() / np.sqrt(N))
        z2 = abs(D2.mean()) / (D2.std() / np.sqrt(N))
        assert z1 <= 4.0, (T, z1)
        if must_differ:
            assert z2 > 12.0 and D2.mean() < 0, (z2, D2.mean())                          # reference darker than contract
            assert 1.015 < sa.segments / sc.segments < 1.05, (sa.segments, sc.segments)      # 3.22 / 3.13 at depth 16
            # the bottom half of the frame (the ground) carries it
            assert abs(D2[H // 2:].mean()) > 1.3 * abs(D2[:H // 2].mean())            # (measured: 0.0035 against 0.0021)
        else:
            assert z2 <= 4.0, (T, z2)
            assert abs(sa.segments / sc.segments - 1.0) < 2e-3


# ---- the block vote of the group cull (rtw_device.hpp CullGrid: bins + per-bin block sets, one look-up per ray and scan) ------------
def _flat_scene(T, cx, cy, cz, r, seed):
    rng = np.random.default_rng(seed)
    n = len(cx)
    kind = rng.integers(0, 3, n).astype(np.int32)
    param = np.where(kind == 2, 1.5, np.where(kind == 1, rng.uniform(0, 0.5, n), 0.0)).astype(T)
    return dict(n=n, cx=np.asarray(cx, T), cy=np.asarray(cy, T), cz=np.asarray(cz, T), r=np.asarray(r, T), kind=kind,
                ar=rng.uniform(0.2, 1, n).astype(T), ag=rng.uniform(0.2, 1, n).astype(T), ab=rng.uniform(0.2, 1, n).astype(T), param=param)


def _vote_scenes(T):
    rng = np.random.default_rng(23)
    out = {}
    # 1 100 spheres: two groups of 32 blocks, scene copy AND both groups' tables in LDS
    n = 1100
    out["two_groups_lds"] = _flat_scene(T, rng.uniform(-9, 9, n), rng.uniform(-0.3, 0.3, n), rng.uniform(-14, -2, n), rng.uniform(0.05, 0.25, n), 1)
    # a lattice whose sphere boxes end exactly on bin edges: extent 16 over 64 bins = 0.25 per bin, centres on multiples of 0.5, r = 0.25
    gx, gz = np.meshgrid(np.arange(-8, 8.01, 0.5), np.arange(-18, -2 + 0.01, 0.5))
    out["on_bin_edges"] = _flat_scene(T, gx.ravel(), np.zeros(gx.size), gz.ravel(), np.full(gx.size, 0.25), 2)
    # every small sphere in one plane and of one size (one axis of the class's box has the extent of a diameter), plus a ground sphere (huge: in-lane)
    n = 300
    out["flat_layer"] = _flat_scene(T, np.r_[rng.uniform(-6, 6, n), 0.0], np.r_[np.full(n, 0.2), -1000.0], np.r_[rng.uniform(-12, -2, n), -6.0],
                                    np.r_[np.full(n, 0.2), 1000.0], 3)
    # a sparse class: two far-apart clumps (most bins empty) and a few BIG spheres
    n = 200
    cx = np.r_[rng.uniform(-60, -58, n), rng.uniform(58, 60, n), [0.0, 3.0, -3.0]]
    cz = np.r_[rng.uniform(-40, -38, n), rng.uniform(-12, -10, n), [-8.0, -8.0, -8.0]]
    out["two_clumps"] = _flat_scene(T, cx, np.r_[rng.uniform(-1, 1, 2 * n), [0.0, 0.0, 0.0]], cz, np.r_[rng.uniform(0.1, 0.4, 2 * n), [1.5, 1.0, 1.0]], 4)
    # the BIG class (|r| > 4 x the median radius) in Float32: up to 8 of its spheres are tested by every lane itself and its block is dead;
    # 9 stay a block that every half wave visits; a huge ground sphere (>= 16 x the median) is in the list either way
    n = 160
    for nbig in (7, 8, 9, 40):
        cx = np.r_[rng.uniform(-8, 8, n), np.linspace(-7, 7, nbig), 0.0]
        cz = np.r_[rng.uniform(-14, -3, n), rng.uniform(-12, -5, nbig), -8.0]
        cy = np.r_[rng.uniform(-0.2, 0.2, n), np.full(nbig, 0.9), -400.0]
        out[f"big_class_{nbig}_plus_ground"] = _flat_scene(T, cx, cy, cz, np.r_[rng.uniform(0.08, 0.2, n), rng.uniform(0.8, 1.1, nbig), 399.0], 40 + nbig)
    # all spheres concentric (zero extent of the centres; boxes differ by the radii only)
    n = 40
    out["concentric"] = _flat_scene(T, np.zeros(n), np.zeros(n), np.full(n, -6.0), np.linspace(0.5, 2.5, n), 5)
    return out


@pytest.mark.parametrize("T", [np.float32, np.float64])
def test_block_vote_never_skips_a_block_a_ray_can_hit(oracle, rtw, T):
    """group cull (flags 1: table vote on the matrix pipe) == plain scan == oracle on scenes built against the vote: block sets of two
    groups, boxes ending on bin edges, a flat class, mostly empty bins, concentric spheres; wide and narrow cameras"""
    g0 = load_golden("cfg1_2spheres_96x54_16spp_d4_f32")
    cam_wide = {k: np.asarray(v).astype(T) for k, v in g0["cam"].items()}
    scenes = {k: (v, cam_wide) for k, v in _vote_scenes(T).items()}
    # a class far from the origin (the bin offset -glo * inv is then rounded by whole fractions of a bin: host and device must agree on it)
    rng = np.random.default_rng(29)
    n, off = 500, np.array([2000.0, -1000.0, 3000.0])
    far = _flat_scene(T, off[0] + rng.uniform(-6, 6, n), off[1] + rng.uniform(-0.3, 0.3, n), off[2] + rng.uniform(-6, 6, n), rng.uniform(0.1, 0.3, n), 6)
    cam = rtw.default_camera(tuple(off + [0.5, 2.0, 14.0]), tuple(off), (0, 1, 0), 50, 16 / 9, 0.0, 14.0, elem_type=T)
    scenes["far_from_origin"] = (far, {k: getattr(cam, k) for k in oracle.CAM_FIELDS + ("lens_radius",)})
    for name, (flat, cam_d) in scenes.items():
        g = dict(g0, flat=flat, cam=cam_d, image=np.zeros((1, 1, 3), T))
        ref, ost = oracle.render(flat, g["cam"], 96, 54, 4, T=T, max_depth=12, seed=5, n_chunks=2)
        for flags in (1, 0, 5):
            img, st = gpu_render(g, width=96, height=54, spp=4, n_chunks=2, max_depth=12, seed=5, flags=flags)
            assert np.array_equal(img, ref) and st.segments == ost["segments"], (name, flags)
        assert np.unique(ref.reshape(-1, 3), axis=0).shape[0] > 50, name       # the spheres are in view


# ---- test aids need the master switch ------------------------------------------------------------------------------------------
_AID_PROBE = r"""
import json, sys
import numpy as np
import torch
torch.cuda.init()
import rtw_amd as R
T = np.float32
img = R.render(R.scene_4_spheres(elem_type=T), R.t_default_cam(elem_type=T), 640, 16, depth=4, seed=3)
st = R.last_stats()
import hashlib
print(json.dumps({"sha": hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest(), "grid": st["grid_blocks"], "block": st["block_threads"]}))
"""


def _aid_probe(env_extra):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RTW_ENABLE_TEST_AIDS", "RTW_SCAN", "RTW_POOL", "RTW_JOB_PIXELS"):
        env.pop(k, None)
    env.update(env_extra)
    p = subprocess.run([sys.executable, "-c", _AID_PROBE], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])


def test_stray_environment_switches_change_nothing():
    """RTW_SCAN=valu / RTW_POOL=1 / RTW_JOB_PIXELS=1 in a caller's environment: without RTW_ENABLE_TEST_AIDS=1 the launch geometry of
    a default render is what rtw_params alone gives; with the master switch the aids act (and the image stays the same)."""
    base = _aid_probe({})
    stray = _aid_probe({"RTW_SCAN": "valu", "RTW_POOL": "1", "RTW_JOB_PIXELS": "1", "RTW_NO_HUGE": "1", "RTW_PHASE_PROFILE": "1"})
    assert stray == base
    valu = _aid_probe({"RTW_ENABLE_TEST_AIDS": "1", "RTW_SCAN": "valu"})
    assert valu["sha"] == base["sha"] and valu["grid"] != base["grid"]                  # the all-VALU kernel runs 7 waves per SIMD, not 5
    pool = _aid_probe({"RTW_ENABLE_TEST_AIDS": "1", "RTW_POOL": "1"})               # (the ray-pool kernel is a `make POOL=1` build option: the default library ignores the aid)
    assert pool["sha"] == base["sha"] and pool["block"] == (1024 if os.environ.get("RTW_TEST_POOL") == "1" else 256)


# ---- the in-library device list as bench.py times it --------------------------------------------------------------------------
def test_bench_in_library_devices_leg():
    """`python bench.py --in-library-devices N` (tools/gpu_multi.sh calls nothing else for this path): host-buffer entry point with a device
    list, peer gather and the in-library RCCL reduce, per-device kernel times, gather path, frame equal to the one-device frame.  On a
    one-GPU box the ordinals repeat and the leg says so (`emulation`)."""
    from test_gpu_round3 import _bench
    import torch
    have = torch.cuda.device_count()
    n = max(2, have)
    p, d = _bench(["--in-library-devices", str(n), "--steps", "1", "--spp", "8", "--width", "640"], timeout=600)
    assert p.returncode == 0 and d is not None, p.stderr[-3000:]
    il = d["in_library_devices"]
    assert il["n_devices"] == n and il["emulation"] == (n > have)
    for k in ("peer", "rccl_reduce"):
        assert il[k]["frame_sha256_equal"] is True and il[k]["ms"] > 0 and il[k]["kernel_ms_max"] > 0, (k, il[k])
    assert len(il["peer"]["per_device_kernel_ms"]) == n
    assert il["rccl_reduce"]["gather_path_names"] == ["rccl_reduce"]
    if have > 1:
        assert il["peer"]["gather_path_names"] in (["peer"], ["host_staged"], ["peer", "same_device"])
    else:
        assert il["peer"]["gather_path_names"] == ["same_device"]
