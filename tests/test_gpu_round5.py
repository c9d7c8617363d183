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
    for mode, bits in (("reference", 0), ("contract", _capi.FLAG_NUMERICS_CONTRACT), ("reference_fma", _capi.FLAG_NUMERICS_REFERENCE_FMA),
                       ("reference_fma2", _capi.FLAG_NUMERICS_REFERENCE_FMA2)):
        g = load_golden(name, numerics=mode)
        for flags in (0, _capi.FLAG_SCAN_VALU, _capi.FLAG_GROUP_CULL, _capi.FLAG_GROUP_CULL | _capi.FLAG_SCAN_VALU, _capi.FLAG_RAY_POOL):
            img, st = gpu_render(g, flags=flags | bits)
            assert np.array_equal(img, g["image"]) and st.segments == g["segments"], (mode, flags)
        seen[mode] = (g["image"], g["segments"])
    assert seen["reference"][1] > seen["reference_fma"][1] > seen["contract"][1] and seen["reference_fma2"][1] > seen["contract"][1]               # tmin re-hits of the ground sphere
    assert not np.array_equal(seen["reference"][0], seen["contract"][0])
    with pytest.raises(_capi.RtwError, match="exclude each other"):
        gpu_render(load_golden(name), flags=_capi.FLAG_NUMERICS_CONTRACT | _capi.FLAG_NUMERICS_REFERENCE_FMA)


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
    pool = _aid_probe({"RTW_ENABLE_TEST_AIDS": "1", "RTW_POOL": "1"})
    assert pool["sha"] == base["sha"] and pool["block"] == 1024


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
