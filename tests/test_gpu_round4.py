"""Round-4 GPU tests (all through the C ABI):
  * (the ray-pool kernel's tests moved to tests/test_gpu_pool.py in round 6: the kernel is a `make POOL=1` build option now);
  * Float64 at full scale: configs[4]'s geometry (3840x2160, depth 50) in all three scan modes, and against the live oracle;
  * the soak slice on a FIXED seed list (reproducible)."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN_CASES, ROOT, all_numerics, load_golden
from test_gpu_render import gpu_render
from test_gpu_round2 import _random_spheres_case

pytestmark = pytest.mark.gpu

FLAG_CULL, FLAG_COMPACT, FLAG_VALU, FLAG_POOL = 1, 2, 4, 8


# ---- Float64 at full scale --------------------------------------------------------------------------------------------------
@all_numerics
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


@all_numerics
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


@all_numerics
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


# ---- several devices behind the C ABI -----------------------------------------------------------------------------------------
_PROBE = r"""
import hashlib, json, sys
import numpy as np
import torch
torch.cuda.init()
import rtw_amd as R
spec = json.loads(sys.argv[1])
T = np.float64 if spec.pop("__dtype__", "f32") == "f64" else np.float32
R.reseed()
scene = R.scene_random_spheres(elem_type=T)
cam = R.t_cam1(elem_type=T)
out = {}
for name, kw in spec.items():
    img = R.render(scene, cam, 200, 12, depth=16, seed=5, **kw)
    st = R.last_stats()
    out[name] = {"sha": hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest(), "gather_path": st["gather_path"], "segments": st["segments"],
                 "per_device": st["per_device"]}
print(json.dumps(out))
"""


def _probe(spec, env_extra=None):
    import json
    import subprocess
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, "-c", _PROBE, json.dumps(spec)], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_gather_branches_on_one_device(dtype):
    """The branches of the in-library multi-device gather that a one-GPU box cannot reach by itself, forced by the test aids of
    the library (RTW_ENABLE_TEST_AIDS=1): RTW_DEBUG_REMOTE_SHARDS=1 makes every shard but the first render into its own buffer and COPY it into the gather
    buffer (hipMemcpyPeerAsync, the cross-device branch); with RTW_DEBUG_NO_PEER=1 the copy takes the host-staged fallback (pinned
    staging, D2H + H2D) that a platform without peer access gets; and the RCCL reduce (one rank: a communicator holds a GPU once).  Same
    frame as one device in both precisions, rtw_stats_t.gather_path says which ran, rtw_stats_devices lists every shard."""
    aids = {"RTW_ENABLE_TEST_AIDS": "1"}
    one = _probe({"__dtype__": dtype, "one": {}})["one"]
    assert one["per_device"] == [[0, one["per_device"][0][1]]] and one["per_device"][0][1] > 0
    same = _probe({"__dtype__": dtype, "x": {"devices": [0, 0, 0]}})["x"]
    assert same["sha"] == one["sha"] and same["gather_path"] == 8 and same["segments"] == one["segments"]          # RTW_GATHER_SAME_DEVICE
    assert [d for d, _ in same["per_device"]] == [0, 0, 0] and all(ms > 0 for _, ms in same["per_device"])
    peer = _probe({"__dtype__": dtype, "x": {"devices": [0, 0, 0]}}, dict(aids, RTW_DEBUG_REMOTE_SHARDS="1"))["x"]
    assert peer["sha"] == one["sha"] and peer["gather_path"] == 1                                                # RTW_GATHER_PEER
    staged = _probe({"__dtype__": dtype, "x": {"devices": [0, 0, 0, 0, 0]}}, dict(aids, RTW_DEBUG_REMOTE_SHARDS="1", RTW_DEBUG_NO_PEER="1"))["x"]
    assert staged["sha"] == one["sha"] and staged["gather_path"] == 2                                            # RTW_GATHER_HOST_STAGED
    rccl = _probe({"__dtype__": dtype, "x": {"devices": [0], "rccl_reduce": True}})["x"]
    assert rccl["sha"] == one["sha"] and rccl["gather_path"] == 4                                                # RTW_GATHER_RCCL
    # without the master switch the aids are dead: the same-device path again
    dead = _probe({"__dtype__": dtype, "x": {"devices": [0, 0, 0]}}, {"RTW_DEBUG_REMOTE_SHARDS": "1"})["x"]
    assert dead["gather_path"] == 8 and dead["sha"] == one["sha"]


def test_in_library_rccl_reduce_one_rank():
    """RTW_FLAG_RCCL_REDUCE with a device list of one: librccl is loaded on demand, ncclCommInitAll + ncclReduce run behind the C
    ABI (no torch.distributed), the frame is the one-device frame.  (A communicator cannot hold the same GPU twice, so more than
    one rank needs more than one GPU: test_multi_gpu_* below.)"""
    r = _probe({"one": {}, "rccl": {"devices": [0], "rccl_reduce": True}})
    assert r["rccl"]["sha"] == r["one"]["sha"] and r["rccl"]["gather_path"] == 4 and r["rccl"]["segments"] == r["one"]["segments"]
    from rtw_amd._capi import RtwError
    import rtw_amd as R
    with pytest.raises(RtwError, match="distinct devices"):
        R.render(R.scene_2_spheres(elem_type=np.float32), R.t_default_cam(), 96, 1, devices=[0, 0], rccl_reduce=True)


def _n_gpus():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


@pytest.mark.skipif(_n_gpus() < 2, reason="needs >= 2 MI355X (the driver's multi-GPU tier); one-GPU boxes cover the same code through the test aids above")
def test_multi_gpu_in_library_paths_match_one_device():
    """N physical devices behind the C ABI: compact shards gathered with peer copies (peer access must be ON: gather_path ==
    RTW_GATHER_PEER) and the RCCL reduce of zero-padded frames -- both the one-device frame, bit for bit."""
    n = _n_gpus()
    r = _probe({"one": {}, "peer": {"devices": list(range(n))}, "all": {"devices": "all"}, "rccl": {"devices": list(range(n)), "rccl_reduce": True}})
    assert r["peer"]["sha"] == r["one"]["sha"] == r["all"]["sha"] == r["rccl"]["sha"]
    assert r["peer"]["gather_path"] in (1, 2), r["peer"]             # peer copies (or the documented fallback where the platform refuses peer access)
    assert r["rccl"]["gather_path"] == 4
    assert r["peer"]["segments"] == r["one"]["segments"] == r["rccl"]["segments"]


@pytest.mark.skipif(_n_gpus() < 2, reason="needs >= 2 MI355X: bench.py --gpus 2 over the real nccl (RCCL) backend")
@pytest.mark.parametrize("collective", ["reduce", "gather"])
def test_multi_gpu_bench_rccl_matches_one_rank(collective):
    from test_gpu_round3 import SMALL, _bench
    p1, one = _bench(["--gpus", "1"] + SMALL)
    assert p1.returncode == 0, p1.stderr[-2000:]
    p2, two = _bench(["--gpus", "2", "--collective", collective] + SMALL)
    assert p2.returncode == 0, p2.stderr[-3000:]
    assert two["backend"] == "nccl" and two["world_size_observed"] == 2 and two["one_device_emulation"] is False
    assert two["frame_sha256"] == one["frame_sha256"]
    assert two["collective_ms"] > 0 and two["render_ms_max"] >= two["render_ms_min"] > 0
    assert sorted(r["device"] for r in two["per_rank"]) == [0, 1]


# ---- huge spheres tested in-lane (DevScene::huge) -------------------------------------------------------------------------------
@all_numerics
@pytest.mark.parametrize("T", [np.float32, np.float64])
@pytest.mark.parametrize("layout", ["one", "two_coincident", "two_nested_negative", "three", "first_and_last"])
def test_huge_spheres_in_lane(oracle, T, layout):
    """Spheres of radius >= 16 x the median (at most two: a ground sphere) are tested exactly by every lane instead of through the
    matrix-pipe filter, their filter rows disabled.  The closest hit and the tie rule (the LATER sphere wins an exact tie, src/hit.jl:38-50)
    must come out as the oracle's: one huge sphere, two coincident ones (every hit is a tie between them), a huge sphere inside a
    hollow (negative-radius) one, three (the third stays in the filter), huge spheres at both ends of the list."""
    from test_gpu_round2 import _stress_rays
    from test_gpu_units import run_unit
    rng = np.random.default_rng({"one": 1, "two_coincident": 2, "two_nested_negative": 3, "three": 4, "first_and_last": 5}[layout])
    n = 70
    cx, cy, cz = [rng.uniform(-8, 8, n) for _ in range(3)]
    r = rng.uniform(0.1, 0.4, n)
    big = {"one": [(30, 0, -300.0, 0, 300.0)],
           "two_coincident": [(10, 0, -300.0, 0, 300.0), (50, 0, -300.0, 0, 300.0)],
           "two_nested_negative": [(5, 0, 0, 0, 40.0), (6, 0, 0, 0, -60.0)],
           "three": [(0, 0, -300.0, 0, 300.0), (33, 0, 0, 400.0, 380.0), (69, 100.0, 0, 0, 90.0)],
           "first_and_last": [(0, 0, -300.0, 0, 300.0), (69, 0, 0, 0, 50.0)]}[layout]
    for (i, x, y, z, rr) in big:
        cx[i], cy[i], cz[i], r[i] = x, y, z, rr
    flat = dict(n=n, cx=cx.astype(T), cy=cy.astype(T), cz=cz.astype(T), r=r.astype(T), kind=np.zeros(n, np.int32),
                ar=np.ones(n, T), ag=np.ones(n, T), ab=np.ones(n, T), param=np.zeros(n, T))
    m = 65536
    rays = _stress_rays(rng, flat, m, T, 8.0)
    tmin = T(1e-4)
    ref_idx, ref_t = oracle.hit_world_batch(flat, rays, tmin, np.inf, T)
    x = np.concatenate([rays.astype(np.float64), np.full((m, 1), float(tmin)), np.full((m, 1), np.inf)], 1)
    for op in (13, 14):                         # the matrix-pipe scan (huge spheres in-lane) and its culling form (all spheres through the filter)
        y = run_unit(op, x, 9, T, flat=flat)
        bad = (y[:, 0].astype(np.int64) != ref_idx) | ((ref_idx >= 0) & (y[:, 1] != ref_t.astype(np.float64)))
        assert not bad.any(), (op, int(bad.sum()), np.flatnonzero(bad)[:5])
    if layout == "two_coincident":
        hits = ref_idx[np.isin(ref_idx, [10, 50])]
        assert hits.size > 1000 and (hits == 50).all()          # every hit of the pair is an exact tie: the later one


def test_huge_sphere_path_off_gives_the_same_frame():
    """RTW_NO_HUGE=1 (A/B aid, read at scene upload; like every aid only under RTW_ENABLE_TEST_AIDS=1) sends every sphere through the filter again: same frame, same counters"""
    on = _probe({"x": {}})["x"]
    off = _probe({"x": {}}, {"RTW_ENABLE_TEST_AIDS": "1", "RTW_NO_HUGE": "1"})["x"]
    assert on["sha"] == off["sha"] and on["segments"] == off["segments"]


# ---- the bench line -----------------------------------------------------------------------------------------------------------
def test_bench_line_contract_and_live_counters():
    """`python bench.py` (small spp): every key of the bench contract, the roofline object in the fixed yard-sticks, HBM traffic and issue-busy
    MEASURED by the run itself (rocprofv3 --pmc passes it spawns; `traffic_static` false), the ray-pool leg with the same frame hash, the
    end-to-end value next to the device-resident one, and the reference's published configuration as its own leg."""
    from test_gpu_round3 import _bench
    p, d = _bench(["--steps", "1", "--warmup", "1", "--spp", "16", "--cpu-seconds", "0.5"], timeout=900)
    assert p.returncode == 0 and d is not None, p.stderr[-3000:]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["dtype"] == "f32" and d["vs_baseline"] is None and "workload" in d["config"] and d["higher_is_better"] is True
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["peak"] == 2500.0 and abs(r["frac"] - r["achieved"] / 2500.0) < 1e-3
    assert r["algorithmic"]["fp32_vector_peak"] == 157.3 and r["issue_model"]["model"] is True
    assert r["traffic_static"] is False and r["traffic"] > 0 and r["traffic_detail"]["write_size_bytes"] >= 1920 * 1080 * 3 * 4 * 0.9
    assert 0.0 < r["issue_busy"]["sum"] < 1.2 and r["issue_busy"]["mfma_instructions"] > 0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port"
    assert d["value_end_to_end"] and d["end_to_end"]["value"] == d["value_end_to_end"] and d["kernel_only"] >= d["value"] * 0.9
    assert d["ray_pool"] is None                                  # (round 6: the ray-pool kernel is a `make POOL=1` build option, not in the default library)
    assert d["accelerated"]["frame_sha256_equal"] is True and d["scan_valu"]["frame_sha256_equal"] is True
    assert d["collective_ms"] == 0.0 and d["render_ms_max"] == d["render_ms_min"] > 0
    # (the f64 legs belong to the headline workload only: --spp 16 is not it)
    assert d["f64_4k"] is None and d["f64_1080p_d16"] is None
    # round 5: the numerics mode and its counter in the line, the other modes as legs with their own (different) frames, the in-library device list
    assert d["numerics"] == "reference" and d["config"]["numerics"].startswith("reference") and 3.0 < d["segments_per_sample"] < 5.0
    nl = d["numerics_legs"]
    assert set(nl) == {"contract", "reference_fma2"} and nl["contract"]["frame_sha256"] != d["frame_sha256"]
    assert nl["contract"]["segments_per_sample"] < nl["reference_fma2"]["segments_per_sample"] < d["segments_per_sample"]
    il = d["in_library_devices"]
    assert il["n_devices"] >= 2 and il["peer"]["frame_sha256_equal"] is True and il["rccl_reduce"]["frame_sha256_equal"] is True
    assert len(il["peer"]["per_device_kernel_ms"]) == il["n_devices"] and il["rccl_reduce"]["gather_path_names"] == ["rccl_reduce"]
