"""The roofline yard-sticks of bench.py are FIXED since round 4 (the reported `frac` fell for three rounds while the kernel got
2.4x faster, because the peak was re-defined with every reformulation).  This test pins the definitions: whoever changes one of them has to
change this file too.  No GPU needed: roofline_of() is plain arithmetic on the counted tests and the kernel time."""
import os
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _wl(dtype="f32", W=1920, H=1080, spp=1000, depth=50, ray_pool=False):
    args = types.SimpleNamespace(ray_pool=ray_pool)
    return types.SimpleNamespace(W=W, H=H, spp=spp, depth=depth, dtype=dtype, n_spheres=485, c=types.SimpleNamespace(args=args))


TESTS = 3966283173735            # 8.178e9 segments x 485 spheres: BASELINE configs[2]


def test_guide_peaks_do_not_move():
    assert bench.MFMA_F16_PEAK_TFLOPS == 2500.0 and bench.VALU_PEAK_TFLOPS == {"f32": 157.3, "f64": 78.6}
    assert bench.FLOP_PER_TEST == 17 and bench.MFMA_FLOP_PER_TEST == 64 and bench.HBM_PEAK_GBS == 8000.0
    assert bench.ALIGNBIT_CYCLES == {"alignbit_2_cycles": 2.0, "alignbit_4p3_cycles": 4.3}
    assert bench.PUBLISHED_MSAMPLES == 1.617


@pytest.mark.parametrize("k_ms,frac,fp32v", [(882.826, None, 0.4856), (446.049, 0.2276, 0.9610), (364.478, 0.2786, 1.1761)])
def test_rounds_1_to_3_re_expressed(k_ms, frac, fp32v):
    """the driver-measured kernel times of rounds 1-3 (BENCH_r01..r03.json) in the fixed yard-sticks"""
    valu = frac is None                         # round 1 ran the scan on the vector ALUs
    r = bench.roofline_of(_wl(), k_ms / 1e3, TESTS, 3.9438, cull=False, valu=valu, world=1, shard_div=1)
    assert r["algorithmic"]["frac_fp32_vector"] == pytest.approx(fp32v, abs=2e-4)
    if valu:
        assert r["bound"] == "valu_fp32" and r["peak"] == 157.3 and r["frac"] == pytest.approx(fp32v, abs=2e-4) and r["issue_model"] is None
    else:
        assert r["bound"] == "mfma" and r["peak"] == 2500.0 and r["unit"] == "TFLOP/s"
        assert r["achieved"] == pytest.approx(TESTS * 64 / (k_ms / 1e3) / 1e12, rel=1e-4)
        assert r["frac"] == pytest.approx(frac, abs=2e-4) == pytest.approx(r["achieved"] / 2500.0, abs=2e-4)
        m = r["issue_model"]
        assert m["model"] is True
        assert m["alignbit_2_cycles"]["peak_algorithmic_TFLOPs"] == pytest.approx(445.6, abs=0.1)
        assert m["alignbit_4p3_cycles"]["peak_algorithmic_TFLOPs"] == pytest.approx(322.2, abs=0.1)
        assert m["alignbit_2_cycles"]["frac"] == pytest.approx(r["algorithmic"]["achieved"] / 445.6, abs=2e-3)


def test_required_roofline_keys_and_cull_mode():
    r = bench.roofline_of(_wl(), 0.36, TESTS, 3.9438, cull=False, valu=False, world=1, shard_div=1)
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r
    assert r["hbm"]["algorithmic_bytes"] == 1920 * 1080 * 3 * 4 + 485 * 12 * 4 and r["hbm"]["peak_GBs"] == 8000.0
    c = bench.roofline_of(_wl(), 0.34, TESTS, 3.9438, cull=True, valu=False, world=1, shard_div=1)
    assert c["achieved"] is None and c["frac"] is None and c["algorithmic"] is None        # tests are skipped: no roofline fraction
    f = bench.roofline_of(_wl("f64", 3840, 2160), 1.13, 4 * TESTS, 2.71, cull=False, valu=False, world=1, shard_div=1)
    assert f["kernel"] == "rtw::trace_kernel<double>" and f["hbm"]["algorithmic_bytes"] == 3840 * 2160 * 3 * 8 + 485 * 12 * 8
