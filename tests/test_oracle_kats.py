"""Pin the CPU oracle: every value assertion the reference's own tests hold for this path
(/root/reference/test/runtests.jl:131,135,180; src/pluto_RayTracingWeekend.jl:603-615) plus the
hand-derivable KATs of SURVEY.md A.4.  CPU only."""
import numpy as np
import pytest

F = [np.float32, np.float64]


def test_reflect_kat_exact(oracle):
    # test/runtests.jl:180  reflect(SA[0.6,-0.8,0.0], SA[0.0,1.0,0.0]) == SA[0.6,0.8,0.0]
    assert np.array_equal(oracle.reflect([0.6, -0.8, 0.0], [0.0, 1.0, 0.0], np.float64), np.array([0.6, 0.8, 0.0]))


def test_near_zero_kat(oracle):
    # test/runtests.jl:131  !near_zero(SA[0.4,0.5,0.1]); threshold is on the SQUARED length (src/vec.jl:20)
    assert not oracle.near_zero([0.4, 0.5, 0.1])
    assert oracle.near_zero([1e-3, 1e-3, 1e-3])          # 3e-6 < 1e-5
    assert not oracle.near_zero([3e-3, 1e-3, 1e-3])      # 1.1e-5


def test_refract_kats(oracle):
    # src/pluto_RayTracingWeekend.jl:603-615 (== test/runtests.jl:203-211, commented there)
    d, n = [0.6, -0.8, 0.0], [0.0, 1.0, 0.0]
    assert np.array_equal(oracle.refract(d, n, 1.0), np.array([0.6, -0.8, 0.0]))          # unchanged angle, exact
    assert np.allclose(oracle.refract(d, n, 2.0), [0.87519, -0.483779, 0.0], atol=1e-3)   # abs() under sqrt + normalize
    assert np.allclose(oracle.refract(d, n, 0.5), [0.3, -0.953939, 0.0], atol=1e-3)


@pytest.mark.parametrize("T", F)
def test_skycolor_kats(oracle, T):
    # src/ray_color.jl:1-6: down -> white, up -> (0.5, 0.7, 1.0) with the Float64 literal 0.7
    assert np.array_equal(oracle.skycolor([0, -1, 0], T), [1.0, 1.0, 1.0])
    assert np.array_equal(oracle.skycolor([0, 1, 0], T), [0.5, 0.7, 1.0])


@pytest.mark.parametrize("T", F)
def test_hit_sphere_kats(oracle, T):
    # SURVEY A.4 from src/hit.jl:12-35
    h = oracle.hit_sphere([0, 0, -1], 0.5, [0, 0, 0], [0, 0, -1], 1e-4, np.inf, T)
    assert h["t"] == 0.5 and np.array_equal(h["p"], [0, 0, -0.5]) and np.array_equal(h["n"], [0, 0, 1]) and h["front"]
    h = oracle.hit_sphere([0, 0, -1], -0.5, [0, 0, 0], [0, 0, -1], 1e-4, np.inf, T)   # negative radius flips the normal
    assert h["t"] == 0.5 and np.array_equal(h["n"], [0, 0, 1]) and not h["front"]
    assert oracle.hit_sphere([0, 0, -1], 0.5, [0, 0, 0], [0, 1, 0], 1e-4, np.inf, T) is None     # miss
    assert oracle.hit_sphere([0, 0, -1], 0.5, [0, 0, 0], [0, 0, -1], 1e-4, 0.4, T) is None       # beyond tmax
    h = oracle.hit_sphere([0, 0, 0], 1.0, [0, 0, 0], [0, 0, -1], 1e-4, np.inf, T)                # from inside: far root
    assert h["t"] == 1.0 and not h["front"]


def test_hit_world_tie_and_order(oracle):
    # src/hit.jl:38-50: `closest` shrinks; a later sphere at exactly the same t wins
    flat = dict(n=3, cx=[0, 0, 0], cy=[0, 0, 0], cz=[-3, -1, -1], r=[0.5, 0.5, 0.5], kind=[0, 0, 0],
                ar=[1, 1, 1], ag=[1, 1, 1], ab=[1, 1, 1], param=[0, 0, 0])
    idx, rec = oracle.hit_world(flat, [0, 0, 0], [0, 0, -1], 1e-4, np.inf)
    assert idx == 2 and rec[0] == 0.5
    idx, _ = oracle.hit_world(flat, [0, 0, 0], [0, 1, 0], 1e-4, np.inf)
    assert idx == -1


@pytest.mark.parametrize("T", F)
def test_reflectance_kats(oracle, T):
    # src/light.jl:19-25: r0 = ((1-eta)/(1+eta))^2 = 0.04 for eta = 1/1.5; grazing -> 1
    assert abs(float(oracle.reflectance(1.0, T(1) / T(1.5), T)) - 0.04) < 1e-6
    assert float(oracle.reflectance(0.0, T(1) / T(1.5), T)) == 1.0


@pytest.mark.parametrize("T", F)
def test_default_camera_kat(oracle, T):
    # SURVEY A.4 from src/camera.jl:18-35: default_camera(SA{T}[0,0,0])
    c = oracle.default_camera((0, 0, 0), (0, 0, -1), (0, 1, 0), 90, 16 / 9, 0, 1, T)
    assert np.array_equal(c["w"], [0, 0, 1]) and np.array_equal(c["u"], [1, 0, 0]) and np.array_equal(c["v"], [0, 1, 0])
    assert np.allclose(c["lower_left_corner"], [-16 / 9, -1, -1], rtol=1e-6)
    assert np.allclose(c["horizontal"], [32 / 9, 0, 0], rtol=1e-6) and np.array_equal(c["vertical"], [0, 2, 0])
    assert c["lens_radius"] == 0
    # get_ray(cam, 0, 0).dir == normalize(lower_left_corner); lens sample is drawn even with lens_radius 0
    st0 = oracle.rng_seed(7)
    ray, st1 = oracle.get_ray(c, 0.0, 0.0, st0, T)
    llc = c["lower_left_corner"].astype(np.float64)
    assert np.allclose(ray[3:], llc / np.linalg.norm(llc), rtol=1e-6) and np.array_equal(ray[:3], [0, 0, 0])
    assert not np.array_equal(st0, st1)


def test_rng_structure(oracle):
    # Xoroshiro128+ (RandomNumbers.jl 1.5.3 restated; unpinned vs Julia): output = x + y of the
    # pre-update state, update with constants 55/14/36; Float32/Float64 from the low 23/52 bits.
    st = oracle.rng_seed(1)
    x, y = int(st[0]), int(st[1])
    out = oracle.rng_next(st)
    M = (1 << 64) - 1
    rotl = lambda v, k: ((v << k) | (v >> (64 - k))) & M
    assert out == (x + y) & M
    s1 = x ^ y
    assert int(st[0]) == rotl(x, 55) ^ s1 ^ ((s1 << 14) & M) and int(st[1]) == rotl(s1, 36)
    st = oracle.rng_seed(1); u = oracle.rng_next(st)
    st = oracle.rng_seed(1); f = oracle.rng_float(st, np.float32)
    assert f == np.float32(((u & 0x7FFFFF) / 2 ** 23))
    st = oracle.rng_seed(1); d = oracle.rng_float(st, np.float64)
    assert d == (u & ((1 << 52) - 1)) / 2 ** 52
    # streams of distinct (pixel, chunk) differ; the same key reproduces
    assert not np.array_equal(oracle.rng_stream(1, 0, 0), oracle.rng_stream(1, 0, 1))
    assert not np.array_equal(oracle.rng_stream(1, 0, 0), oracle.rng_stream(1, 1, 0))
    assert np.array_equal(oracle.rng_stream(1, 5, 2), oracle.rng_stream(1, 5, 2))


def test_rng_provisional_fixture(oracle):
    z = np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "rng_provisional.npz"))
    st = oracle.rng_seed(1)
    assert np.array_equal(st, z["state_seed1"])
    assert [oracle.rng_next(st) for _ in range(16)] == [int(v) for v in z["u64_seed1"]]
    st = oracle.rng_seed(1)
    assert np.array_equal(np.array([oracle.rng_float(st, np.float32) for _ in range(16)], np.float32), z["f32_seed1"])
    assert np.array_equal(oracle.rng_stream(1, 12345, 3), z["stream_1_12345_3"])


def test_scatter_draw_counts(oracle):
    """Metal consumes a unit-sphere sample even with fuzz = 0; Dielectric draws one number only
    when refraction is possible (src/material.jl:31-34,47)."""
    rec = [1.0, 0, 0, 0, 0, 1, 0, 1.0]   # t, p, n=(0,1,0), front
    d = np.array([0.6, -0.8, 0.0])
    st0 = oracle.rng_seed(3)
    out, st1 = oracle.scatter(1, [0.7, 0.6, 0.5], 0.0, d, rec, st0)          # Metal, fuzz 0
    assert not np.array_equal(st0, st1)
    assert np.allclose(out[3:6], [0.6, 0.8, 0.0], atol=1e-15) and np.array_equal(out[6:9], [0.7, 0.6, 0.5])
    # Dielectric from inside at a grazing angle: total internal reflection, no random drawn
    rec_in = [1.0, 0, 0, 0, 0, 1, 0, 0.0]
    dg = np.array([0.8, -0.6, 0.0])                       # sin = 0.8, 1.5 * 0.8 > 1
    out, st2 = oracle.scatter(2, [1, 1, 1], 1.5, dg, rec_in, st0)
    assert np.array_equal(st0, st2)
    assert np.allclose(out[3:6], [0.8, 0.6, 0.0]) and np.array_equal(out[6:9], [1, 1, 1])
    # Dielectric from outside: one draw
    out, st3 = oracle.scatter(2, [1, 1, 1], 1.5, d, rec, st0)
    st_chk = st0.copy(); oracle.rng_next(st_chk)
    assert np.array_equal(st3, st_chk)


def test_height_rule(rtw):
    # src/render.jl:11-12  image_width ÷ 16//9   
    for w, h in [(96, 54), (200, 112), (320, 180), (400, 225), (1920, 1080), (3840, 2160)]:
        assert rtw.image_height(w) == h


def test_exact_fixed_point_accumulation_vs_rational_arithmetic(oracle):
    """PIXEL_STREAM pixel accumulation (oracle/rtw_oracle.c fx_add / fx_to_double): the 64.64
    fixed-point sum rounded once equals the exactly computed rational sum (each term truncated
    towards zero at 2^-64) rounded to nearest-even, for any order of the terms."""
    from fractions import Fraction
    rng = np.random.default_rng(1)
    for trial in range(300):
        n = int(rng.integers(1, 200))
        x = rng.uniform(0, 8, n) * rng.choice([1, 1, 1, 1e-3, 1e-9, -1, 2.0 ** -30, 1e5], n)
        s, bad = oracle.fx_sum(x)
        tot = Fraction(0)
        for v in x:
            q = abs(Fraction(float(v))) * 2 ** 64
            tot += Fraction(q.numerator // q.denominator, 2 ** 64) * (1 if v >= 0 else -1)
        assert bad == 0 and s == float(tot), (trial, s, float(tot))          # Fraction -> float rounds correctly
        assert oracle.fx_sum(x[::-1].copy())[0] == s and oracle.fx_sum(rng.permutation(x))[0] == s
    assert oracle.fx_sum([0.1] * 10) == (1.0, 0)                                # sequential binary64 adds give 0.9999999999999999
    assert oracle.fx_sum([2.0 ** 31 - 1, 0.5]) == (2147483647.5, 0)
    assert oracle.fx_sum([1.0, 2.0 ** -53]) == (1.0, 0) and oracle.fx_sum([1.0, 2.0 ** -53, 2.0 ** -60])[0] == 1.0 + 2.0 ** -52
    for poison in (float("nan"), float("inf"), -float("inf"), 2.0 ** 31, -2.0 ** 40):
        s, bad = oracle.fx_sum([1.0, poison])
        assert bad == 1 and np.isnan(s)
