"""Tier T0 on the GPU: every device function of the hot path, through the C ABI
(rtw_unit_f32/_f64), against the CPU oracle on the same explicit inputs -- bit-exact.

Slot layouts (8-byte slots: doubles for reals, raw uint64 for RNG state words), per item:
  op 0 hit_sphere  in  c[3] r o[3] d[3] tmin tmax          out hit t p[3] n[3] front
  op 1 reflect     in  v[3] n[3]                           out r[3]
  op 2 refract     in  d[3] n[3] ratio                     out r[3]
  op 3 reflectance in  cos ratio                           out R
  op 4 scatter     in  state[2] kind albedo[3] param d[3] rec{t p[3] n[3] front}
                   out state[2] o[3] d[3] att[3]
  op 5 get_ray     in  state[2] s t                        out state[2] o[3] d[3]
  op 6 skycolor    in  d[3]                                out c[3] (binary64)
  op 7 rng         in  state[2]                            out state[2] u[4]
  op 8 hit_world   in  o[3] d[3] tmin tmax                 out idx t p[3] n[3] front
  op 9 ray_color   in  state[2] o[3] d[3] depth            out state[2] c[3] segments
"""
import ctypes as C

import numpy as np
import pytest

from conftest import CamObj, load_golden

# every test once per numerics mode of the ray-sphere test (conftest.numerics): oracle and device switch together
pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("numerics")]
F = [np.float32, np.float64]


def run_unit(op, x, n_out, T, flat=None, cam=None):
    from rtw_amd import _capi
    L = _capi.lib()
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.zeros((x.shape[0], n_out), np.float64)
    S = keep = Cm = None
    if flat is not None:
        S, keep = _capi.make_scene(flat, T)
    if cam is not None:
        Cm = _capi.make_camera(cam, T)
    fn = L.rtw_unit_f64 if T is np.float64 else L.rtw_unit_f32
    _capi.check(fn(op | _capi.numerics_unit_bits(), x.shape[0], x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p),
                   C.byref(S) if S is not None else None, C.byref(Cm) if Cm is not None else None))
    return y


def unit_dirs(rng, n, T):
    v = rng.normal(size=(n, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    return v.astype(T).astype(np.float64)


def states(n, seed=11):
    import rtw_oracle as O
    return np.stack([O.rng_stream(seed, i, 0) for i in range(n)])


@pytest.mark.parametrize("T", F)
def test_rng(oracle, T):
    st = states(64)
    y = run_unit(7, st.view(np.float64), 6, T)
    for i in range(64):
        s = st[i].copy()
        u = [float(oracle.rng_float(s, T)) for _ in range(4)]
        assert list(y[i, 2:6]) == u
        assert np.array_equal(y[i, :2].view(np.uint64), s)


@pytest.mark.parametrize("T", F)
def test_hit_sphere(oracle, T):
    rng = np.random.default_rng(1)
    n = 512
    c = rng.uniform(-3, 3, (n, 3)).astype(T).astype(np.float64)
    r = (rng.uniform(0.1, 2.0, n) * rng.choice([1, 1, 1, -1], n)).astype(T).astype(np.float64)
    o = rng.uniform(-3, 3, (n, 3)).astype(T).astype(np.float64)
    d = unit_dirs(rng, n, T)
    o[:64] = c[:64]                                           # origin at the centre (inside)
    tmax = np.where(rng.random(n) < 0.3, rng.uniform(0.5, 4, n), np.inf).astype(T).astype(np.float64)
    x = np.concatenate([c, r[:, None], o, d, np.full((n, 1), float(T(1e-4))), tmax[:, None]], axis=1)
    y = run_unit(0, x, 9, T)
    hits = 0
    for i in range(n):
        h = oracle.hit_sphere(c[i], r[i], o[i], d[i], T(1e-4), tmax[i], T)
        if h is None:
            assert y[i, 0] == 0
        else:
            hits += 1
            assert y[i, 0] == 1 and y[i, 1] == h["t"] and np.array_equal(y[i, 2:5], h["p"])
            assert np.array_equal(y[i, 5:8], h["n"]) and bool(y[i, 8]) == h["front"]
    assert hits > 50
    # the ground sphere of scene_random_spheres (r = 1000): catastrophic-cancellation regime
    x = np.array([[0, -1000, -1, 1000, 13, 2, 3, *unit_dirs(rng, 1, T)[0] * [1, -1, 1], float(T(1e-4)), np.inf]])
    x[0, 8] = -abs(x[0, 8])
    y = run_unit(0, x, 9, T)
    h = oracle.hit_sphere(x[0, :3], 1000, x[0, 4:7], x[0, 7:10], T(1e-4), np.inf, T)
    assert (h is None and y[0, 0] == 0) or (y[0, 1] == h["t"] and np.array_equal(y[0, 5:8], h["n"]))


@pytest.mark.parametrize("T", F)
def test_light_transport(oracle, T):
    rng = np.random.default_rng(2)
    n = 256
    v, nn = unit_dirs(rng, n, T), unit_dirs(rng, n, T)
    ratio = rng.choice([1 / 1.5, 1.5, 1.0, 2.0, 0.5], n).astype(T).astype(np.float64)
    y1 = run_unit(1, np.concatenate([v, nn], 1), 3, T)
    y2 = run_unit(2, np.concatenate([v, nn, ratio[:, None]], 1), 3, T)
    cos = rng.uniform(0, 1, n).astype(T).astype(np.float64)
    y3 = run_unit(3, np.stack([cos, ratio], 1), 1, T)
    y6 = run_unit(6, v, 3, T)
    for i in range(n):
        assert np.array_equal(y1[i], oracle.reflect(v[i], nn[i], T))
        assert np.array_equal(y2[i], oracle.refract(v[i], nn[i], ratio[i], T))
        assert y3[i, 0] == oracle.reflectance(cos[i], ratio[i], T)
        assert np.array_equal(y6[i], oracle.skycolor(v[i], T))
    # the reference's own KATs, on the device (test/runtests.jl:180; pluto...:603-615)
    k = np.array([[0.6, -0.8, 0.0, 0.0, 1.0, 0.0]])
    assert np.array_equal(run_unit(1, k, 3, np.float64)[0], [0.6, 0.8, 0.0])
    assert np.array_equal(run_unit(2, np.append(k, [[1.0]], 1), 3, np.float64)[0], [0.6, -0.8, 0.0])
    assert np.allclose(run_unit(2, np.append(k, [[2.0]], 1), 3, np.float64)[0], [0.87519, -0.483779, 0.0], atol=1e-3)
    assert np.allclose(run_unit(2, np.append(k, [[0.5]], 1), 3, np.float64)[0], [0.3, -0.953939, 0.0], atol=1e-3)


@pytest.mark.parametrize("T", F)
def test_scatter(oracle, T):
    rng = np.random.default_rng(3)
    n = 384
    st = states(n, 5)
    kind = rng.integers(0, 3, n)
    albedo = rng.uniform(0, 1, (n, 3)).astype(T).astype(np.float64)
    param = np.where(kind == 1, rng.uniform(0, 5, n), 1.5).astype(T).astype(np.float64)
    param[kind == 1][:8] = 0.0
    d, nrm = unit_dirs(rng, n, T), unit_dirs(rng, n, T)
    flip = np.sum(d * nrm, 1) > 0
    nrm[flip] *= -1                                            # the hit normal faces the incoming ray
    front = rng.integers(0, 2, n).astype(np.float64)
    p = rng.uniform(-2, 2, (n, 3)).astype(T).astype(np.float64)
    rec = np.concatenate([np.ones((n, 1)), p, nrm, front[:, None]], 1)
    x = np.concatenate([st.view(np.float64), kind[:, None].astype(np.float64), albedo, param[:, None], d, rec], 1)
    y = run_unit(4, x, 11, T)
    for i in range(n):
        out, s1 = oracle.scatter(kind[i], albedo[i], param[i], d[i], rec[i], st[i], T)
        assert np.array_equal(y[i, 2:11], out.astype(np.float64)), (i, kind[i])
        assert np.array_equal(y[i, :2].view(np.uint64), s1)


@pytest.mark.parametrize("T", F)
def test_get_ray(oracle, rtw, T):
    for cam in (rtw.t_cam1(elem_type=T), rtw.t_cam2(elem_type=T), rtw.t_default_cam(elem_type=T)):
        rng = np.random.default_rng(4)
        n = 128
        st = states(n, 9)
        s, t = rng.uniform(0, 1, n).astype(T).astype(np.float64), rng.uniform(0, 1, n).astype(T).astype(np.float64)
        y = run_unit(5, np.concatenate([st.view(np.float64), s[:, None], t[:, None]], 1), 8, T, cam=cam)
        for i in range(n):
            out, s1 = oracle.get_ray(cam, s[i], t[i], st[i], T)
            assert np.array_equal(y[i, 2:8], out.astype(np.float64))
            assert np.array_equal(y[i, :2].view(np.uint64), s1)


@pytest.mark.parametrize("name", ["cfg2_random_320x180_64spp_d16_f32", "random_64x36_8spp_d50_f64",
                                  "diel_bubble_96x54_8spp_d16_f32"])
def test_hit_world_and_ray_color(oracle, name):
    g = load_golden(name)
    T = g["image"].dtype.type
    flat, cam = g["flat"], CamObj(g["cam"])
    rng = np.random.default_rng(6)
    n = 256
    st = states(n, 13)
    # camera rays through random image positions
    rays = np.zeros((n, 6))
    for i in range(n):
        r, _ = oracle.get_ray(g["cam"], rng.uniform(0, 1), rng.uniform(0, 1), st[i], T)
        rays[i] = r
    x = np.concatenate([rays, np.full((n, 1), float(T(1e-4))), np.full((n, 1), np.inf)], 1)
    y = run_unit(8, x, 9, T, flat=flat)
    nhit = 0
    for i in range(n):
        idx, rec = oracle.hit_world(flat, rays[i, :3], rays[i, 3:], T(1e-4), np.inf, T)
        assert int(y[i, 0]) == idx
        if idx >= 0:
            nhit += 1
            assert np.array_equal(y[i, 1:9], rec.astype(np.float64))
    assert nhit > 20
    x = np.concatenate([st.view(np.float64), rays, np.full((n, 1), float(g["depth"]))], 1)
    y = run_unit(9, x, 6, T, flat=flat)
    for i in range(n):
        col, s1 = oracle.ray_color(flat, rays[i, :3], rays[i, 3:], g["depth"], st[i], oracle.PRODUCT_FORWARD, T)
        assert np.array_equal(y[i, 2:5], col), i
        assert np.array_equal(y[i, :2].view(np.uint64), s1)
