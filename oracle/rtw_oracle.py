"""ctypes binding of the CPU oracle (oracle/librtw_oracle.so).

TEST INFRASTRUCTURE ONLY (see rtw_oracle.h): imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librtw_oracle.so")

PIXEL_STREAM, REF_SERIAL = 0, 1
PRODUCT_REFERENCE, PRODUCT_FORWARD = 0, 1
# the deciding arithmetic of hit(::Sphere), src/hit.jl:16-18 (rtw_oracle.h "NUMERICS MODES")
NUMERICS_REFERENCE, NUMERICS_CONTRACT, NUMERICS_REFERENCE_FMA, NUMERICS_REFERENCE_FMA2 = 0, 1, 2, 3
NUMERICS = {"reference": 0, "contract": 1, "reference_fma": 2, "reference_fma2": 3}
_default_numerics = NUMERICS_REFERENCE


def numerics_code(n):
    """'reference' / 'contract' / 'reference_fma' / 'reference_fma2', a code 0..3, or None (= the current default)"""
    if n is None:
        return _default_numerics
    if isinstance(n, str):
        return NUMERICS[n]
    n = int(n)
    if n not in NUMERICS.values():
        raise ValueError(f"unknown numerics mode {n}")
    return n


def set_numerics(n):
    """Default numerics mode of render() / pixel_samples() AND the mode of the unit-level helpers (hit_sphere, hit_world,
    hit_world_batch, ray_color), which have no parameter for it.  Returns the previous mode."""
    global _default_numerics
    prev = _default_numerics
    _default_numerics = numerics_code(n)
    if lib().rtwo_set_numerics(_default_numerics) != 0:
        raise ValueError(f"unknown numerics mode {n}")
    return prev


class numerics:
    """with O.numerics('contract'): ..."""
    def __init__(self, n):
        self.n = n

    def __enter__(self):
        self.prev = set_numerics(self.n)
        return self

    def __exit__(self, *a):
        set_numerics(self.prev)
        return False


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("rtw_oracle.c", "rtw_oracle_impl.h", "rtw_oracle.h")]
    if (not force and os.path.exists(LIB_PATH)
            and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "librtw_oracle.so"])
    return LIB_PATH


def _scene_struct(ct):
    class S(C.Structure):
        _fields_ = [("n", C.c_int32)] + [(k, C.POINTER(ct)) for k in ("cx", "cy", "cz", "r")] + \
                   [("kind", C.POINTER(C.c_int32))] + [(k, C.POINTER(ct)) for k in ("ar", "ag", "ab", "param")]
    return S


def _camera_struct(ct):
    class Cam(C.Structure):
        _fields_ = [(k, ct * 3) for k in ("origin", "lower_left_corner", "horizontal", "vertical", "u", "v", "w")] + \
                   [("lens_radius", ct)]
    return Cam


SceneF32, SceneF64 = _scene_struct(C.c_float), _scene_struct(C.c_double)
CameraF32, CameraF64 = _camera_struct(C.c_float), _camera_struct(C.c_double)
CAM_FIELDS = ("origin", "lower_left_corner", "horizontal", "vertical", "u", "v", "w")


class Params(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("spp", C.c_int32), ("max_depth", C.c_int32),
                ("seed", C.c_uint64), ("rng_mode", C.c_int32), ("ref_threads", C.c_int32),
                ("n_chunks", C.c_int32), ("product_order", C.c_int32), ("omp_threads", C.c_int32),
                ("gamma", C.c_int32), ("numerics", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("segments", C.c_uint64), ("sphere_tests", C.c_uint64),
                ("rng_draws", C.c_uint64), ("cand_disc", C.c_uint64), ("cand_forward", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.rtwo_rng_next.restype = C.c_uint64
        L.rtwo_rng_f32.restype = C.c_float
        L.rtwo_rng_f64.restype = C.c_double
        L.rtwo_reflectance_f32.restype = C.c_float
        L.rtwo_reflectance_f32.argtypes = [C.c_float, C.c_float]
        L.rtwo_reflectance_f64.restype = C.c_double
        L.rtwo_reflectance_f64.argtypes = [C.c_double, C.c_double]
        L.rtwo_rng_seed.argtypes = [C.c_uint64, C.c_void_p]
        L.rtwo_rng_stream.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]
        _lib = L
    return _lib


def _is64(T):
    return np.dtype(T) == np.float64


def _ct(T):
    return C.c_double if _is64(T) else C.c_float


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def default_n_chunks(spp):
    """the device's default rule (include/rtw_hip.h rtw_params.n_chunks)"""
    return min(int(spp), 256)


def make_scene(flat, T):
    ct = _ct(T)
    S = (SceneF64 if _is64(T) else SceneF32)()
    keep = []
    S.n = int(flat["n"])
    for k in ("cx", "cy", "cz", "r", "ar", "ag", "ab", "param"):
        a = np.ascontiguousarray(flat[k], dtype=T); keep.append(a)
        setattr(S, k, a.ctypes.data_as(C.POINTER(ct)))
    kind = np.ascontiguousarray(flat["kind"], dtype=np.int32); keep.append(kind)
    S.kind = kind.ctypes.data_as(C.POINTER(C.c_int32))
    return S, keep


def make_camera(cam, T):
    """cam: any object with the 7 vec3 fields + lens_radius (e.g. rtw_amd.Camera or a dict)"""
    ct = _ct(T)
    Cm = (CameraF64 if _is64(T) else CameraF32)()
    get = (lambda k: cam[k]) if isinstance(cam, dict) else (lambda k: getattr(cam, k))
    for k in CAM_FIELDS:
        setattr(Cm, k, (ct * 3)(*[float(x) for x in np.asarray(get(k), dtype=T)]))
    Cm.lens_radius = float(np.dtype(T).type(get("lens_radius")))
    return Cm


def camera_to_dict(Cm, T):
    d = {k: np.array(list(getattr(Cm, k)), dtype=T) for k in CAM_FIELDS}
    d["lens_radius"] = np.dtype(T).type(Cm.lens_radius)
    return d


def render(flat_scene, cam, width, height, spp, *, T=np.float32, max_depth=16, seed=1, rng_mode=PIXEL_STREAM,
           ref_threads=1, n_chunks=None, product_order=PRODUCT_FORWARD, omp_threads=0, gamma=True, numerics=None):
    """Oracle render.  Returns (img[i, j, c] of dtype T, stats dict)."""
    L = lib()
    S, keep = make_scene(flat_scene, T)
    Cm = make_camera(cam, T)
    if n_chunks is None or n_chunks <= 0:
        n_chunks = default_n_chunks(spp)
    P = Params(int(width), int(height), int(spp), int(max_depth), int(seed), int(rng_mode), int(ref_threads),
               int(n_chunks), int(product_order), int(omp_threads), 1 if gamma else 0, numerics_code(numerics))
    out = np.empty(int(width) * int(height) * 3, dtype=T)
    st = Stats()
    fn = L.rtwo_render_f64 if _is64(T) else L.rtwo_render_f32
    rc = fn(C.byref(S), C.byref(Cm), C.byref(P), _p(out), C.byref(st))
    if rc != 0:
        raise RuntimeError(f"oracle render failed: {rc}")
    del keep
    img = out.reshape(int(width), int(height), 3).transpose(1, 0, 2)
    return img, {k: getattr(st, k) for k, _ in st._fields_}


def pixel_samples(flat_scene, cam, width, height, spp, i, j, *, T=np.float32, max_depth=16, seed=1, n_chunks=None,
                  product_order=PRODUCT_FORWARD, numerics=None):
    """Radiance of every sample of pixel (i, j) (1-based) in PIXEL_STREAM mode, in sample order -> float64 [spp, 3]."""
    if n_chunks is None:
        n_chunks = default_n_chunks(spp)
    S, keep = make_scene(flat_scene, T)
    Cm = make_camera(cam, T)
    P = Params(int(width), int(height), int(spp), int(max_depth), int(seed), PIXEL_STREAM, 1,
               int(n_chunks), int(product_order), 1, 0, numerics_code(numerics))
    out = np.zeros((int(spp), 3), np.float64)
    fn = getattr(lib(), "rtwo_pixel_samples_f64" if _is64(T) else "rtwo_pixel_samples_f32")
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    rc = fn(C.byref(S), C.byref(Cm), C.byref(P), int(i), int(j), _p(out))
    del keep
    if rc:
        raise ValueError(f"rtwo_pixel_samples failed: {rc}")
    return out


def default_camera(lookfrom, lookat, vup, vfov, aspect, aperture, focus_dist, T=np.float32):
    L = lib()
    ct = _ct(T)
    Cm = (CameraF64 if _is64(T) else CameraF32)()
    fn = L.rtwo_default_camera_f64 if _is64(T) else L.rtwo_default_camera_f32
    fn.argtypes = [ct * 3, ct * 3, ct * 3, ct, ct, ct, ct, C.c_void_p]
    f = lambda v: (ct * 3)(*[float(x) for x in np.asarray(v, dtype=T)])
    tt = np.dtype(T).type
    fn(f(lookfrom), f(lookat), f(vup), float(tt(vfov)), float(tt(aspect)), float(tt(aperture)),
       float(tt(focus_dist)), C.byref(Cm))
    return camera_to_dict(Cm, T)


def scene_random_spheres(seed=1, T=np.float32):
    L = lib()
    arrs = {k: np.zeros(512, dtype=T) for k in ("cx", "cy", "cz", "r", "ar", "ag", "ab", "param")}
    kind = np.zeros(512, dtype=np.int32)
    fn = L.rtwo_scene_random_spheres_f64 if _is64(T) else L.rtwo_scene_random_spheres_f32
    fn.argtypes = [C.c_uint64] + [C.c_void_p] * 9
    n = fn(int(seed), _p(arrs["cx"]), _p(arrs["cy"]), _p(arrs["cz"]), _p(arrs["r"]), _p(kind),
           _p(arrs["ar"]), _p(arrs["ag"]), _p(arrs["ab"]), _p(arrs["param"]))
    flat = {k: v[:n].copy() for k, v in arrs.items()}
    flat["kind"] = kind[:n].copy()
    flat["n"] = n
    return flat


# ---- unit-level helpers (T0 tier) ------------------------------------------------------------
def _v(a, T):
    return np.ascontiguousarray(a, dtype=T)


def reflect(v, n, T=np.float64):
    out = np.zeros(3, T)
    getattr(lib(), "rtwo_reflect_f64" if _is64(T) else "rtwo_reflect_f32")(_p(_v(v, T)), _p(_v(n, T)), _p(out))
    return out


def refract(d, n, ratio, T=np.float64):
    out = np.zeros(3, T)
    fn = getattr(lib(), "rtwo_refract_f64" if _is64(T) else "rtwo_refract_f32")
    fn.argtypes = [C.c_void_p, C.c_void_p, _ct(T), C.c_void_p]
    fn(_p(_v(d, T)), _p(_v(n, T)), float(ratio), _p(out))
    return out


def reflectance(cos_t, ratio, T=np.float64):
    fn = getattr(lib(), "rtwo_reflectance_f64" if _is64(T) else "rtwo_reflectance_f32")
    return np.dtype(T).type(fn(float(cos_t), float(ratio)))


def near_zero(v, T=np.float64):
    return bool(getattr(lib(), "rtwo_near_zero_f64" if _is64(T) else "rtwo_near_zero_f32")(_p(_v(v, T))))


def skycolor(d, T=np.float64):
    out = np.zeros(3, np.float64)
    getattr(lib(), "rtwo_skycolor_f64" if _is64(T) else "rtwo_skycolor_f32")(_p(_v(d, T)), _p(out))
    return out


def hit_sphere(c, r, o, d, tmin, tmax, T=np.float64):
    """-> None or dict(t, p, n, front)"""
    rec = np.zeros(8, T)
    fn = getattr(lib(), "rtwo_hit_sphere_f64" if _is64(T) else "rtwo_hit_sphere_f32")
    ct = _ct(T)
    fn.argtypes = [C.c_void_p, ct, C.c_void_p, C.c_void_p, ct, ct, C.c_void_p]
    ok = fn(_p(_v(c, T)), float(r), _p(_v(o, T)), _p(_v(d, T)), float(tmin), float(tmax), _p(rec))
    if not ok:
        return None
    return dict(t=rec[0], p=rec[1:4].copy(), n=rec[4:7].copy(), front=bool(rec[7]))


def hit_world(flat_scene, o, d, tmin, tmax, T=np.float64):
    S, keep = make_scene(flat_scene, T)
    rec = np.zeros(8, T)
    fn = getattr(lib(), "rtwo_hit_world_f64" if _is64(T) else "rtwo_hit_world_f32")
    ct = _ct(T)
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, ct, ct, C.c_void_p]
    idx = fn(C.byref(S), _p(_v(o, T)), _p(_v(d, T)), float(tmin), float(tmax), _p(rec))
    del keep
    return idx, rec


def hit_world_batch(flat_scene, rays, tmin, tmax, T=np.float64):
    """rays: [n, 6] (o, d) -> (idx[n] int32, t[n] of dtype T), all rays in one C call (OpenMP)"""
    S, keep = make_scene(flat_scene, T)
    rays = np.ascontiguousarray(rays, dtype=T)
    n = rays.shape[0]
    idx = np.zeros(n, np.int32)
    t = np.zeros(n, T)
    fn = getattr(lib(), "rtwo_hit_world_batch_f64" if _is64(T) else "rtwo_hit_world_batch_f32")
    ct = _ct(T)
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_long, ct, ct, C.c_void_p, C.c_void_p]
    fn(C.byref(S), _p(rays), n, float(tmin), float(tmax), _p(idx), _p(t))
    del keep
    return idx, t


def scatter(kind, albedo, param, d, rec8, state, T=np.float64):
    """-> (out9 = o,d,att ; new state)"""
    st = np.array(state, dtype=np.uint64)
    out = np.zeros(9, T)
    fn = getattr(lib(), "rtwo_scatter_f64" if _is64(T) else "rtwo_scatter_f32")
    fn.argtypes = [C.c_int, C.c_void_p, _ct(T), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    fn(int(kind), _p(_v(albedo, T)), float(param), _p(_v(d, T)), _p(_v(rec8, T)), _p(st), _p(out))
    return out, st


def get_ray(cam, s, t, state, T=np.float64):
    Cm = make_camera(cam, T)
    st = np.array(state, dtype=np.uint64)
    out = np.zeros(6, T)
    fn = getattr(lib(), "rtwo_get_ray_f64" if _is64(T) else "rtwo_get_ray_f32")
    fn.argtypes = [C.c_void_p, _ct(T), _ct(T), C.c_void_p, C.c_void_p]
    fn(C.byref(Cm), float(s), float(t), _p(st), _p(out))
    return out, st


def ray_color(flat_scene, o, d, depth, state, product_order=PRODUCT_FORWARD, T=np.float64):
    S, keep = make_scene(flat_scene, T)
    st = np.array(state, dtype=np.uint64)
    out = np.zeros(3, np.float64)
    fn = getattr(lib(), "rtwo_ray_color_f64" if _is64(T) else "rtwo_ray_color_f32")
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    fn(C.byref(S), _p(_v(o, T)), _p(_v(d, T)), int(depth), int(product_order), _p(st), _p(out))
    del keep
    return out, st


def splitmix64(state):
    """one SplitMix64 step on a 1-element uint64 array (advanced in place) -> output"""
    fn = lib().rtwo_splitmix64; fn.argtypes = [C.c_void_p]; fn.restype = C.c_uint64
    return int(fn(_p(state)))


def rng_seed(seed):
    st = np.zeros(2, np.uint64); lib().rtwo_rng_seed(int(seed), _p(st)); return st


def rng_stream(seed, pixel, chunk):
    st = np.zeros(2, np.uint64); lib().rtwo_rng_stream(int(seed), int(pixel), int(chunk), _p(st)); return st


def rng_next(st):
    fn = lib().rtwo_rng_next; fn.argtypes = [C.c_void_p]; return int(fn(_p(st)))


def rng_float(st, T=np.float64):
    fn = getattr(lib(), "rtwo_rng_f64" if _is64(T) else "rtwo_rng_f32"); fn.argtypes = [C.c_void_p]
    return np.dtype(T).type(fn(_p(st)))


def fx_sum(values):
    """exact 64.64 fixed-point sum of binary64 values rounded once -> (sum, n_poisoned)"""
    a = np.ascontiguousarray(values, dtype=np.float64)
    fn = lib().rtwo_fx_sum
    fn.restype = C.c_double
    fn.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    bad = C.c_int(0)
    return float(fn(_p(a), int(a.size), C.byref(bad))), int(bad.value)


def max_threads():
    return int(lib().rtwo_max_threads())
