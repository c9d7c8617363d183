"""Scene / material / camera structs of the reference, host side.

Mirrors /root/reference/src/structs.jl:8-35 (``Hittable``, ``HittableList``, ``Material``,
``Sphere``), src/material.jl:3-5,25-29,37-39 (``Lambertian``, ``Metal(albedo, fuzz=0)``,
``Dielectric``) and src/camera.jl:1-41 (``Camera``, both ``default_camera`` methods).
All arithmetic is done in ``elem_type`` with numpy scalars, one rounding per operation, in the
order the reference writes it, so a Float32 camera is bit-identical to the oracle's mirror.
"""
from dataclasses import dataclass

import numpy as np

LAMBERTIAN, METAL, DIELECTRIC = 0, 1, 2


def _vec3(v, T):
    a = np.asarray(v, dtype=T).reshape(3)
    return a


class Material:
    """``abstract type Material{T}`` (src/structs.jl:13)."""


@dataclass
class Lambertian(Material):
    albedo: np.ndarray  # Vec3{T}

    def __post_init__(self):
        self.albedo = np.asarray(self.albedo)
        if self.albedo.dtype not in (np.float32, np.float64):
            self.albedo = self.albedo.astype(np.float64)


@dataclass
class Metal(Material):
    albedo: np.ndarray
    fuzz: float = 0.0  # src/material.jl:28

    def __post_init__(self):
        self.albedo = np.asarray(self.albedo)
        if self.albedo.dtype not in (np.float32, np.float64):
            self.albedo = self.albedo.astype(np.float64)


@dataclass
class Dielectric(Material):
    ir: float  # index of refraction


class Hittable:
    """``abstract type Hittable`` (src/structs.jl:8)."""


@dataclass
class Sphere(Hittable):
    center: np.ndarray
    radius: float
    mat: Material


class HittableList(list):
    """``const HittableList = Vector{Hittable}`` (src/structs.jl:10)."""


def flatten_scene(scene, T):
    """HittableList -> the SoA arrays the C ABI takes (include/rtw_hip.h ``rtw_scene_*``).

    Non-``Sphere`` hittables and unknown materials raise ``TypeError`` (the reference would
    raise a ``MethodError`` inside ``hit``/``scatter``).
    """
    T = np.dtype(T).type
    n = len(scene)
    cx = np.zeros(n, T); cy = np.zeros(n, T); cz = np.zeros(n, T); r = np.zeros(n, T)
    kind = np.zeros(n, np.int32)
    ar = np.zeros(n, T); ag = np.zeros(n, T); ab = np.zeros(n, T); param = np.zeros(n, T)
    for i, s in enumerate(scene):
        if not isinstance(s, Sphere):
            raise TypeError(f"scene[{i}] is {type(s).__name__}; only Sphere is supported on this path")
        c = _vec3(s.center, T)
        cx[i], cy[i], cz[i], r[i] = c[0], c[1], c[2], T(s.radius)
        m = s.mat
        if isinstance(m, Lambertian):
            kind[i] = LAMBERTIAN; a = _vec3(m.albedo, T); ar[i], ag[i], ab[i] = a
        elif isinstance(m, Metal):
            kind[i] = METAL; a = _vec3(m.albedo, T); ar[i], ag[i], ab[i] = a; param[i] = T(m.fuzz)
        elif isinstance(m, Dielectric):
            kind[i] = DIELECTRIC; ar[i] = ag[i] = ab[i] = T(1); param[i] = T(m.ir)
        else:
            raise TypeError(f"scene[{i}].mat is {type(m).__name__}; expected Lambertian, Metal or Dielectric")
    return dict(n=n, cx=cx, cy=cy, cz=cz, r=r, kind=kind, ar=ar, ag=ag, ab=ab, param=param)


@dataclass
class Camera:
    """``struct Camera{T}`` (src/camera.jl:1-10), field order preserved."""
    origin: np.ndarray
    lower_left_corner: np.ndarray
    horizontal: np.ndarray
    vertical: np.ndarray
    u: np.ndarray
    v: np.ndarray
    w: np.ndarray
    lens_radius: float

    @property
    def elem_type(self):
        return self.origin.dtype.type


def _tand(x, T):
    """``tand`` (src/camera.jl:23): exact at multiples of 45 deg, else extended precision -> T."""
    xl = np.longdouble(x)
    m = np.fmod(xl, np.longdouble(180))
    if m == 0:
        return T(0)
    if m in (45, -135):
        return T(1)
    if m in (-45, 135):
        return T(-1)
    pi = np.longdouble("3.14159265358979323846264338327950288")
    return T(np.tan(m * (pi / np.longdouble(180))))


def _normalize(a):
    T = a.dtype.type
    inv = T(1) / np.sqrt((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2])
    return np.array([inv * a[0], inv * a[1], inv * a[2]], dtype=T)


def _cross(a, b):
    return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], dtype=a.dtype)


def default_camera(lookfrom=(0, 0, 0), lookat=(0, 0, -1), vup=(0, 1, 0), vfov=90, aspect_ratio=16 / 9,
                   aperture=0, focus_dist=1, *, elem_type=np.float32):
    """``default_camera`` (src/camera.jl:18-41); ``vfov`` in degrees; thin lens."""
    T = np.dtype(elem_type).type
    lookfrom, lookat, vup = _vec3(lookfrom, T), _vec3(lookat, T), _vec3(vup, T)
    vfov, aspect_ratio, aperture, focus_dist = T(vfov), T(aspect_ratio), T(aperture), T(focus_dist)
    viewport_height = T(2) * _tand(vfov / T(2), T)          # :23
    viewport_width = aspect_ratio * viewport_height          # :24
    w = _normalize(lookfrom - lookat)                        # :26
    u = _normalize(_cross(vup, w))                           # :27
    v = _cross(w, u)                                         # :28
    origin = lookfrom
    horizontal = (focus_dist * viewport_width) * u           # :31
    vertical = (focus_dist * viewport_height) * v            # :32
    lower_left_corner = ((origin - horizontal / T(2)) - vertical / T(2)) - focus_dist * w  # :33
    lens_radius = aperture / T(2)                            # :34
    return Camera(origin, lower_left_corner.astype(T), horizontal.astype(T), vertical.astype(T), u, v, w, lens_radius)


def image_height(image_width):
    """``image_width ÷ 16//9`` (src/render.jl:11-12)."""
    return (int(image_width) * 9) // 16
