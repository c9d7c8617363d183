"""Scene builders and benchmark cameras of the reference, host side.

Mirrors /root/reference/src/scenes.jl:1-84 and the camera presets of
/root/reference/src/proto/proto.jl:17-22.  ``scene_random_spheres`` draws from ``trand`` exactly
like the reference (thread-1 generator, whatever its state): call ``reseed()`` first to get the
benchmark scene (src/proto/proto.jl:198-199).
"""
import numpy as np

from .rng import random_between, trand
from .structs import Dielectric, HittableList, Lambertian, Metal, Sphere, default_camera


def _sa(T, *v):
    return np.array(v, dtype=T)


def scene_2_spheres(*, elem_type=np.float32):
    """src/scenes.jl:2-11"""
    T = np.dtype(elem_type).type
    s = HittableList()
    s.append(Sphere(_sa(T, 0, 0, -1), T(0.5), Lambertian(_sa(T, 0.7, 0.3, 0.3))))
    s.append(Sphere(_sa(T, 0, -100.5, -1), T(100), Lambertian(_sa(T, 0.8, 0.8, 0.0))))
    return s


def scene_4_spheres(*, elem_type=np.float32):
    """src/scenes.jl:16-23"""
    T = np.dtype(elem_type).type
    s = scene_2_spheres(elem_type=T)
    s.append(Sphere(_sa(T, -1, 0, -1), T(0.5), Metal(_sa(T, 0.8, 0.8, 0.8), T(0.3))))
    s.append(Sphere(_sa(T, 1, 0, -1), T(0.5), Metal(_sa(T, 0.8, 0.6, 0.2), T(0.8))))
    return s


def scene_diel_spheres(left_radius=0.5, *, elem_type=np.float32):
    """src/scenes.jl:25-39 (negative ``left_radius`` = hollow glass bubble)"""
    T = np.dtype(elem_type).type
    s = HittableList()
    s.append(Sphere(_sa(T, 0, 0, -1), T(0.5), Lambertian(_sa(T, 0.1, 0.2, 0.5))))
    s.append(Sphere(_sa(T, 0, -100.5, -1), T(100), Lambertian(_sa(T, 0.8, 0.8, 0.0))))
    s.append(Sphere(_sa(T, -1, 0, -1), T(left_radius), Dielectric(T(1.5))))
    s.append(Sphere(_sa(T, 1, 0, -1), T(0.5), Metal(_sa(T, 0.8, 0.6, 0.2), T(0))))
    return s


def scene_blue_red_spheres(*, elem_type=np.float64):
    """src/scenes.jl:41-47.  ``R = cos(pi/4)`` is a Float64, so the reference's ``Sphere``
    constructor only accepts it for ``elem_type=Float64``; same here."""
    T = np.dtype(elem_type).type
    if T is not np.float64:
        raise TypeError("scene_blue_red_spheres: R = cos(pi/4) is Float64; the reference only constructs it for Float64")
    R = np.cos(np.pi / 4)
    s = HittableList()
    s.append(Sphere(_sa(T, -R, 0, -1), T(R), Lambertian(_sa(T, 0, 0, 1))))
    s.append(Sphere(_sa(T, R, 0, -1), T(R), Lambertian(_sa(T, 1, 0, 0))))
    return s


def scene_random_spheres(*, elem_type=np.float32):
    """src/scenes.jl:49-84"""
    T = np.dtype(elem_type).type
    s = HittableList()
    s.append(Sphere(_sa(T, 0, -1000, -1), T(1000), Lambertian(_sa(T, 0.5, 0.5, 0.5))))   # :53
    for a in range(-11, 11):
        for b in range(-11, 11):                                                       # :56
            choose_mat = trand(T)                                                      # :57
            cx = T(a) + T(0.9) * trand(T)                                              # :58
            cy = T(0.2)
            cz = T(b) + T(0.9) * trand(T)
            dx, dy, dz = cx - T(4), cy - T(0.2), cz - T(0)
            if np.sqrt((dx * dx + dy * dy) + dz * dz) < T(0.9):                        # :61
                continue
            center = _sa(T, cx, cy, cz)
            if choose_mat < T(0.8):                                                    # :63-66
                a0, a1, a2 = trand(T), trand(T), trand(T)
                b0, b1, b2 = trand(T), trand(T), trand(T)
                s.append(Sphere(center, T(0.2), Lambertian(_sa(T, a0 * b0, a1 * b1, a2 * b2))))
            elif choose_mat < T(0.95):                                                 # :67-71
                a0 = random_between(T(0.5), T(1.0)); a1 = random_between(T(0.5), T(1.0)); a2 = random_between(T(0.5), T(1.0))
                fuzz = random_between(T(0.0), T(5.0))
                s.append(Sphere(center, T(0.2), Metal(_sa(T, a0, a1, a2), fuzz)))
            else:                                                                      # :72-75
                s.append(Sphere(center, T(0.2), Dielectric(T(1.5))))
    s.append(Sphere(_sa(T, 0, 1, 0), T(1), Dielectric(T(1.5))))                        # :78
    s.append(Sphere(_sa(T, -4, 1, 0), T(1), Lambertian(_sa(T, 0.4, 0.2, 0.1))))        # :79
    s.append(Sphere(_sa(T, 4, 1, 0), T(1), Metal(_sa(T, 0.7, 0.6, 0.5), T(0))))        # :81
    return s


def t_default_cam(*, elem_type=np.float32):
    """``default_camera(SA{T}[0,0,0])`` (src/proto/proto.jl:17)"""
    return default_camera((0, 0, 0), elem_type=elem_type)


def t_cam1(*, elem_type=np.float32):
    """``default_camera([13,2,3],[0,0,0],[0,1,0],20,16/9,0.1,10.0)`` (src/proto/proto.jl:19)"""
    return default_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20, 16 / 9, 0.1, 10.0, elem_type=elem_type)


def t_cam2(*, elem_type=np.float32):
    """src/proto/proto.jl:21-22: aperture 2.0, focus at the look-at distance"""
    d = np.linalg.norm(np.array([3.0, 3.0, 2.0]) - np.array([0.0, 0.0, -1.0]))
    return default_camera((3, 3, 2), (0, 0, -1), (0, 1, 0), 20, 16 / 9, 2.0, d, elem_type=elem_type)
