"""raytracingweekend.jl_amd -- MI355X (gfx950) implementation of ONE hot path of
claforte/RayTracingWeekend.jl: ``render -> ray_color -> hit/scatter``.

Host-side mirror of the reference's exported interface for that path
(/root/reference/src/RayTracingWeekend.jl:9-31): same names, argument meaning and defaults.
The compute is ``lib/librtw_hip.so`` (hand-written HIP, C ABI in include/rtw_hip.h); this
package only flattens the scene, calls it through ctypes and wraps the result.  There is no
CPU fallback: without the built library or without a GPU every render call raises.

The directory name contains a dot, so it is imported through the alias module ``rtw_amd``
at the repository root (``import rtw_amd``).
"""
from .rng import TRNG, Xoroshiro128Plus, reseed, trand, random_between  # noqa: F401
from .structs import (  # noqa: F401
    Camera, Dielectric, HittableList, Lambertian, Material, Metal, Sphere, default_camera,
    flatten_scene, image_height,
)
from .scenes import (  # noqa: F401
    scene_2_spheres, scene_4_spheres, scene_blue_red_spheres, scene_diel_spheres,
    scene_random_spheres, t_cam1, t_cam2, t_default_cam,
)
from .render import DeviceRenderer, render, last_stats  # noqa: F401
from .shard import compact_elems, compact_to_frame_index, local_tile_count, owned_pixel_mask, render_sharded  # noqa: F401
from . import imageio  # noqa: F401

__all__ = [
    "TRNG", "Xoroshiro128Plus", "reseed", "trand", "random_between",
    "Camera", "Dielectric", "HittableList", "Lambertian", "Material", "Metal", "Sphere",
    "default_camera", "flatten_scene", "image_height",
    "scene_2_spheres", "scene_4_spheres", "scene_blue_red_spheres", "scene_diel_spheres",
    "scene_random_spheres", "t_cam1", "t_cam2", "t_default_cam",
    "DeviceRenderer", "render", "last_stats", "owned_pixel_mask", "render_sharded", "compact_elems",
    "compact_to_frame_index", "local_tile_count",
]
