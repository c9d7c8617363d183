"""``render(scene, cam, image_width=400, n_samples=1)`` -- the drop-in boundary.

Host-side mirror of /root/reference/src/render.jl:8-44.  Same positional signature, same
result (an ``H x W`` image of ``RGB{T}``, ``H = image_width ÷ 16//9``, gamma-2 applied,
unclamped), deterministic for a fixed seed.  Keyword extras only: ``depth`` (the reference
hard-wires 16 through ``ray_color``'s default, src/ray_color.jl:14), ``seed``, ``n_chunks``,
``device``.  All compute happens in librtw_hip.so; there is no CPU fallback.
"""
import ctypes as C
import threading

import numpy as np

from . import _capi
from .structs import Camera, flatten_scene, image_height

_tls = threading.local()      # last_stats() is per calling thread, like rtw_stats() itself


def _as_image(flat, height, width):
    """column-major ``Matrix{RGB{T}}`` buffer -> numpy view indexed [i, j, channel]"""
    return flat.reshape(width, height, 3).transpose(1, 0, 2)


def render(scene, cam, image_width=400, n_samples=1, *, depth=16, seed=1, n_chunks=0, device=-1, gamma=True,
           group_cull=False, devices=None, scan_valu=False, ray_pool=False, rccl_reduce=False, numerics=None):
    """Render ``scene`` through ``cam``; returns ``img[i, j, :]`` (row i, column j, RGB) of the
    camera's element type, memory-identical to the reference's ``Matrix{RGB{T}}``.
    ``group_cull=True`` selects the opt-in accelerated scan (same image, include/rtw_hip.h);
    ``scan_valu=True`` the all-VALU plain scan (RTW_FLAG_SCAN_VALU: same image, for A/B measurements);
    ``ray_pool=True`` the ray-pool kernel (RTW_FLAG_RAY_POOL: rays parked in LDS between stages; same image, slower) -- `make POOL=1` builds only,
    the default library raises RtwError -7.
    ``devices``: ``"all"`` or a list of HIP ordinals -- the 8x8 tiles are dealt to those devices
    inside the library (``rtw_params.n_devices/device_ids``); the image is the same for any list.
    ``numerics``: the deciding arithmetic of ``hit(::Sphere)`` (src/hit.jl:16-18): ``"reference"`` (default: StaticArrays' un-fused
    dot, one rounding per written operation), ``"reference_fma2"`` (both squares contracted) or ``"contract"`` (the three FMA
    chains of ABI 2); include/rtw_hip.h RTW_FLAG_NUMERICS_*.
    ``rccl_reduce=True`` (with ``devices``): the shards are put together by ONE ncclReduce of zero-padded frames inside the
    library (RTW_FLAG_RCCL_REDUCE) instead of peer copies of compact shards; ``last_stats()["gather_path"]`` says which path ran."""
    if not isinstance(cam, Camera):
        raise TypeError("cam must be a Camera")
    T = cam.elem_type
    L = _capi.lib()
    height = image_height(image_width)
    if int(image_width) <= 0 or height <= 0:
        raise ValueError(f"image_width={image_width} gives an empty {height} x {image_width} image")
    if int(n_samples) <= 0:
        raise ValueError("n_samples must be >= 1")
    flat = flatten_scene(scene, T)
    S, keep = _capi.make_scene(flat, T)
    Cm = _capi.make_camera(cam, T)
    P = _capi.make_params(image_width, height, n_samples, depth, seed, n_chunks, 0, 1, device, 1 if gamma else 0,
                           (_capi.FLAG_GROUP_CULL if group_cull else 0) | (_capi.FLAG_SCAN_VALU if scan_valu else 0) |
                           (_capi.FLAG_RAY_POOL if ray_pool else 0) | (_capi.FLAG_RCCL_REDUCE if rccl_reduce else 0), devices=devices,
                           numerics=numerics)
    out = np.empty(height * int(image_width) * 3, dtype=T)
    fn = L.rtw_render_f64 if _capi.is_f64(T) else L.rtw_render_f32
    _capi.check(fn(C.byref(S), C.byref(Cm), C.byref(P), out.ctypes.data_as(C.c_void_p)))
    st = _capi.Stats()
    _capi.check(L.rtw_stats(C.byref(st)))
    _tls.stats = {k: getattr(st, k) for k, _ in st._fields_}
    _tls.stats["per_device"] = _stats_devices(L)
    del keep
    return _as_image(out, height, int(image_width))


def _stats_devices(L, cap=64):
    """[(HIP ordinal, kernel ms)] of the shards of the calling thread's last render (rtw_stats_devices)"""
    n = C.c_int32(0)
    dev = (C.c_int32 * cap)()
    ms = (C.c_double * cap)()
    _capi.check(L.rtw_stats_devices(cap, C.byref(n), dev, ms))
    return [(int(dev[k]), float(ms[k])) for k in range(min(cap, n.value))]


def last_stats():
    """Counters/timings of the calling thread's most recent render (include/rtw_hip.h ``rtw_stats_t``)."""
    return getattr(_tls, "stats", None)


class DeviceRenderer:
    """The device-resident variant used by bench.py and the multi-process shard path: the scene
    stays in HBM, the image is written into a caller-provided device buffer (a torch tensor's
    ``data_ptr()``), work is enqueued on the caller's HIP stream."""

    def __init__(self, scene, cam, device=-1):
        self.T = cam.elem_type
        self.L = _capi.lib()
        flat = flatten_scene(scene, self.T)
        S, keep = _capi.make_scene(flat, self.T)
        self.n_spheres = int(flat["n"])
        self.cam = _capi.make_camera(cam, self.T)
        self.handle = C.c_void_p()
        up = self.L.rtw_scene_upload_f64 if _capi.is_f64(self.T) else self.L.rtw_scene_upload_f32
        _capi.check(up(C.byref(S), int(device), C.byref(self.handle)))
        del keep

    def render_into(self, d_out_ptr, image_width, n_samples, *, depth=16, seed=1, n_chunks=0,
                    shard_index=0, shard_count=1, stream=0, gamma=True, group_cull=False, compact=False,
                    scan_valu=False, n_elems=None, ray_pool=False, job_pixels=0, numerics=None):
        """Enqueue one render into device memory at ``d_out_ptr``: H*W*3 elements, or with
        ``compact=True`` only this shard's tiles (``shard.compact_elems`` elements -- whole 8x8 tiles, which for a ragged
        frame can exceed H*W*3), tile-major.  ``n_elems``: the buffer's length in elements; checked when given."""
        height = image_height(image_width)
        if n_elems is not None:
            from .shard import compact_elems
            need = compact_elems(image_width, shard_index, shard_count) if compact else height * int(image_width) * 3
            if int(n_elems) < need:
                raise ValueError(f"output buffer holds {n_elems} elements, this render writes {need}")
        flags = ((_capi.FLAG_GROUP_CULL if group_cull else 0) | (_capi.FLAG_COMPACT_TILES if compact else 0) |
                 (_capi.FLAG_SCAN_VALU if scan_valu else 0) | (_capi.FLAG_RAY_POOL if ray_pool else 0))
        P = _capi.make_params(image_width, height, n_samples, depth, seed, n_chunks, shard_index, shard_count,
                              -1, 1 if gamma else 0, flags, job_pixels=job_pixels, numerics=numerics)
        fn = self.L.rtw_render_device_f64 if _capi.is_f64(self.T) else self.L.rtw_render_device_f32
        _capi.check(fn(self.handle, C.byref(self.cam), C.byref(P), C.c_void_p(int(d_out_ptr)),
                       C.c_void_p(int(stream))))
        return height

    def stats(self):
        st = _capi.Stats()
        _capi.check(self.L.rtw_stats(C.byref(st)))
        return {k: getattr(st, k) for k, _ in st._fields_}

    def close(self):
        if self.handle:
            self.L.rtw_scene_free(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
