"""ctypes binding of librtw_hip.so (C ABI: include/rtw_hip.h).  No fallback of any kind."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTW_HIP_LIB") or os.path.join(_HERE, "lib", "librtw_hip.so")   # env: kernel A/B experiments only

#: every symbol include/rtw_hip.h declares
SYMBOLS = [
    "rtw_abi_version", "rtw_device_count", "rtw_last_error", "rtw_render_f32", "rtw_render_f64",
    "rtw_scene_upload_f32", "rtw_scene_upload_f64", "rtw_scene_free", "rtw_render_device_f32",
    "rtw_render_device_f64", "rtw_stats", "rtw_stats_devices", "rtw_unit_f32", "rtw_unit_f64", "rtw_shutdown",
]


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


class Params(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("spp", C.c_int32), ("max_depth", C.c_int32),
                ("seed", C.c_uint64), ("n_chunks", C.c_int32), ("shard_index", C.c_int32),
                ("shard_count", C.c_int32), ("device", C.c_int32), ("gamma", C.c_int32), ("flags", C.c_int32),
                ("n_devices", C.c_int32), ("job_pixels", C.c_int32), ("device_ids", C.POINTER(C.c_int32))]


class Stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("segments", C.c_uint64), ("sphere_tests", C.c_uint64),
                ("kernel_ms", C.c_double), ("total_ms", C.c_double), ("n_chunks", C.c_int32),
                ("grid_blocks", C.c_int32), ("block_threads", C.c_int32), ("gather_path", C.c_int32)]


_lib = None


class RtwError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"librtw_hip error {code}: {msg}")
        self.code = code


def lib():
    """Load librtw_hip.so; raise loudly if it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C raytracingweekend.jl_amd/csrc`).  There is no CPU fallback for the render path.")
    L = C.CDLL(LIB_PATH)
    L.rtw_last_error.restype = C.c_char_p
    L.rtw_abi_version.restype = C.c_int
    for name in SYMBOLS:
        getattr(L, name)  # AttributeError if the ABI is incomplete
    L.rtw_scene_upload_f32.argtypes = [C.POINTER(SceneF32), C.c_int, C.POINTER(C.c_void_p)]
    L.rtw_scene_upload_f64.argtypes = [C.POINTER(SceneF64), C.c_int, C.POINTER(C.c_void_p)]
    L.rtw_scene_free.argtypes = [C.c_void_p]
    L.rtw_render_device_f32.argtypes = [C.c_void_p, C.POINTER(CameraF32), C.POINTER(Params), C.c_void_p, C.c_void_p]
    L.rtw_render_device_f64.argtypes = [C.c_void_p, C.POINTER(CameraF64), C.POINTER(Params), C.c_void_p, C.c_void_p]
    L.rtw_render_f32.argtypes = [C.POINTER(SceneF32), C.POINTER(CameraF32), C.POINTER(Params), C.c_void_p]
    L.rtw_render_f64.argtypes = [C.POINTER(SceneF64), C.POINTER(CameraF64), C.POINTER(Params), C.c_void_p]
    L.rtw_stats.argtypes = [C.POINTER(Stats)]
    L.rtw_stats_devices.argtypes = [C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    L.rtw_unit_f32.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(SceneF32), C.POINTER(CameraF32)]
    L.rtw_unit_f64.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(SceneF64), C.POINTER(CameraF64)]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise RtwError(rc, lib().rtw_last_error().decode("utf-8", "replace"))


def is_f64(T):
    return np.dtype(T) == np.float64


def make_scene(flat, T):
    """dict from structs.flatten_scene -> (ctypes struct, keep-alive list)"""
    ct = C.c_double if is_f64(T) else C.c_float
    S = (SceneF64 if is_f64(T) else SceneF32)()
    keep = []
    S.n = int(flat["n"])
    for k in ("cx", "cy", "cz", "r", "ar", "ag", "ab", "param"):
        a = np.ascontiguousarray(flat[k], dtype=T)
        keep.append(a)
        setattr(S, k, a.ctypes.data_as(C.POINTER(ct)))
    kind = np.ascontiguousarray(flat["kind"], dtype=np.int32)
    keep.append(kind)
    S.kind = kind.ctypes.data_as(C.POINTER(C.c_int32))
    return S, keep


def make_camera(cam, T):
    ct = C.c_double if is_f64(T) else C.c_float
    Cm = (CameraF64 if is_f64(T) else CameraF32)()
    for k in ("origin", "lower_left_corner", "horizontal", "vertical", "u", "v", "w"):
        setattr(Cm, k, (ct * 3)(*[float(x) for x in np.asarray(getattr(cam, k), dtype=T)]))
    Cm.lens_radius = float(np.dtype(T).type(cam.lens_radius))
    return Cm


FLAG_GROUP_CULL = 1      # include/rtw_hip.h RTW_FLAG_GROUP_CULL
FLAG_COMPACT_TILES = 2   # include/rtw_hip.h RTW_FLAG_COMPACT_TILES
FLAG_SCAN_VALU = 4       # include/rtw_hip.h RTW_FLAG_SCAN_VALU
FLAG_RAY_POOL = 8        # include/rtw_hip.h RTW_FLAG_RAY_POOL
FLAG_RCCL_REDUCE = 16    # include/rtw_hip.h RTW_FLAG_RCCL_REDUCE
FLAG_NUMERICS_CONTRACT = 32        # include/rtw_hip.h RTW_FLAG_NUMERICS_CONTRACT
FLAG_NUMERICS_REFERENCE_FMA2 = 128  # include/rtw_hip.h RTW_FLAG_NUMERICS_REFERENCE_FMA2
GATHER_PEER, GATHER_HOST_STAGED, GATHER_RCCL, GATHER_SAME_DEVICE = 1, 2, 4, 8    # rtw_stats_t.gather_path bits
ABI_VERSION = 4

# The deciding arithmetic of the ray-sphere test (src/hit.jl:16-18; include/rtw_hip.h RTW_FLAG_NUMERICS_*).
# name -> (rtw_params.flags bits, the mode code in bits 8-9 of a unit op)
NUMERICS = {"reference": (0, 0), "contract": (FLAG_NUMERICS_CONTRACT, 1), "reference_fma2": (FLAG_NUMERICS_REFERENCE_FMA2, 3)}
_default_numerics = "reference"


def set_default_numerics(name):
    """TEST HELPER (tests/conftest.py runs every parity module once per mode with it) -- process-wide, not for product code: name the mode
    per call instead.  The mode ``render`` / ``DeviceRenderer`` / ``make_params`` use when none is named.  Returns the previous default."""
    global _default_numerics
    if name not in NUMERICS:
        raise ValueError(f"numerics must be one of {sorted(NUMERICS)}")
    prev, _default_numerics = _default_numerics, name
    return prev


def numerics_name(numerics=None):
    name = _default_numerics if numerics is None else numerics
    if name not in NUMERICS:
        raise ValueError(f"numerics must be one of {sorted(NUMERICS)}")
    return name


def numerics_flags(numerics=None):
    return NUMERICS[numerics_name(numerics)][0]


def numerics_unit_bits(numerics=None):
    """what to OR into the ``op`` of rtw_unit_f32/_f64"""
    return NUMERICS[numerics_name(numerics)][1] << 8


def make_params(width, height, spp, max_depth=16, seed=1, n_chunks=0, shard_index=0, shard_count=1,
                device=-1, gamma=1, flags=0, devices=None, job_pixels=0, numerics=None):
    """``numerics``: "reference" (default) / "contract" / "reference_fma2", or None = the module default
    (``set_default_numerics``).  ``flags`` may name the mode instead (a NUMERICS bit); naming a DIFFERENT mode both ways is a ValueError
    (the mode changes the image: it is never overridden silently).
    ``devices``: None / int ordinal -> one device; "all" -> every visible device (n_devices = -1);
    a sequence of ordinals -> that device list (host-buffer entry points only)."""
    flags = int(flags)
    named = flags & (FLAG_NUMERICS_CONTRACT | FLAG_NUMERICS_REFERENCE_FMA2)
    if not named:
        flags |= numerics_flags(numerics)
    elif numerics is not None and numerics_flags(numerics) != named:
        raise ValueError(f"numerics={numerics!r} conflicts with the NUMERICS bit already in flags (0x{named:x})")
    P = Params(int(width), int(height), int(spp), int(max_depth), int(seed), int(n_chunks),
               int(shard_index), int(shard_count), int(device), int(gamma), flags, 0, int(job_pixels), None)
    if devices is None:
        return P
    if isinstance(devices, str):
        if devices != "all":
            raise ValueError("devices must be None, 'all', an ordinal or a sequence of ordinals")
        P.n_devices = -1
    elif isinstance(devices, (int, np.integer)):
        P.device = int(devices)
    else:
        ids = (C.c_int32 * len(devices))(*[int(d) for d in devices])
        P._keep_ids = ids                     # keep the array alive as long as the struct
        if len(devices) == 1:
            P.device = int(devices[0])
        P.n_devices = len(devices)            # (a list of one: the same device, named both ways)
        P.device_ids = C.cast(ids, C.POINTER(C.c_int32))
    return P
