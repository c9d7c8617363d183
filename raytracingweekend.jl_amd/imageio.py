"""Output step after the path: ``Matrix{RGB{T}}`` -> 8-bit PPM / PNG,
and an image-difference report for parity write-ups.  The reference has no writer at all
(/root/reference/README.md:138,170); pixels follow Images.jl's ``N0f8`` conversion:
``round(clamp(x, 0, 1) * 255)``.  Pure Python + zlib, host side only."""
import struct
import zlib

import numpy as np


def to_u8(img):
    """float image (gamma already applied by ``render``) -> uint8 [H, W, 3]"""
    a = np.nan_to_num(np.asarray(img, dtype=np.float64), nan=0.0, posinf=1.0, neginf=0.0)
    return np.rint(np.clip(a, 0.0, 1.0) * 255.0).astype(np.uint8)


def save_ppm(img, path):
    u = to_u8(img)
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (u.shape[1], u.shape[0]))
        f.write(np.ascontiguousarray(u).tobytes())
    return path


def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def save_png(img, path):
    u = np.ascontiguousarray(to_u8(img))
    h, w, _ = u.shape
    raw = b"".join(b"\x00" + u[i].tobytes() for i in range(h))      # filter type 0 per scanline
    png = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + \
        _chunk(b"IDAT", zlib.compress(raw, 6)) + _chunk(b"IEND", b"")
    with open(path, "wb") as f:
        f.write(png)
    return path


def load_ppm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"P6"
        w, h = map(int, f.readline().split())
        assert int(f.readline()) == 255
        return np.frombuffer(f.read(w * h * 3), np.uint8).reshape(h, w, 3)


def diff_report(a, b):
    """per-channel difference statistics of two float images of equal shape"""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    if a.shape != b.shape:
        raise ValueError(f"shape mismatch {a.shape} vs {b.shape}")
    d = np.abs(a - b)
    mse = float(np.mean((a - b) ** 2))
    return dict(shape=a.shape, identical=bool(np.array_equal(a, b)), channels_differing=int((a != b).sum()),
                max_abs=float(d.max()), mean_abs=float(d.mean()),
                frac_within_2e3=float((d <= 2e-3).mean()),
                psnr_db=float("inf") if mse == 0 else float(10 * np.log10(1.0 / mse)),
                u8_differing=int((to_u8(a) != to_u8(b)).sum()))
