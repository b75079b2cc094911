"""Tiny PNG writer (zlib only) for eyeballing films; sRGB gamma as Film::write_image (film.rs:465-520)."""
import struct
import zlib

import numpy as np


def write_png(path, rgb_linear):
    a = np.asarray(rgb_linear, np.float32)
    a = np.where(a <= 0.0031308, 12.92 * a, 1.055 * np.power(np.maximum(a, 0), 1 / 2.4) - 0.055)
    a = (np.clip(a, 0, 1) * 255 + 0.5).astype(np.uint8)
    h, w, _ = a.shape
    raw = b"".join(b"\x00" + a[y].tobytes() for y in range(h))

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def write_png_u8(path, rgb8):
    """8-bit RGB as it is (no gamma), rows top to bottom; stored without compression so that the bytes do not depend on the zlib build"""
    a = np.ascontiguousarray(rgb8, np.uint8)
    h, w, _ = a.shape
    raw = b"".join(b"\x00" + a[y].tobytes() for y in range(h))

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 0)) + chunk(b"IEND", b""))
