#!/usr/bin/env python3
"""Render one of the stand-in scenes through librspt and write an 8-bit sRGB PNG the way Film::write_image does
(film.rs:445-527: linear RGB -> gamma_correct -> clamp(255 v + 0.5)).  usage: render_png.py <cornell|textured> <out.png> [res] [spp]"""
import struct
import sys
import zlib

import numpy as np

sys.path.insert(0, ".")


def write_png(path, rgb8):
    h, w, _ = rgb8.shape
    raw = b"".join(b"\x00" + rgb8[y].tobytes() for y in range(h))
    def chunk(t, d):
        c = struct.pack(">I", len(d)) + t + d
        return c + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def main():
    import torch  # noqa: F401  (HIP runtime first)
    from rs_pbrt_amd import lib, scenes
    which, out = sys.argv[1], sys.argv[2]
    res = int(sys.argv[3]) if len(sys.argv) > 3 else 500
    spp = int(sys.argv[4]) if len(sys.argv) > 4 else 64
    lib.init(0)
    if which == "cornell":
        sc = scenes.cornell_box(lib.bvh_build)
        rd = scenes.cornell_render_desc(res=res, spp=spp)
        w = h = res
    else:
        from tests.util import TEXTURED_LOOK_AT, textured_room
        sc = textured_room(lib.bvh_build)
        w, h = res, res * 3 // 4
        rd = scenes.make_render_desc(w, h, spp, TEXTURED_LOOK_AT, 45, max_depth=5)
    with lib.DeviceScene(sc) as ds:
        film, st = lib.render(ds, rd)
    rgb = scenes.film_to_rgb(film).reshape(h, w, 3)
    g = np.where(rgb <= 0.0031308, 12.92 * rgb, 1.055 * np.power(np.maximum(rgb, 0), 1 / 2.4) - 0.055)
    write_png(out, np.clip(255.0 * g + 0.5, 0, 255).astype(np.uint8))
    print("wrote", out, "%.1f Msamples/s" % (st["samples"] / st["t_render_s"] / 1e6))


if __name__ == "__main__":
    main()
