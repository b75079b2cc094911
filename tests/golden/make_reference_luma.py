#!/usr/bin/env python3
"""Derives a 50x50 linear-luminance thumbnail from the only renderer output that ships in the reference tree:
docs/source/cornell_box_256_pixelsamples.png (500x500, 8-bit sRGB, `Sampler "sobol" 256 spp`, `Integrator "path"`,
docs/source/getting_started.rst:159-209).  Its scene file (rs-pbrt-test-scenes/pbrt/cornell_box/cornell_box.pbrt) is
NOT in the tree, so materials and light radiance are unknown: the thumbnail is a structural sanity anchor (camera,
film orientation, framing, light position), not a numerical golden.  Needs PIL; run where /root/reference exists.

    python tests/golden/make_reference_luma.py
"""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/docs/source/cornell_box_256_pixelsamples.png"


def luma50(rgb_linear):
    y = 0.212671 * rgb_linear[..., 0] + 0.715160 * rgb_linear[..., 1] + 0.072169 * rgb_linear[..., 2]
    return y.reshape(50, 10, 50, 10).mean(axis=(1, 3)).astype(np.float32)


def main():
    a = np.asarray(Image.open(SRC).convert("RGB"), np.float32) / 255.0
    lin = np.where(a <= 0.04045, a / 12.92, ((a + 0.055) / 1.055) ** 2.4)  # inverse of Film::write_image's gamma_correct
    np.savez_compressed(os.path.join(HERE, "reference_cornell_256spp_luma50.npz"), luma=luma50(lin),
                        source=np.array("rs_pbrt docs/source/cornell_box_256_pixelsamples.png (500x500 sobol 256 spp path)"))


if __name__ == "__main__":
    main()
