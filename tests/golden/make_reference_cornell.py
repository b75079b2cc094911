#!/usr/bin/env python3
"""The only renderer output that ships in the reference tree, as a test fixture: docs/source/cornell_box_8_pixelsamples.png and
docs/source/cornell_box_256_pixelsamples.png (500 x 500, 8-bit sRGB; `rs_pbrt --path .../cornell_box.pbrt`, Sampler "sobol" with 8 and
256 pixel samples, Integrator "path": docs/source/getting_started.rst:150-209).  This script stores their pixels, unchanged, in
tests/golden/reference_cornell_docs.npz (the reference tree does not exist on the GPU box).  Needs PIL; run where /root/reference exists:

    python tests/golden/make_reference_cornell.py            # writes the fixture
    python tests/golden/make_reference_cornell.py --recover  # = python tools/recover_cornell_docs.py: re-runs the searches and prints what they find (minutes)

The scene FILE of those renders (rs-pbrt-test-scenes/pbrt/cornell_box/cornell_box.pbrt) is not in the tree.  rs_pbrt_amd/scenes.py
`cornell_box_docs` is that scene as recovered from the two images, in this order (every step is a search whose objective is agreement
with the images, and each one is re-run by --recover):

 1. fov: the box's frame edges in the 256-spp image (sub-pixel, from the coverage of the edge pixels) -> 39.148 degrees.
 2. radiance and albedos: least squares on the 256-spp image -> L = 100, walls 0.4, red (0.5, 0, 0), green (0, 0.5, 0), blocks 0.5
    (the fit returns 100.7, 0.397 .. 0.399, 0.497 .. 0.500; the red wall's G and B bytes are exactly 0).
 3. handedness: with the public Cornell data and the camera mirrored (`Scale -1 1 1`) the oracle's 8-spp noise does not correlate with the
    reference's (0.1); with the WORLD mirrored (x negated) it does — cross products (ts = cross(ns, ss)) see the difference.
 4. BSDF frames: a triangle's first edge is its dpdu (default uv), i.e. the frame every cosine-sampled bounce is built on.  Per visible
    triangle, the noise correlation over its footprint picks one of 2 diagonals x 3 rotations x 2 windings; all of them come out as fans
    (k, k+1, k+2), (k, k+2, k+3) of the quad's vertex cycle.  Faces the camera does not see: coordinate descent on the byte differences.
 5. the emitter's two triangles (choice, barycentric mapping): all 36 triangulations that face down; one stands out (log-ratio IQR of the
    directly lit faces 0.05 against >= 0.08).
 6. block corners and the light's height: local scans of the byte differences around each silhouette edge (half-unit steps).

What remains unexplained after that is in tests/test_reference_pin.py's docstrings: 2 - 2.5 % of the 8-spp noise variance.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/docs/source/cornell_box_%d_pixelsamples.png"


def main():
    from PIL import Image
    px = {n: np.asarray(Image.open(SRC % n).convert("RGB"), np.uint8) for n in (8, 256)}
    assert px[8].shape == px[256].shape == (500, 500, 3)
    np.savez_compressed(os.path.join(HERE, "reference_cornell_docs.npz"), spp8=px[8], spp256=px[256],
                        source=np.array("rs_pbrt docs/source/cornell_box_{8,256}_pixelsamples.png, pixels unchanged (8-bit sRGB)"))
    print("wrote reference_cornell_docs.npz", os.path.getsize(os.path.join(HERE, "reference_cornell_docs.npz")), "bytes")


if __name__ == "__main__":
    if "--recover" in sys.argv:
        sys.path.insert(0, os.path.join(HERE, "..", ".."))
        from tools import recover_cornell_docs
        recover_cornell_docs.main()
    else:
        main()
