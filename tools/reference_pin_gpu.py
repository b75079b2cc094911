#!/usr/bin/env python3
"""The product against the reference's own renders (tests/test_reference_pin.py has the story): renders scenes.cornell_box_docs on the GPU at
8 and 256 spp through the C ABI and prints the byte agreement of each film with the reference's PNG (tests/golden/reference_cornell_docs.npz).
usage (through gpurun): python tools/reference_pin_gpu.py > gpurun_out/<tag>/reference_pin_gpu.txt"""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from rs_pbrt_amd import lib, scenes  # noqa: E402
from tests.test_reference_pin import G, agreement  # noqa: E402

lib.init(0)
sc = scenes.cornell_box_docs(lib.bvh_build)
ds = lib.DeviceScene(sc)
print("library %s, scene: scenes.cornell_box_docs (32 triangles + 2 emitting), 500 x 500, sobol, path maxdepth 5" % lib.source_hash())
for spp in (8, 256):
    rd = scenes.cornell_docs_render_desc(spp)
    lib.render(ds, rd)
    t = time.time(); film, st = lib.render(ds, rd); dt = time.time() - t
    e, w1, w4 = agreement(film, G["spp%d" % spp])
    print("%3d spp: pixels byte-equal to the reference's PNG %.4f, within 1 / 255 %.4f, within 4 / 255 %.4f   (rspt_render %.1f ms, %.0f Msamples/s)" % (spp, e, w1, w4, dt * 1e3, 250000 * spp / dt / 1e6))
ds.close(); lib.shutdown()
