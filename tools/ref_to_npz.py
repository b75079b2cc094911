#!/usr/bin/env python3
"""Pack one rust_shim/refdump.rs output directory into tests/golden/ref_<name>.npz (what tests/test_reference_fixtures.py consumes).
usage: python tools/ref_to_npz.py <dump dir> <scene name as in tools/export_pbrt.py SCENES>"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pack(d, name, dst):
    meta = json.load(open(os.path.join(d, "meta.json")))
    node_dt = np.dtype([("bmin", "<f4", 3), ("bmax", "<f4", 3), ("offset", "<i4"), ("n_prims", "<u2"), ("axis", "u1"), ("pad", "u1")])
    out = dict(meta=json.dumps(meta), name=name,
               bvh_nodes=np.fromfile(os.path.join(d, "bvh_nodes.bin"), node_dt),
               bvh_prims=np.fromfile(os.path.join(d, "bvh_prims.bin"), "<f4").reshape(-1, 9),
               film=np.fromfile(os.path.join(d, "film.bin"), "<f4").reshape(-1, 4))
    for opt, shape in (("li.bin", (-1, 8)), ("hits.bin", (-1, 17))):
        p = os.path.join(d, opt)
        if os.path.exists(p) and os.path.getsize(p):
            out[opt[:-4]] = np.fromfile(p, "<f4").reshape(shape)
    p = os.path.join(d, "occluded.bin")
    if os.path.exists(p):
        out["occluded"] = np.fromfile(p, np.uint8)
    np.savez_compressed(dst, **out)
    return out


def main():
    d, name = sys.argv[1], sys.argv[2]
    dst = os.path.join(ROOT, "tests", "golden", "ref_%s.npz" % name)
    out = pack(d, name, dst)
    print("wrote", dst, {k: getattr(v, "shape", None) for k, v in out.items()})


if __name__ == "__main__":
    main()
