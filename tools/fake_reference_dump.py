#!/usr/bin/env python3
"""Self-test of the reference-fixture pipeline WITHOUT rs_pbrt: writes, from this repo's own CPU oracle, the files rust_shim/refdump.rs
would write for one scene of tools/export_pbrt.py (same names, same layouts), so that tools/ref_to_npz.py and the checks of
tests/test_reference_fixtures.py can be run end to end before a real dump exists.  What comes out is the oracle compared with itself: it
pins NOTHING and must never be committed as tests/golden/ref_*.npz (the test that uses this writes into a temporary directory).
usage: python tools/fake_reference_dump.py <scene name> <out dir>"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def write_dump(oracle, name, d):
    from export_pbrt import EXTRA, SCENES, camera_of, render_kwargs
    from rs_pbrt_amd import abi, lib, scenes
    os.makedirs(d, exist_ok=True)
    mk, _cam, xres, yres, spp, depth = SCENES[name]
    sc = mk(lib.bvh_build, scenes)
    look_at, fov = camera_of(name, scenes)
    rd = scenes.make_render_desc(xres, yres, spp, look_at, fov, max_depth=depth, **render_kwargs(name, scenes))
    nt_nodes, nt_prims = sc.n_top
    sc.nodes[:nt_nodes].tofile(os.path.join(d, "bvh_nodes.bin"))          # LinearBVHNode == rspt_bvh_node, 32 B
    tri = sc.P[sc.prims["v"][:nt_prims]].reshape(-1, 9).astype(np.float32)
    tri[sc.prims["mesh"][:nt_prims] == abi.MESH_INSTANCE] = np.nan           # refdump.rs tri_vertices: a TransformedPrimitive has no triangle
    tri.tofile(os.path.join(d, "bvh_prims.bin"))
    if EXTRA.get(name, {}).get("integrator") == "directlighting":
        r = oracle.render_integrator(sc, rd, "direct", strategy=EXTRA[name].get("direct_strategy", "all"), threads=4, want_li=True)
    else:
        r = oracle.render(sc, rd, threads=4, want_li=True)
    r["film"].astype("<f4").tofile(os.path.join(d, "film.bin"))
    w = rd.crop_px[2] - rd.crop_px[0]
    npix, nspp = r["li"].shape[:2]
    pix = np.repeat(np.arange(npix), nspp)
    rows = np.zeros((npix * nspp, 8), np.float32)                            # px, py, sample, p_film.x, p_film.y, r, g, b
    rows[:, 0] = rd.crop_px[0] + pix % w; rows[:, 1] = rd.crop_px[1] + pix // w; rows[:, 2] = np.tile(np.arange(nspp), npix)
    rows[:, 5:8] = r["li"].reshape(-1, 3)
    rows.tofile(os.path.join(d, "li.bin"))
    rays = np.fromfile(os.path.join(ROOT, "tests", "golden", "ref_scenes", "rays.bin"), abi.RAY_DT)
    h = oracle.trace(sc, rays)
    hit = h["prim"] != abi.MISS
    hits = np.zeros((len(rays), 17), np.float32)
    hits[hit, 0] = 1.0; hits[hit, 1] = h["t"][hit]
    top = hit & (h["prim"] < nt_prims)
    hits[top, 8:17] = sc.P[sc.prims["v"][h["prim"][top]]].reshape(-1, 9)
    hits[hit & ~top, 8:17] = np.nan                                           # an instanced hit has lost its primitive (transform.rs:856)
    hits.tofile(os.path.join(d, "hits.bin"))
    (oracle.trace(sc, rays, any_hit=True)["prim"] == 0).astype(np.uint8).tofile(os.path.join(d, "occluded.bin"))
    json.dump(dict(crop_px=[int(x) for x in rd.crop_px], sample_bounds=[int(x) for x in rd.sample_bounds], spp=int(rd.spp),
                   note="FABRICATED from the repo's own oracle by tools/fake_reference_dump.py: not a reference fixture"), open(os.path.join(d, "meta.json"), "w"))


if __name__ == "__main__":
    from oracle import pyoracle
    write_dump(pyoracle, sys.argv[1], sys.argv[2])
