"""Pins the CPU oracle against REAL rs_pbrt output, when such output exists.

The reference ships no tests or golden vectors for the render path and this image has no Rust toolchain, so the
oracle is "parity unpinned" by the reference until someone with cargo runs rust_shim/refdump.rs once per scene of
tests/golden/ref_scenes/ (see oracle/REFERENCE_FIXTURES.md) and commits tests/golden/ref_<scene>.npz.  Every fixture found is
checked; with none, the test SKIPS and says so (it never passes vacuously)."""
import glob
import json
import os

import numpy as np
import pytest

from rs_pbrt_amd import abi, scenes

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "ref_*.npz")))


def test_scene_files_for_the_reference_are_current():
    """tests/golden/ref_scenes/*.pbrt are what tools/export_pbrt.py writes today (so a fixture made from them matches scenes.py)"""
    import subprocess
    import sys
    d = os.path.join(HERE, "golden", "ref_scenes")
    before = {f: open(os.path.join(d, f), "rb").read() for f in os.listdir(d)}
    subprocess.check_call([sys.executable, os.path.join(HERE, "..", "tools", "export_pbrt.py")], stdout=subprocess.DEVNULL)
    after = {f: open(os.path.join(d, f), "rb").read() for f in os.listdir(d)}
    assert before == after and any(f.endswith(".pbrt") for f in after)


def test_exported_texture_directives_follow_make_texture(oracle):
    """the procedural Cornell variant as text for rs_pbrt: float / spectrum namespaces by use, the dots argument swap of api.rs:1228 /
    :1531, a 3-D mapping's world_to_texture as the CTM (Transform reads columns), materials referring to textures by name — and the
    scene itself renders in the oracle with every texture class contributing"""
    text = open(os.path.join(HERE, "golden", "ref_scenes", "cornell_procedural.pbrt")).read()
    for cls in ("constant", "scale", "mix", "checkerboard", "dots", "fbm", "wrinkled", "windy", "marble"):
        assert '" "%s"' % cls in text, cls
    assert 'Texture "tex3" "float" "windy"' in text and '"spectrum" "marble"' in text and '"float" "checkerboard"' not in text
    dots = [l for l in text.splitlines() if '"spectrum" "dots"' in l][0]
    assert dots.index('"rgb inside" [0.629999995') < dots.index('"rgb outside" [0.725000024')   # scenes: dots_texture(outside = red, inside = white)
    assert "Transform [0.0199999996 0 0 0 0 0.0199999996 0 0 0 0 0.0199999996 0 0 0 0 1]" in text
    assert 'Material "matte" "texture Kd" "tex2" "float sigma" [0] "texture bumpmap" "tex5"' in text
    assert '"texture roughness" "tex' in text and '"texture uroughness" "tex' in text and '"float uv" [0 0 1 0 1 1 0 1]' in text
    from rs_pbrt_amd import lib
    sc = scenes.cornell_box(lib.bvh_build, "procedural")
    a = oracle.render(sc, scenes.cornell_render_desc(res=32, spp=4), threads=4)
    b = oracle.render(scenes.cornell_box(lib.bvh_build, "matte"), scenes.cornell_render_desc(res=32, spp=4), threads=4)
    assert a["counters"]["nan_samples"] == 0 and np.isfinite(a["film"]).all()
    assert np.array_equal(a["film"][:, 3], b["film"][:, 3]) and not np.allclose(a["film"][:, :3], b["film"][:, :3], atol=1e-3)


def test_exported_image_textures_carry_the_generators_texels(oracle):
    """cornell_imagemap: the PNGs next to the scene file decode (8-bit RGB, stored) to the bytes the generator divided by 255, every
    imagemap directive spells out gamma false / scale / filter / wrap / mapping, float images are declared in the float namespace"""
    import struct
    import zlib
    d = os.path.join(HERE, "golden", "ref_scenes")
    text = open(os.path.join(d, "cornell_imagemap.pbrt")).read()
    lines = [l for l in text.splitlines() if '"imagemap"' in l]
    assert len(lines) == 7 and all('"bool gamma" ["false"]' in l and '"float maxanisotropy" [8]' in l for l in lines)
    assert sum('"float" "imagemap"' in l for l in lines) == 2 and {w for l in lines for w in ("repeat", "clamp", "black") if '["%s"]' % w in l} == {"repeat", "clamp", "black"}
    from rs_pbrt_amd import lib
    sc = scenes.cornell_box(lib.bvh_build, "imagemap")
    for k, src in sc.builder.image_src.items():
        raw = open(os.path.join(d, "cornell_imagemap_img%d.png" % k), "rb").read()
        assert raw[:8] == b"\x89PNG\r\n\x1a\n"
        w, h, depth, ctype = struct.unpack(">IIBB", raw[16:26])
        i = raw.index(b"IDAT")
        n = struct.unpack(">I", raw[i - 4:i])[0]
        px = np.frombuffer(zlib.decompress(raw[i + 4:i + 4 + n]), np.uint8).reshape(h, 1 + 3 * w)[:, 1:].reshape(h, w, 3)
        assert (depth, ctype) == (8, 2) and np.array_equal(px, src["u8"])
    a = oracle.render(sc, scenes.cornell_render_desc(res=32, spp=4), threads=4)
    assert a["counters"]["nan_samples"] == 0 and np.isfinite(a["film"]).all() and a["film"][:, :3].max() > 0.1


@pytest.mark.skipif(not FIXTURES, reason="parity unpinned: no tests/golden/ref_*.npz (made from real rs_pbrt by rust_shim/refdump.rs) is committed yet")
@pytest.mark.parametrize("path", FIXTURES or ["-"])
def test_oracle_against_rs_pbrt_output(oracle, path):
    assert "FABRICATED" not in str(np.load(path, allow_pickle=False)["meta"]), "a fabricated dump (tools/fake_reference_dump.py) was committed as a fixture"
    check_fixture(oracle, path)


@pytest.mark.parametrize("name", ["cornell_mixed", "instanced_room", "instanced_moving", "cornell_fog_volpath", "cornell_02sequence", "cornell_directlighting", "cornell_ao", "sky_blocks", "cornell_imagemap", "cornell_gaussian", "alpha_cutouts"])
def test_fixture_pipeline_end_to_end_with_a_fabricated_dump(oracle, tmp_path, name):
    """NOT a pin: the oracle's own output written in refdump.rs's file layout, packed by tools/ref_to_npz.py and run through the very
    checks a real fixture gets — so that the day a dump from rs_pbrt arrives, a failure means the oracle, not the plumbing"""
    import sys
    sys.path.insert(0, os.path.join(HERE, "..", "tools"))
    from fake_reference_dump import write_dump
    from ref_to_npz import pack
    write_dump(oracle, name, str(tmp_path / "dump"))
    dst = str(tmp_path / ("ref_%s.npz" % name))
    pack(str(tmp_path / "dump"), name, dst)
    check_fixture(oracle, dst)


def render_as(oracle, sc, rd, extra):
    """the oracle's render for a scene of export_pbrt.SCENES (DirectLightingIntegrator has its own entry point there)"""
    if extra.get("integrator") == "directlighting":
        return oracle.render_integrator(sc, rd, "direct", strategy=extra.get("direct_strategy", "all"), threads=4, want_li=True)
    return oracle.render(sc, rd, threads=4, want_li=True)


def check_fixture(oracle, path):
    import sys
    sys.path.insert(0, os.path.join(HERE, "..", "tools"))
    from export_pbrt import EXTRA, SCENES, camera_of, render_kwargs
    from rs_pbrt_amd import lib
    z = np.load(path, allow_pickle=False)
    name = str(z["name"]); meta = json.loads(str(z["meta"]))
    mk, _cam, xres, yres, spp, depth = SCENES[name]
    sc = mk(lib.bvh_build, scenes)
    # 1. BVHAccel::new: the flattened node array, bit for bit, and the primitive order (by vertex positions)
    # (the dump holds the top-level aggregate; a TransformedPrimitive's row is NaN, refdump.rs tri_vertices)
    nt_nodes, nt_prims = sc.n_top
    if name == "instanced_moving":
        # the boxes of ROTATING moving instances come from rspt_motion_bounds, whose derivative coefficients are the closed form of the reference's expanded
        # polynomials (csrc/motion_bounds.h): equal to a few ulps of the box size, not bit for bit (tests/test_motion_bounds.py) — the tree's shape, leaf
        # contents and split axes must still be the reference's, and every box equal within 4e-6 of the scene's extent
        mine_n, ref_n = sc.nodes[:nt_nodes], z["bvh_nodes"]
        assert len(mine_n) == len(ref_n) and all(np.array_equal(mine_n[k], ref_n[k]) for k in ("offset", "n_prims", "axis"))
        ext = float((ref_n["bmax"][0] - ref_n["bmin"][0]).max())
        assert np.abs(mine_n["bmin"] - ref_n["bmin"]).max() <= 4e-6 * ext and np.abs(mine_n["bmax"] - ref_n["bmax"]).max() <= 4e-6 * ext
    else:
        assert sc.nodes[:nt_nodes].tobytes() == z["bvh_nodes"].tobytes()
    inst = np.isnan(z["bvh_prims"]).any(axis=1)
    assert np.array_equal(inst, sc.prims["mesh"][:nt_prims] == abi.MESH_INSTANCE)
    mine = sc.P[sc.prims["v"][:nt_prims]].reshape(-1, 9)
    assert np.array_equal(mine[~inst], z["bvh_prims"][~inst])
    # 2. Scene::intersect / intersect_p on the committed rays
    if "hits" in z:
        rays = np.fromfile(os.path.join(HERE, "golden", "ref_scenes", "rays.bin"), abi.RAY_DT)
        h = oracle.trace(sc, rays)
        ref = z["hits"]
        assert np.array_equal(h["prim"] != abi.MISS, ref[:, 0] == 1.0)
        hit = ref[:, 0] == 1.0
        assert np.array_equal(h["t"][hit], ref[hit, 1])
        known = hit & ~np.isnan(ref[:, 8:17]).any(axis=1)   # an instanced hit has lost its primitive in v0.9.12 (transform.rs:856)
        assert np.array_equal(sc.P[sc.prims["v"][h["prim"][known]]].reshape(-1, 9), ref[known, 8:17])
        assert np.array_equal(oracle.trace(sc, rays, any_hit=True)["prim"] == 0, z["occluded"] == 1)
    # 3. the frame: filter weights exact; radiance per camera sample bit for bit where the dump has it, else film RMSE
    look_at, fov = camera_of(name, scenes)
    rd = scenes.make_render_desc(xres, yres, spp, look_at, fov, max_depth=depth, **render_kwargs(name, scenes))
    assert list(rd.crop_px) == meta["crop_px"] and list(rd.sample_bounds) == meta["sample_bounds"] and int(rd.spp) == meta["spp"]
    r = render_as(oracle, sc, rd, EXTRA.get(name, {}))
    if "filter" in EXTRA.get(name, {}):   # unequal weights: a pixel's sum depends on the order its tiles were merged in (film.rs:346-371)
        assert np.allclose(r["film"][:, 3], z["film"][:, 3], rtol=1e-5)
    else:
        assert np.array_equal(r["film"][:, 3], z["film"][:, 3])
    a, b = scenes.film_to_rgb(r["film"]), scenes.film_to_rgb(z["film"])
    assert np.sqrt(np.mean((a.astype(np.float64) - b) ** 2)) < 1e-6
    if "li" in z:
        li = z["li"]
        px, py, sn = li[:, 0].astype(int), li[:, 1].astype(int), li[:, 2].astype(int)
        w = rd.crop_px[2] - rd.crop_px[0]
        mine_li = r["li"][(py - rd.crop_px[1]) * w + (px - rd.crop_px[0]), sn]
        same = (mine_li == li[:, 5:8]).all(-1)
        assert same.mean() > 0.999, "per-sample radiance differs from rs_pbrt in %.3f %% of the samples" % (100 * (1 - same.mean()))
