"""The oracle against REAL rs_pbrt output: the two renders of the Cornell box that ship in the reference's documentation
(docs/source/cornell_box_{8,256}_pixelsamples.png, `Sampler "sobol"` 8 / 256 spp, `Integrator "path"`, getting_started.rst:150-209;
their pixels, unchanged, are tests/golden/reference_cornell_docs.npz).  The scene FILE of those renders is not in the reference tree;
scenes.cornell_box_docs is that scene as recovered from the images (tools/recover_cornell_docs.py) — so the claim these tests hold is:
"there is a scene description of a few round numbers under which the oracle reproduces the reference's image byte for byte in 94 % of the
pixels of the 8-spp render (all three channels; 98.5 % within 4 / 255) and in 99 % of the pixels of the 256-spp render within 1 / 255, and
every single ingredient of that description and of the oracle's sampling sits on a sharp optimum of the agreement".  A pixel of the 8-spp
image is the mean of 8 paths of up to 5 bounces whose every vertex depends on the Sobol' index of the pixel, the generator matrices, the
camera ray, the triangle and BVH code, the shading frame, the cosine sampling, the spatial light distribution's choice between the
emitter's two triangles, the triangle sampling, the shadow rays, the MIS weights, the film's box filter and write_image's gamma — so a byte
match is a sample-for-sample match of all of those (what it does not cover: every other material, texture, light, sampler, integrator,
media, instancing: for those the oracle is still pinned by first-principles tests only).
The 6 % of pixels that differ are scattered evenly (no face, edge or shadow stands out), 1.7 % by one byte step, 4.2 % by more — and they
are ONE-SIDED: the reference is the brighter one in all of them (the last test of this file).  It holds 0.33 % more energy than the oracle,
in single samples (about one path in 130) that carry an extra contribution of the size of an ordinary light sample, often of a pure wall
colour (a path that has been to the red or the green wall).  0.33 % is what this scene's paths carry beyond five bounces (with
`maxdepth` 100 the oracle's total equals the reference's to 4e-5) — but none of the ways to let paths run on that were tried (maxdepth
6 .. 100, Russian roulette by luminance / from the throughput of the previous bounce / from bounce 3 or 5 / with its sample from any
other Sobol' dimension up to 100) puts the extra energy into the pixels where the reference has it: each LOWERS the share of equal pixels.  Every
scalar of the scene is a strict optimum at steps of 0.005 units / 0.0002 rad, so it is not the recovery's resolution.  Read the other way round: of the 94.2 % of pixels without such an extra, 99.85 % are
byte-identical.  Open: an rs_pbrt of another vintage behind the documentation's pictures (a roulette fed from outside the sampler would make
94 % the ceiling), or something this oracle (and the GPU with it) still gets wrong.
"""
import os

import numpy as np
import pytest

from rs_pbrt_amd import abi, scenes

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_cornell_docs.npz"))
THREADS = os.cpu_count() or 8


def to_u8(rgb):
    """Film::write_image (film.rs:465-520): gamma_correct, clamp(255 v + 0.5, 0, 255) as u8"""
    a = np.asarray(rgb, np.float32)
    a = np.where(a <= 0.0031308, 12.92 * a, 1.055 * np.power(np.maximum(a, 0), np.float32(1 / 2.4)) - 0.055)
    return np.clip(255.0 * a + 0.5, 0, 255).astype(np.int32)


def agreement(film, ref_u8):
    """shares of pixels whose largest byte difference over the three channels is 0 / <= 1 / <= 4"""
    d = np.abs(to_u8(scenes.film_to_rgb(film)).reshape(ref_u8.shape) - ref_u8.astype(np.int32)).max(-1)
    return float((d == 0).mean()), float((d <= 1).mean()), float((d <= 4).mean())


def test_oracle_reproduces_the_references_8spp_png(oracle):
    sc = scenes.cornell_box_docs(oracle.bvh_build)
    r = oracle.render(sc, scenes.cornell_docs_render_desc(8), threads=THREADS)
    exact, w1, w4 = agreement(r["film"], G["spp8"])
    assert exact > 0.94 and w1 > 0.957 and w4 > 0.985, (exact, w1, w4)   # measured: 0.9408, 0.9579, 0.9853 (the strict bar is the last test of this file)


@pytest.mark.parametrize("what, bar", [("mirrored world instead of mirrored camera", 0.25), ("uniform light choice", 0.40), ("power light choice", 0.40),
                                       ("maxdepth 4", 0.60), ("halton", 0.15), ("L = 100.5", 0.60), ("walls 0.402", 0.65), ("fov + 0.004", 0.93)])
def test_what_the_byte_agreement_is_sensitive_to(oracle, what, bar):
    """The same comparison with one ingredient changed: the share of byte-equal pixels (0.94) falls below `bar`.  Measured: the same picture
    from a mirrored world (other BSDF frames: ts = cross(ns, ss)) 0.16; Distribution1D over the lights instead of SpatialLightDistribution
    0.31; maxdepth 4 instead of 5 0.50; Halton 0.09; half a percent more radiance or albedo 0.50 / 0.54; fov 39.150 instead of 39.146 0.91."""
    kw, scene_kw, mirror_world = {}, {}, False
    if what.startswith("mirrored world"):
        mirror_world = True
    elif what.endswith("light choice"):
        kw["light_strategy"] = abi.LIGHTS_UNIFORM if what.startswith("uniform") else abi.LIGHTS_POWER
    elif what == "maxdepth 4":
        kw["max_depth"] = 4
    elif what == "halton":
        kw["sampler"] = "halton"
    elif what == "L = 100.5":
        scene_kw["L"] = 100.5
    elif what == "walls 0.402":
        scene_kw["white"] = 0.402
    sc = scenes.cornell_box_docs(oracle.bvh_build, mirror_world=mirror_world, **scene_kw)
    rd = scenes.cornell_docs_render_desc(8, mirror_camera=not mirror_world, **kw)
    if what == "fov + 0.004":
        rd.raster_to_camera[:] = scenes.make_render_desc(500, 500, 8, scenes.CORNELL_DOCS_LOOK_AT, scenes.CORNELL_DOCS_FOV + 0.004).raster_to_camera[:]
    exact, _, _ = agreement(oracle.render(sc, rd, threads=THREADS)["film"], G["spp8"])
    assert exact < bar, (what, exact)


def test_oracle_reproduces_the_references_256spp_png_in_every_sixth_tile(oracle):
    """256 paths per pixel; every sixth 16 x 16 tile of the Morton order (shard 0 of 6 — a crop window would not do: the Sobol' index of a pixel
    depends on the sample bounds, sobol.rs:40-60).  Measured on the whole frame (35 s on 8 cores): 0.710 equal, 0.989 within 1, 1.000 within 4."""
    sc = scenes.cornell_box_docs(oracle.bvh_build)
    r = oracle.render(sc, scenes.cornell_docs_render_desc(256, shard=(0, 6, 1)), threads=THREADS)
    mine = r["film"][:, 3] > 0
    assert 0.15 < mine.mean() < 0.18
    d = np.abs(to_u8(scenes.film_to_rgb(r["film"])) - G["spp256"].reshape(-1, 3).astype(np.int32)).max(-1)[mine]
    exact, w1, w4 = float((d == 0).mean()), float((d <= 1).mean()), float((d <= 4).mean())
    assert exact > 0.65 and w1 > 0.98 and w4 > 0.998, (exact, w1, w4)   # measured on these tiles: 0.703, 0.989, 0.9992


def test_outside_the_references_extra_contributions_every_pixel_is_byte_identical(oracle):
    """The strict form of the pin.  The reference's 8-spp picture carries extra, strictly positive contributions in about
    one path of 130 that v0.9.12's source does not explain (module docstring; experiments/reference_pin/README.md) — wherever it does NOT
    (no byte of the reference above the oracle's), the two must agree byte for byte: measured 99.85 % of those pixels, the rest one byte step
    with the oracle the brighter one (0.14 % of the picture).  A regression anywhere in the sampler / camera / traversal / shading / light
    selection / film shows up two-sided and fails this at once (every ingredient test above falls to 9 - 69 % exact)."""
    sc = scenes.cornell_box_docs(oracle.bvh_build)
    ours = to_u8(scenes.film_to_rgb(oracle.render(sc, scenes.cornell_docs_render_desc(8), threads=THREADS)["film"])).reshape(500, 500, 3)
    ref = G["spp8"].astype(np.int32)
    no_extra = (ref - ours).max(-1) <= 0
    exact = (ours == ref).all(-1)
    assert no_extra.mean() > 0.94                                   # measured 0.9422
    assert exact[no_extra].mean() >= 0.998, exact[no_extra].mean()   # measured 0.9985
    assert ((ours - ref).max(-1) > 1).mean() < 0.0005               # the oracle above the reference by more than one byte step: 0.0001 of the pixels
