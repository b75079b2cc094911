"""-m gpu: the whole path (raygen -> trace -> shade -> film) through rspt_render* against the
oracle's restatement of SamplerIntegrator::render + PathIntegrator::li.

Tolerances: since glibc_libm.h restates the host libm's sinf / cosf / logf / log2f / expf / acosf / atan2f exactly, every camera sample's
radiance is bit-identical to the oracle's (test_radiance_of_every_sample_is_bit_identical_to_the_oracle); the per-test bars below (shares of
identical samples, RMSE 1e-5 ... 3e-4) date from when the device library's last-ulp differences were still in the path and are kept as
they are — they are all met with zero difference now.  The film differs from the oracle's only in the order in which splats from
neighbouring tiles are added (atomics).
"""
import os

import numpy as np
import pytest

from rs_pbrt_amd import abi, scenes
from tests.util import GALLERY_LOOK_AT, film_rmse, gallery, small_soup

pytestmark = pytest.mark.gpu
THREADS_ALL = __import__("os").cpu_count() or 8
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _render_pair(gpu, oracle, sc, rd, want_li=True):
    ds = gpu.DeviceScene(sc)
    try:
        film, st = gpu.render(ds, rd)
        li = gpu.render_samples(ds, rd)[0] if want_li else None
    finally:
        ds.close()
    ref = oracle.render(sc, rd, threads=8, want_li=want_li)
    if want_li:   # THE bar of these tests: every sample's radiance, bit for bit (glibc_libm.h).  Films are compared after it by RMSE only because
                  # a pixel's f32 sum over its samples is taken in another order (weights, being sums of equal terms, are compared bit for bit)
        assert np.array_equal(li, ref["li"]), "per-sample radiance differs from the oracle in %d of %d samples" % (int((li != ref["li"]).any(axis=2).sum()), li.shape[0] * li.shape[1])
    return film, li, st, ref


@pytest.mark.parametrize("variant", ["matte", "mixed", "rough", "procedural", "imagemap", "layered"])
def test_cornell_matches_oracle(gpu, oracle, variant):
    sc = scenes.cornell_box(gpu.bvh_build, variant=variant)
    rd = scenes.cornell_render_desc(res=64, spp=16)
    film, li, st, ref = _render_pair(gpu, oracle, sc, rd)
    assert np.array_equal(film[:, 3], ref["film"][:, 3])
    assert film_rmse(film, ref["film"]) < 1e-5
    assert st["nan_samples"] == ref["counters"]["nan_samples"] == 0
    assert st["samples"] == ref["counters"]["samples"] == 64 * 64 * 16


def test_soup_matches_oracle_depth8(gpu, oracle):
    sc = small_soup(gpu.bvh_build)
    rd = scenes.soup_render_desc(res=96, spp=8, max_depth=8)
    film, li, st, ref = _render_pair(gpu, oracle, sc, rd)
    assert np.array_equal(film[:, 3], ref["film"][:, 3])
    assert film_rmse(film, ref["film"]) < 1e-5


@pytest.mark.parametrize("strategy", [abi.LIGHTS_UNIFORM, abi.LIGHTS_POWER, abi.LIGHTS_SPATIAL])
def test_light_strategies(gpu, oracle, strategy):
    sc = scenes.cornell_box(gpu.bvh_build)
    rd = scenes.cornell_render_desc(res=48, spp=8, light_strategy=strategy)
    film, _, _, ref = _render_pair(gpu, oracle, sc, rd, want_li=False)
    assert film_rmse(film, ref["film"]) < 1e-5


def test_smooth_normals_plastic(gpu, oracle):
    """interpolated shading normals (has_n) + plastic (Lambert + microfacet), three area lights"""
    sc = scenes.statue_standin(gpu.bvh_build, grid=96)
    rd = scenes.statue_render_desc(xres=96, yres=54, spp=8)
    film, li, st, ref = _render_pair(gpu, oracle, sc, rd)
    assert np.array_equal(film[:, 3], ref["film"][:, 3])
    assert film_rmse(film, ref["film"]) < 1e-5


def test_thin_lens_and_gaussian_filter(gpu, oracle):
    """depth of field (lens samples, concentric disk) and a radius-2 filter: every sample splats into
    up to 16 pixels through the atomic path, so the film tolerance is f32 re-association only."""
    sc = scenes.cornell_box(gpu.bvh_build)
    rd = scenes.cornell_render_desc(res=48, spp=8, lens_radius=8.0, focal_distance=1000.0, filter_radius=(2.0, 2.0),
                                    filter_table=scenes.gaussian_filter_table((2.0, 2.0)))
    film, li, st, ref = _render_pair(gpu, oracle, sc, rd)
    assert np.allclose(film[:, 3], ref["film"][:, 3], rtol=1e-5)
    assert film_rmse(film, ref["film"]) < 1e-5


@pytest.mark.parametrize("case", ["gaussian-2", "anisotropic", "wide-4", "crop-shards-ranges", "02sequence", "halton-clamped"])
def test_wide_pixel_filters_gather_form(gpu, oracle, case, monkeypatch):
    """filters wider than a pixel take the gather form of the film stage (kernels.h k_film_gather: every pixel sums what reaches it, no atomics): against the
    oracle's film, against the atomic form (RSPT_FILM_GATHER=0), per-sample radiance through the same kernel, shards / sample slices / crop windows adding up"""
    sc = scenes.cornell_box(gpu.bvh_build)
    kw = dict(res=56, spp=8)
    if case == "gaussian-2":
        kw.update(filter_radius=(2.0, 2.0), filter_table=scenes.gaussian_filter_table((2.0, 2.0)))
    elif case == "anisotropic":
        kw.update(filter_radius=(1.5, 2.75), filter_table=scenes.gaussian_filter_table((1.5, 2.75), alpha=1.0))
    elif case == "wide-4":
        kw.update(filter_radius=(4.0, 3.5), filter_table=scenes.gaussian_filter_table((4.0, 3.5), alpha=0.5))
    elif case == "crop-shards-ranges":
        kw.update(filter_radius=(2.0, 2.0), filter_table=scenes.gaussian_filter_table((2.0, 2.0)), crop=(0.2, 0.83, 0.1, 0.7))
    elif case == "02sequence":
        kw.update(filter_radius=(2.0, 2.0), filter_table=scenes.gaussian_filter_table((2.0, 2.0)), sampler="02sequence")
    elif case == "halton-clamped":
        kw.update(spp=6, filter_radius=(2.5, 2.5), filter_table=scenes.gaussian_filter_table((2.5, 2.5)), sampler="halton", max_sample_luminance=1.5)
    rd = scenes.cornell_render_desc(**kw)
    ref = oracle.render(sc, rd, threads=8, want_li=case != "02sequence")
    films = {}
    with gpu.DeviceScene(sc) as ds:
        for form in ("1", "0"):
            monkeypatch.setenv("RSPT_FILM_GATHER", form)
            film, st = gpu.render(ds, rd)
            assert st["samples"] == ref["counters"]["samples"] and st["nan_samples"] == ref["counters"]["nan_samples"]
            assert np.allclose(film[:, 3], ref["film"][:, 3], rtol=1e-5) and film_rmse(film, ref["film"]) < 1e-5, form
            films[form] = film
            if case != "02sequence":
                assert np.array_equal(gpu.render_samples(ds, rd)[0], ref["li"]), form
        assert np.allclose(films["1"], films["0"], rtol=2e-5, atol=1e-6)
        monkeypatch.setenv("RSPT_FILM_GATHER", "1")
        again, _ = gpu.render(ds, rd)
        assert np.array_equal(again, films["1"])   # the gather form adds in a fixed order: the same film every time
        if case == "gaussian-2":   # several batches per frame (pixels and sample slices): a block's halo pixels then sit in other batches, which add their share when they run
            monkeypatch.setenv("RSPT_BATCH", "4096")
            split, st2 = gpu.render(ds, rd)
            monkeypatch.delenv("RSPT_BATCH")
            assert st2["samples"] == st["samples"] and np.allclose(split, films["1"], rtol=2e-5, atol=1e-6)
        if case == "crop-shards-ranges":
            total = np.zeros_like(films["1"])
            for r in range(3):
                for rng in ((0, 3), (3, 5)):
                    f, _ = gpu.render(ds, scenes.cornell_render_desc(shard=(r, 3, 1), sample_range=rng, **kw))
                    total += f
            assert np.allclose(total, films["1"], rtol=2e-5, atol=1e-6)


def test_wide_filter_under_a_pixel_sampler_in_several_passes(gpu, oracle, monkeypatch):
    """the pixel samplers hand their samples to the film stage pass by pass (RSPT_SERIAL_SAMPLES rows of pixels at a time), each pass with its own pixel list:
    the gather form rebuilds its position index per pass — the frame must not depend on how many passes there were"""
    sc = scenes.cornell_box(gpu.bvh_build)
    rd = scenes.cornell_render_desc(res=64, spp=4, sampler="02sequence", filter_radius=(2.0, 2.0), filter_table=scenes.gaussian_filter_table((2.0, 2.0)))
    ref = oracle.render(sc, rd, threads=8)
    with gpu.DeviceScene(sc) as ds:
        one, _ = gpu.render(ds, rd)
        monkeypatch.setenv("RSPT_SERIAL_SAMPLES", str(16 * 4 * 16 * 3))   # three rows of pixels per pass
        many, _ = gpu.render(ds, rd)
    assert np.allclose(one[:, 3], ref["film"][:, 3], rtol=1e-5) and film_rmse(one, ref["film"]) < 1e-5
    assert np.allclose(many, one, rtol=2e-5, atol=1e-6)


def test_crop_window_and_non_square(gpu, oracle):
    sc = scenes.cornell_box(gpu.bvh_build)
    rd = scenes.make_render_desc(100, 60, 4, scenes.CORNELL_LOOK_AT, scenes.CORNELL_FOV, crop=(0.25, 0.8, 0.1, 0.9))
    film, _, st, ref = _render_pair(gpu, oracle, sc, rd, want_li=False)
    assert film.shape == ref["film"].shape
    assert np.array_equal(film[:, 3], ref["film"][:, 3])
    assert film_rmse(film, ref["film"]) < 1e-5


def test_null_material_and_no_lights(gpu, oracle):
    """a primitive without material is passed through (path.rs:109-116); a scene without lights is black"""
    sc = scenes.cornell_box(gpu.bvh_build)
    sc.prims["material"][sc.prims["material"] == 1] = abi.NO_MATERIAL  # red wall becomes a null boundary
    rd = scenes.cornell_render_desc(res=48, spp=4)
    film, li, st, ref = _render_pair(gpu, oracle, sc, rd)
    assert film_rmse(film, ref["film"]) < 1e-5
    sb = scenes.SceneBuilder()
    m = sb.add_material(scenes.matte((0.5, 0.5, 0.5)))
    sb.add_quad([(0, 0, 0), (0, 0, 559), (549, 0, 559), (549, 0, 0)], m)
    dark = sb.finish(gpu.bvh_build)
    film, _, _, ref = _render_pair(gpu, oracle, dark, rd, want_li=False)
    assert np.array_equal(film, ref["film"]) and not film[:, :3].any()


def test_tile_shards_sum_to_full_frame(gpu):
    """multi-GPU decomposition on one device: the films of the Morton-tile shards add up to the
    unsharded film (exactly: with the box filter every pixel's own samples come from one shard)."""
    sc = scenes.cornell_box(gpu.bvh_build)
    ds = gpu.DeviceScene(sc)
    try:
        full, _ = gpu.render(ds, scenes.cornell_render_desc(res=80, spp=4))
        acc = np.zeros_like(full)
        n = 0
        for r in range(3):
            f, st = gpu.render(ds, scenes.cornell_render_desc(res=80, spp=4, shard=(r, 3, 2)))
            acc += f
            n += st["samples"]
    finally:
        ds.close()
    assert n == 80 * 80 * 4
    assert np.allclose(acc, full, rtol=1e-6, atol=1e-7)
    assert np.array_equal(acc[:, 3], full[:, 3])


def test_batching_is_invisible(gpu, monkeypatch):
    """the film must not depend on how samples are cut into wavefront batches"""
    sc = scenes.cornell_box(gpu.bvh_build)
    rd = scenes.cornell_render_desc(res=64, spp=16)
    ds = gpu.DeviceScene(sc)
    try:
        a, _ = gpu.render(ds, rd)
        monkeypatch.setenv("RSPT_BATCH", "5000")
        b, _ = gpu.render(ds, rd)
        monkeypatch.setenv("RSPT_BATCH", "70000")
        c, _ = gpu.render(ds, rd)
    finally:
        ds.close()
    assert np.array_equal(a[:, 3], b[:, 3])
    assert np.allclose(a, b, rtol=1e-6, atol=1e-7) and np.allclose(a, c, rtol=1e-6, atol=1e-7)


def test_counters_match_oracle(gpu, oracle, monkeypatch):
    """node / triangle / ray counts that feed the roofline's algorithmic bytes agree with the oracle's
    (not bit-equal: a last-ulp sin/cos difference can change an individual path)"""
    monkeypatch.setenv("RSPT_COUNTERS", "1")
    sc = small_soup(gpu.bvh_build)
    rd = scenes.soup_render_desc(res=64, spp=8)
    film, _, st, ref = _render_pair(gpu, oracle, sc, rd, want_li=False)
    c = ref["counters"]
    for k_gpu, k_ref in (("nodes_visited", "nodes_visited"), ("tris_tested", "tris_tested"), ("rays_closest", "rays_closest"), ("rays_any", "rays_any")):
        assert abs(st[k_gpu] - c[k_ref]) <= 1e-3 * c[k_ref] + 8, (k_gpu, st[k_gpu], c[k_ref])
    assert st["alg_bytes"] > 0


def test_golden_cornell_film(gpu):
    """committed oracle output (tests/golden/make_golden.py) — no oracle build needed on the box"""
    g = np.load(os.path.join(GOLDEN, "cornell_matte_32x32x8.npz"))
    sc = scenes.cornell_box(gpu.bvh_build)
    rd = scenes.cornell_render_desc(res=32, spp=8)
    ds = gpu.DeviceScene(sc)
    try:
        film, _ = gpu.render(ds, rd)
        li, _ = gpu.render_samples(ds, rd)
    finally:
        ds.close()
    assert np.array_equal(film[:, 3], g["film"][:, 3])
    assert film_rmse(film, g["film"]) < 1e-5
    assert np.array_equal(li, g["li"])   # the committed fixture (= today's oracle, tests/test_oracle_kat.py), every sample bit for bit


@pytest.mark.parametrize("lights", ["all", "delta", "area"])
def test_gallery_remaining_materials_and_delta_lights(gpu, oracle, lights):
    """substrate (FresnelBlend), uber with opacity (specular transmission + Lambert + microfacet +
    specular lobes), translucent (Lambertian / microfacet transmission), rough glass, Oren-Nayar;
    point / spot / distant lights next to an area light"""
    from tests.util import GALLERY_LOOK_AT, gallery
    sc = gallery(gpu.bvh_build, lights)
    rd = scenes.make_render_desc(80, 60, 16, GALLERY_LOOK_AT, 60, max_depth=6)
    film, li, st, ref = _render_pair(gpu, oracle, sc, rd)
    assert np.array_equal(film[:, 3], ref["film"][:, 3])
    assert film_rmse(film, ref["film"]) < 2e-5
    assert st["nan_samples"] == ref["counters"]["nan_samples"]


@pytest.mark.parametrize("strategy", [abi.LIGHTS_POWER, abi.LIGHTS_SPATIAL])
def test_gallery_light_strategies_with_delta_lights(gpu, oracle, strategy):
    from tests.util import GALLERY_LOOK_AT, gallery
    sc = gallery(gpu.bvh_build, "all")
    rd = scenes.make_render_desc(64, 48, 8, GALLERY_LOOK_AT, 60, max_depth=4, light_strategy=strategy)
    film, _, _, ref = _render_pair(gpu, oracle, sc, rd, want_li=False)
    assert film_rmse(film, ref["film"]) < 2e-5


@pytest.mark.parametrize("variant", ["matte", "mixed"])
def test_halton_sampler_matches_oracle(gpu, oracle, variant):
    """HaltonSampler (the reference's default sampler, api.rs:526): index mapping, scrambled radical
    inverses with the PCG-shuffled digit permutations, non-power-of-two spp"""
    sc = scenes.cornell_box(gpu.bvh_build, variant=variant)
    rd = scenes.cornell_render_desc(res=72, spp=12, sampler="halton")
    film, li, st, ref = _render_pair(gpu, oracle, sc, rd)
    assert np.array_equal(film[:, 3], ref["film"][:, 3])
    assert film_rmse(film, ref["film"]) < 1e-5
    assert st["samples"] == 72 * 72 * 12


def test_halton_pixel_center_lens_and_wide_frame(gpu, oracle):
    from tests.util import GALLERY_LOOK_AT, gallery
    sc = gallery(gpu.bvh_build, "all")
    rd = scenes.make_render_desc(200, 150, 5, GALLERY_LOOK_AT, 60, max_depth=4, sampler="halton", sample_at_pixel_center=True, lens_radius=0.05,
                                 focal_distance=6.0)  # > 128 pixels wide: pixel index wraps modulo K_MAX_RESOLUTION
    film, _, _, ref = _render_pair(gpu, oracle, sc, rd, want_li=False)
    assert np.array_equal(film[:, 3], ref["film"][:, 3])
    assert film_rmse(film, ref["film"]) < 2e-5


@pytest.mark.parametrize("kind,with_area", [("constant", False), ("map", False), ("map", True)])
def test_infinite_area_light(gpu, oracle, kind, with_area):
    """K5 shade_miss + InfiniteAreaLight sample_li / pdf_li / le: MIP-map lookups, Distribution2D
    importance sampling, MIS term applied when the BSDF-sampled ray escapes.  acos / atan2 / sin / cos
    of the lat-long mapping differ in the last ulp between libm and the device, hence the looser
    per-sample bar; the film bar is the usual one."""
    from tests.util import SKY_LOOK_AT, sky_scene
    sc = sky_scene(gpu.bvh_build, kind, with_area)
    rd = scenes.make_render_desc(80, 60, 16, SKY_LOOK_AT, 50, max_depth=5)
    film, li, st, ref = _render_pair(gpu, oracle, sc, rd)
    assert np.array_equal(film[:, 3], ref["film"][:, 3])
    assert film_rmse(film, ref["film"]) < (2e-5 if kind == "constant" else 2e-4)
    assert st["nan_samples"] == ref["counters"]["nan_samples"] == 0


def test_infinite_light_power_and_spatial_strategies(gpu, oracle):
    from tests.util import SKY_LOOK_AT, sky_scene
    sc = sky_scene(gpu.bvh_build, "map", True)
    for strategy in (abi.LIGHTS_POWER, abi.LIGHTS_UNIFORM):
        rd = scenes.make_render_desc(48, 36, 8, SKY_LOOK_AT, 50, max_depth=3, light_strategy=strategy)
        film, _, _, ref = _render_pair(gpu, oracle, sc, rd, want_li=False)
        assert film_rmse(film, ref["film"]) < 2e-4


def test_mix_material(gpu, oracle):
    """MixMaterial lobes with their per-lobe scale (sc_opt) through every lobe kind that can carry one"""
    sb = scenes.SceneBuilder()
    white = sb.add_material(scenes.matte((0.7, 0.7, 0.7)))
    m1 = sb.add_material(scenes.mix(scenes.mirror((0.9, 0.9, 0.9)), scenes.matte((0.7, 0.2, 0.2), sigma=20.0), (0.3, 0.4, 0.5)))
    m2 = sb.add_material(scenes.mix(scenes.glass(), scenes.plastic((0.2, 0.5, 0.3), (0.4, 0.4, 0.4), 0.15), (0.6, 0.6, 0.6)))
    m3 = sb.add_material(scenes.mix(scenes.substrate((0.4, 0.4, 0.1), (0.2, 0.2, 0.2), 0.1, 0.1), scenes.translucent(), (0.5, 0.2, 0.8)))
    q = sb.add_quad
    q([(-5, 0, -5), (-5, 0, 5), (5, 0, 5), (5, 0, -5)], white)
    q([(-5, 0, 5), (-5, 6, 5), (5, 6, 5), (5, 0, 5)], white)
    for i, m in enumerate((m1, m2, m3)):
        x = -3.5 + 2.6 * i
        q([(x, 0.3, 1), (x + 2, 0.3, 1), (x + 2, 3.0, 2.2), (x, 3.0, 2.2)], m)
    q([(-1.5, 5.9, -1.5), (1.5, 5.9, -1.5), (1.5, 5.9, 1.5), (-1.5, 5.9, 1.5)], white, emit=(9, 9, 9))
    sc = sb.finish(gpu.bvh_build)
    rd = scenes.make_render_desc(80, 60, 16, ((0, 3, -5.5), (0, 1.8, 2), (0, 1, 0)), 55, max_depth=6)
    film, li, st, ref = _render_pair(gpu, oracle, sc, rd)
    assert np.array_equal(film[:, 3], ref["film"][:, 3])
    assert film_rmse(film, ref["film"]) < 2e-5


@pytest.mark.parametrize("trilinear,wrap,bump", [(False, "repeat", True), (False, "repeat", False), (True, "repeat", True), (False, "clamp", True), (False, "black", False)])
def test_textured_materials_match_oracle(gpu, oracle, trilinear, wrap, bump):
    """image textures (EWA / trilinear MIP lookups, UV + planar mappings, scale textures), camera-ray
    differentials, bump mapping, lobes dropped where the texture is black.  The first hit (differentials, EWA) agrees
    to 2e-6 with 92 % of samples bit-identical.  At later hits the last-ulp sin/cos differences of the sampled
    directions move the hit point by an ulp; a high-contrast texture turns that into ~1e-5 (bilinear weight x texel
    contrast), and Material::bump's finite differences over du = 0.0005 (material.rs:183-189) amplify it another
    2000x into the shading normal — hence the looser film bar with bump maps (north-star bound: 1e-3)."""
    from tests.util import TEXTURED_LOOK_AT, textured_room
    sc = textured_room(gpu.bvh_build, trilinear=trilinear, wrap=wrap, bump=bump)
    rd = scenes.make_render_desc(96, 72, 16, TEXTURED_LOOK_AT, 45, max_depth=4)
    film, li, st, ref = _render_pair(gpu, oracle, sc, rd)
    assert np.array_equal(film[:, 3], ref["film"][:, 3])
    assert film_rmse(film, ref["film"]) < (1e-4 if bump else 2e-5)
    assert st["nan_samples"] == ref["counters"]["nan_samples"] == 0


def test_procedural_textures_and_mappings_match_oracle(gpu, oracle):
    """checkerboard, dots, mix, fbm, marble, windy, wrinkled, spherical / cylindrical mappings, a three-level texture graph
    and a procedural bump map.  Perlin noise is integer hashing + polynomials (bit-exact); logf of the octave count,
    sinf of marble, acosf / atan2f of the spherical mappings carry the usual last-ulp differences, amplified by the
    discontinuities of checker / dots edges only where a sample sits exactly on one."""
    from tests.util import PROCEDURAL_LOOK_AT, procedural_room
    sc = procedural_room(gpu.bvh_build)
    rd = scenes.make_render_desc(96, 64, 16, PROCEDURAL_LOOK_AT, 50, max_depth=3)
    film, li, st, ref = _render_pair(gpu, oracle, sc, rd)
    assert np.array_equal(film[:, 3], ref["film"][:, 3])
    assert film_rmse(film, ref["film"]) < 2e-4
    assert st["nan_samples"] == ref["counters"]["nan_samples"] == 0


def test_roughness_textures_match_oracle(gpu, oracle):
    """float textures on roughness / uroughness / vroughness (plastic.rs:86-92, substrate.rs:76-85, metal.rs, uber.rs): the
    per-hit alpha goes through roughness_to_alpha (one logf) and the 0.001 clamp of TrowbridgeReitzDistribution::new"""
    from tests.util import PROCEDURAL_LOOK_AT, roughness_room
    sc = roughness_room(gpu.bvh_build)
    rd = scenes.make_render_desc(96, 64, 16, PROCEDURAL_LOOK_AT, 50, max_depth=4)
    film, li, st, ref = _render_pair(gpu, oracle, sc, rd)
    assert np.array_equal(film[:, 3], ref["film"][:, 3])
    assert film_rmse(film, ref["film"]) < 2e-4
    assert st["nan_samples"] == ref["counters"]["nan_samples"] == 0


def test_textures_change_the_image_and_lens_differentials(gpu, oracle):
    """thin-lens camera differentials (perspective.rs:245-271) + Halton sampler on the textured room; and the
    textures must matter: the same room with constant colours renders a different film"""
    from tests.util import TEXTURED_LOOK_AT, textured_room
    sc = textured_room(gpu.bvh_build, bump=False, planar=False)
    rd = scenes.make_render_desc(64, 48, 8, TEXTURED_LOOK_AT, 45, max_depth=3, lens_radius=0.05, focal_distance=6.0, sampler="halton")
    film, li, st, ref = _render_pair(gpu, oracle, sc, rd)
    assert film_rmse(film, ref["film"]) < 2e-5
    sb = scenes.SceneBuilder()
    m = sb.add_material(scenes.matte((0.5, 0.5, 0.5)))
    sb.add_quad([(-5, 0, -5), (5, 0, -5), (5, 0, 5), (-5, 0, 5)], m)
    plain = sb.finish(gpu.bvh_build)
    assert (plain.textures["kind"] == abi.TEX_CONSTANT).all() and (sc.textures["kind"] != abi.TEX_CONSTANT).any()


def test_texture_validation(gpu):
    """bad texture tables are rejected on the host"""
    import ctypes as C
    from tests.util import textured_room
    L = gpu.lib()
    for breaker, code in (("image", abi.E_INVALID), ("child", abi.E_INVALID), ("slots", abi.E_INVALID), ("cycle", abi.E_INVALID), ("deep", abi.E_UNSUPPORTED),
                          ("kind", abi.E_UNSUPPORTED), ("mapping", abi.E_UNSUPPORTED)):
        sc = textured_room(gpu.bvh_build)
        k = int(np.nonzero(sc.textures["kind"] == abi.TEX_SCALE)[0][0])
        if breaker == "image":
            sc.textures["image"][0] = 99
        elif breaker == "child":
            sc.textures["tex1"][k] = 1000
        elif breaker == "cycle":
            sc.textures["tex1"][k] = k
        elif breaker == "deep":  # scale(scale(scale(scale(...)))) four levels: make three more scale nodes point at each other in a chain
            leaf = int(np.nonzero(sc.textures["kind"] == abi.TEX_CONSTANT)[0][0])
            chain = [i for i in range(len(sc.textures)) if i not in (k, leaf)][:3]
            for a, b in zip([k] + chain, chain + [leaf]):
                sc.textures["kind"][a] = abi.TEX_SCALE; sc.textures["tex1"][a] = b; sc.textures["tex2"][a] = leaf
        elif breaker == "kind":
            sc.textures["kind"][0] = 42
        elif breaker == "mapping":
            sc.textures["mapping"][0] = abi.MAP_IDENTITY3D
        else:
            sc.materials["kd"][0] = 7777   # a parameter that points past the texture array
        h = C.c_void_p()
        assert L.rspt_scene_create(C.addressof(sc.desc), C.addressof(h)) == code, breaker


def test_golden_textured_room(gpu):
    """committed oracle output for the textured room (tests/golden/make_golden.py)"""
    from tests.util import TEXTURED_LOOK_AT, textured_room
    g = np.load(os.path.join(GOLDEN, "textured_room_48x36x8.npz"))
    sc = textured_room(gpu.bvh_build)
    rd = scenes.make_render_desc(48, 36, 8, TEXTURED_LOOK_AT, 45, max_depth=3)
    ds = gpu.DeviceScene(sc)
    try:
        film, _ = gpu.render(ds, rd)
        li, _ = gpu.render_samples(ds, rd)
    finally:
        ds.close()
    assert np.array_equal(film[:, 3], g["film"][:, 3])
    assert film_rmse(film, g["film"]) < 1e-4
    assert np.array_equal(li, g["li"])   # the committed fixture (= today's oracle, tests/test_oracle_kat.py), every sample bit for bit


@pytest.mark.parametrize("cos_sample,sampler,n", [(True, "sobol", 16), (False, "sobol", 8), (True, "halton", 5)])
def test_ao_integrator_matches_oracle(gpu, oracle, cos_sample, sampler, n):
    """AOIntegrator::li through the shared render loop — closest hit, the pixel sample's slice of the
    sampler's 2-D array (dimensions 5, 6), n shadow rays, sum of the unoccluded terms in array order"""
    sc = scenes.cornell_box(gpu.bvh_build)
    rd = scenes.cornell_render_desc(res=48, spp=4, integrator="ao", ao_samples=n, ao_cos_sample=cos_sample, sampler=sampler)
    film, li, st, ref = _render_pair(gpu, oracle, sc, rd)
    assert np.array_equal(film[:, 3], ref["film"][:, 3])
    assert film_rmse(film, ref["film"]) < 1e-4
    assert 0.5 < li.mean() < 3.2 and st["nan_samples"] == 0


def test_ao_open_plane_is_pi_and_python_mirror(gpu):
    """cosine-sampled AO of an unoccluded point is exactly n * (cos / (cos / pi * n)) = pi; AOIntegrator mirror"""
    from rs_pbrt_amd.integrator import AOIntegrator
    sb = scenes.SceneBuilder()
    m = sb.add_material(scenes.matte((0.5, 0.5, 0.5)))
    sb.add_quad([(-50, 0, -50), (50, 0, -50), (50, 0, 50), (-50, 0, 50)], m)
    sc = sb.finish(gpu.bvh_build)
    cam = scenes.make_render_desc(32, 32, 4, ((0, 2, -3), (0, 0, 0), (0, 1, 0)), 40.0)
    film = AOIntegrator(camera=cam, n_samples=32).render(sc)
    assert np.allclose(film.rgb(), np.pi, atol=1e-4)
    import ctypes as C
    bad = scenes.make_render_desc(8, 8, 1, ((0, 2, -3), (0, 0, 0), (0, 1, 0)), 40.0, integrator="ao", ao_samples=0)
    ds = gpu.DeviceScene(sc)
    try:
        out = np.zeros((64, 4), np.float32)
        assert gpu.lib().rspt_render(ds.handle, C.addressof(bad), out.ctypes.data, None) == abi.E_INVALID
    finally:
        ds.close()


@pytest.mark.parametrize("seed", list(range(101, 125)))
def test_random_scenes_fuzz(gpu, oracle, seed):
    """random rooms with every material recipe / texture binding / light kind drawn at random, both samplers and all
    light strategies: weights exact, film within the bump-map bar (textures + bump maps appear in most of them)"""
    from tests.util import GALLERY_LOOK_AT, random_scene
    sc = random_scene(gpu.bvh_build, seed)
    rd = scenes.make_render_desc(56, 40, 8, GALLERY_LOOK_AT, 55, max_depth=2 + seed % 5, sampler="halton" if seed % 2 else "sobol",
                                 light_strategy=[abi.LIGHTS_SPATIAL, abi.LIGHTS_POWER, abi.LIGHTS_UNIFORM][seed % 3],
                                 lens_radius=0.03 if seed % 4 == 0 else 0.0, focal_distance=6.0)
    film, li, st, ref = _render_pair(gpu, oracle, sc, rd)
    assert np.array_equal(film[:, 3], ref["film"][:, 3])
    assert st["nan_samples"] == ref["counters"]["nan_samples"]
    # (a roughness texture puts one logf in front of every microfacet term of that material: an ulp of alpha moves
    # most of its samples by an ulp or two, so the bit-identical share drops while the differences stay tiny)
    assert film_rmse(film, ref["film"]) < 3e-4


@pytest.mark.parametrize("kw", [dict(), dict(sampler="halton"), dict(integrator="volpath"), dict(integrator="ao", ao_samples=4), dict(integrator="directlighting")])
def test_sample_ranges_add_up_to_the_frame(gpu, oracle, kw):
    """checkpoint / resume (rspt_render_desc.sample_begin / sample_count): films of disjoint sample ranges sum to the full
    film — weights exactly, radiance up to the order of the additions — and each partial film is the oracle's for that range"""
    sc = scenes.cornell_box(gpu.bvh_build, variant="matte" if kw.get("integrator") == "directlighting" else "mixed", fog=scenes.CORNELL_FOG if kw.get("integrator") == "volpath" else None)
    full_rd = scenes.cornell_render_desc(res=48, spp=12 if kw.get("sampler") == "halton" else 16, **kw)
    spp = int(full_rd.spp)
    with gpu.DeviceScene(sc) as ds:
        full, st = gpu.render(ds, full_rd)
        total = np.zeros_like(full)
        n = 0
        for begin, count in ((0, 5), (5, 1), (6, spp - 6)):
            rd = scenes.cornell_render_desc(res=48, spp=spp, sample_range=(begin, count), **kw)
            part, pst = gpu.render(ds, rd)
            ref = oracle.render(sc, rd, threads=8) if kw.get("integrator") != "directlighting" else oracle.render_integrator(sc, rd, "direct", threads=8)
            assert np.array_equal(part[:, 3], ref["film"][:, 3]) and film_rmse(part, ref["film"]) < (1e-5 if count > 1 else 1e-4)
            total += part
            n += pst["samples"]
        from rs_pbrt_amd.lib import RsptError
        with pytest.raises(RsptError):
            gpu.render(ds, scenes.cornell_render_desc(res=48, spp=spp, sample_range=(spp - 2, 3), **kw))
    assert n == st["samples"] and np.array_equal(total[:, 3], full[:, 3])
    assert np.allclose(total[:, :3], full[:, :3], rtol=2e-6, atol=1e-6)


@pytest.mark.parametrize("case", ["cornell-mixed", "cornell-rough-halton", "gallery", "sky", "textured-bump", "procedural", "fog-volpath", "random-105", "random-118", "ao",
                                  "cornell-02sequence"])
def test_radiance_of_every_sample_is_bit_identical_to_the_oracle(gpu, oracle, case):
    """With sinf / cosf / logf / log2f / expf / acosf / atan2f evaluated as the host libm evaluates them (glibc_libm.h) nothing on the path
    rounds differently from the reference any more: every camera sample's radiance equals the oracle's bit for bit — BSDF sampling, microfacet
    lobes, image / procedural textures with EWA and bump maps, infinite-light importance sampling, media, all light strategies.  (The film
    still differs in the last bits where several samples splat into one pixel from different tiles: atomics add in arrival order.)"""
    from tests.util import (GALLERY_LOOK_AT, PROCEDURAL_LOOK_AT, SKY_LOOK_AT, TEXTURED_LOOK_AT, gallery, procedural_room, random_scene, sky_scene, textured_room)
    b = gpu.bvh_build
    if case == "cornell-mixed": sc, rd = scenes.cornell_box(b, variant="mixed"), scenes.cornell_render_desc(res=64, spp=16)
    elif case == "cornell-rough-halton": sc, rd = scenes.cornell_box(b, variant="rough"), scenes.cornell_render_desc(res=64, spp=12, sampler="halton")
    elif case == "gallery": sc, rd = gallery(b), scenes.make_render_desc(64, 48, 16, GALLERY_LOOK_AT, 60.0, max_depth=5)
    elif case == "sky": sc, rd = sky_scene(b, kind="image", with_area=True), scenes.make_render_desc(64, 48, 16, SKY_LOOK_AT, 50.0)
    elif case == "textured-bump": sc, rd = textured_room(b, bump=True), scenes.make_render_desc(96, 72, 16, TEXTURED_LOOK_AT, 45, max_depth=4)
    elif case == "procedural": sc, rd = procedural_room(b), scenes.make_render_desc(96, 64, 16, PROCEDURAL_LOOK_AT, 50, max_depth=3)
    elif case == "fog-volpath":
        from tests.test_gpu_volpath import LOOK, fog_room
        sc, rd = fog_room(b, glass=True), scenes.make_render_desc(64, 48, 16, LOOK, 55.0, integrator="volpath")
    elif case.startswith("random-"):
        seed = int(case.split("-")[1])
        sc, rd = random_scene(b, seed), scenes.make_render_desc(56, 40, 8, GALLERY_LOOK_AT, 55, max_depth=5, light_strategy=[abi.LIGHTS_SPATIAL, abi.LIGHTS_POWER, abi.LIGHTS_UNIFORM][seed % 3])
    elif case == "ao": sc, rd = scenes.cornell_box(b), scenes.cornell_render_desc(res=64, spp=8, integrator="ao", ao_samples=16)
    else: sc, rd = scenes.cornell_box(b, variant="mixed"), scenes.cornell_render_desc(res=80, spp=16, sampler="02sequence")
    film, li, st, ref = _render_pair(gpu, oracle, sc, rd)
    assert np.array_equal(li, ref["li"]) and li.max() > 0
    assert np.array_equal(film[:, 3], ref["film"][:, 3]) and film_rmse(film, ref["film"]) < 1e-7


def test_checkpoint_resume(gpu, tmp_path):
    """integrator.Checkpoint: a render interrupted after 5 of 16 samples, saved, and finished by a fresh object equals the one-shot
    frame; a checkpoint of another frame is refused"""
    from rs_pbrt_amd import integrator
    sc = scenes.cornell_box(gpu.bvh_build, variant="rough")
    pi = integrator.PathIntegrator(camera=scenes.cornell_render_desc(res=48, spp=16))
    with gpu.DeviceScene(sc) as ds:
        whole = pi.render(ds).pixels.reshape(-1, 4)
        a = integrator.Checkpoint(pi)
        assert a.step(ds, 5) == 5 and not a.done
        a.save(tmp_path / "ckpt.npz")
        b = integrator.Checkpoint(integrator.PathIntegrator(camera=scenes.cornell_render_desc(res=48, spp=16)))
        b.load(tmp_path / "ckpt.npz")
        assert b.next_sample == 5
        while not b.done:
            b.step(ds, 4)
        got = b.film().pixels.reshape(-1, 4)
        other = integrator.Checkpoint(integrator.PathIntegrator(camera=scenes.cornell_render_desc(res=48, spp=32)))
        with pytest.raises(ValueError):
            other.load(tmp_path / "ckpt.npz")
    assert np.array_equal(got[:, 3], whole[:, 3]) and np.allclose(got[:, :3], whole[:, :3], rtol=2e-6, atol=1e-6)


@pytest.mark.parametrize("strategy", [abi.LIGHTS_SPATIAL, abi.LIGHTS_POWER, abi.LIGHTS_UNIFORM])
def test_many_area_lights(gpu, oracle, strategy):
    """the C4 axis "many lights" (SURVEY 8d): 98 emissive triangles of different power; light selection through the
    spatial voxel tables / power / uniform distributions must pick the same light for the same sample"""
    sb = scenes.SceneBuilder()
    white = sb.add_material(scenes.matte((0.7, 0.7, 0.7)))
    shiny = sb.add_material(scenes.plastic((0.3, 0.2, 0.5), (0.4, 0.4, 0.4), 0.1))
    q = sb.add_quad
    q([(-5, 0, -5), (-5, 0, 5), (5, 0, 5), (5, 0, -5)], white)
    q([(-5, 0, 5), (-5, 6, 5), (5, 6, 5), (5, 0, 5)], white)
    q([(-2, 0.2, 1), (0, 0.2, 1), (0, 2.5, 2), (-2, 2.5, 2)], shiny)
    q([(1, 0.2, 0.5), (3, 0.2, 1.5), (3, 2.0, 1.5), (1, 2.0, 0.5)], white)
    rng = np.random.default_rng(9)
    for a in range(7):
        for b in range(7):
            cx, cz, h = -4.2 + 1.4 * a, -4.2 + 1.4 * b, 0.15
            q([(cx + h, 5.5, cz - h), (cx + h, 5.5, cz + h), (cx - h, 5.5, cz + h), (cx - h, 5.5, cz - h)], white, emit=tuple(rng.uniform(5, 60, 3)))
    sc = sb.finish(gpu.bvh_build)
    assert len(sc.lights) == 98
    from tests.util import GALLERY_LOOK_AT
    rd = scenes.make_render_desc(72, 54, 16, GALLERY_LOOK_AT, 55, max_depth=3, light_strategy=strategy)
    film, li, st, ref = _render_pair(gpu, oracle, sc, rd)
    assert np.array_equal(film[:, 3], ref["film"][:, 3])
    assert film_rmse(film, ref["film"]) < 2e-5


@pytest.mark.parametrize("sampler", ["sobol", "halton"])
def test_materials_whose_lobe_list_depends_on_textures(gpu, oracle, sampler):
    """Dynamic materials (material_assembly.h): sigma / index / opacity / Kr / Kt / reflect / transmit / eta / k / a glass roughness / a mix
    amount bound to image and checker textures — the texture stage leaves the raw values in the path's rows, the shade stage runs the
    host's own assembly function per hit.  The oracle evaluates compute_scattering_functions at every hit as the reference does; every
    camera sample's radiance must be equal bit for bit."""
    from tests.util import DYNAMIC_LOOK_AT, dynamic_gallery
    sc = dynamic_gallery(gpu.bvh_build)
    from rs_pbrt_amd import lib as _lib
    assert sum(_lib.material_lobes(sc, i)[2] is None for i in range(len(sc.materials))) >= 8   # the slabs' materials are dynamic
    rd = scenes.make_render_desc(96, 40, 16, DYNAMIC_LOOK_AT, 75.0, max_depth=6, sampler=sampler)
    film, li, st, ref = _render_pair(gpu, oracle, sc, rd)
    assert st["nan_samples"] == 0 and np.array_equal(film[:, 3], ref["film"][:, 3])
    assert film_rmse(film, ref["film"]) < 1e-6
    rgb = scenes.film_to_rgb(film).reshape(40, 96, 3)
    assert rgb[10:30].std() > 0.02   # the slabs are in view and differ


@pytest.mark.parametrize("kw", [dict(integrator="volpath"), dict(integrator="volpath", sampler="stratified"), dict(integrator="directlighting"),
                                dict(integrator="directlighting", sampler="halton", direct_strategy="one"), dict(integrator="directlighting", sampler="random"),
                                dict(sampler="02sequence"), dict(sampler="maxmindist")])
def test_dynamic_materials_under_every_integrator_and_sampler(gpu, oracle, kw):
    """the same gallery through the forms that used to refuse it: VolPathIntegrator (wavefront k_vol_shade<DYN> and per lane),
    DirectLightingIntegrator (allow_multiple_lobes = false: the lists are assembled for that; per lane with one lobe record per
    recursion level), PathIntegrator under the pixel samplers (k_tile_serial mode 4)"""
    from tests.util import DYNAMIC_LOOK_AT, dynamic_gallery
    sc = dynamic_gallery(gpu.bvh_build)
    ls = [1] * sc.desc.n_lights
    rd = scenes.make_render_desc(64, 28, 16 if kw.get("sampler") in ("stratified", "maxmindist") else 8, DYNAMIC_LOOK_AT, 75.0, max_depth=4, light_samples=ls, **kw)
    if kw.get("integrator") == "directlighting":
        from tests.test_gpu_directlighting import check
        check(gpu, oracle, sc, rd, kw.get("direct_strategy", "all"), ls if kw.get("direct_strategy", "all") == "all" else None)
    else:
        film, li, st, ref = _render_pair(gpu, oracle, sc, rd)
        assert st["nan_samples"] == 0 and np.array_equal(film[:, 3], ref["film"][:, 3]) and film_rmse(film, ref["film"]) < 1e-6


@pytest.mark.parametrize("kw", [dict(), dict(sampler="halton"), dict(integrator="volpath"), dict(integrator="directlighting"), dict(sampler="02sequence")])
def test_mixes_of_mixes(gpu, oracle, kw):
    """MixMaterial inside MixMaterial (mixmat.rs:43-76): the library flattens the tree into its non-mix materials, each under its immediate
    parent's scale only (a mix ignores the `_scale` it is handed, :50); the oracle recurses through compute_scattering_functions as the
    reference does.  Static trees (constant amounts) and dynamic ones (image / checker amounts, also behind an m2 edge), every integrator
    form: per-sample radiance bit for bit."""
    from tests.util import DYNAMIC_LOOK_AT, nested_mix_gallery
    sc = nested_mix_gallery(gpu.bvh_build)
    from rs_pbrt_amd import lib as _lib
    dyn = [_lib.material_lobes(sc, i)[2] is None for i in range(len(sc.materials))]
    assert sum(dyn) >= 2 and not all(dyn)
    ls = [1] * sc.desc.n_lights
    rd = scenes.make_render_desc(64, 28, 8, DYNAMIC_LOOK_AT, 75.0, max_depth=5, light_samples=ls, **kw)
    if kw.get("integrator") == "directlighting":
        from tests.test_gpu_directlighting import check
        check(gpu, oracle, sc, rd, "all", ls)
    else:
        film, li, st, ref = _render_pair(gpu, oracle, sc, rd)
        assert st["nan_samples"] == 0 and np.array_equal(film[:, 3], ref["film"][:, 3]) and film_rmse(film, ref["film"]) < 1e-6
        assert scenes.film_to_rgb(film).reshape(28, 64, 3)[8:22].std() > 0.02   # the slabs are in view and differ


def test_film_reduce_runs_inside_the_library(gpu):
    """X1 in the product: rspt_comm_unique_id / rspt_comm_init create the RCCL communicator, film_reduce = 1 makes
    rspt_render end with ncclReduce(sum) onto rank 0.  One GPU here, so the world has one rank (the sum of one film is that film,
    bit for bit); what is checked is that the collective is issued by librspt on its own stream, and the argument checks."""
    from rs_pbrt_amd.lib import RsptError
    sc = scenes.cornell_box(gpu.bvh_build)
    rd = scenes.cornell_render_desc(res=48, spp=4)
    with gpu.DeviceScene(sc) as ds:
        plain, _ = gpu.render(ds, rd)
        rd.film_reduce = 1
        with pytest.raises(RsptError):  # no communicator yet
            gpu.render(ds, rd)
        gpu.comm_init(0, 1, gpu.comm_unique_id())
        try:
            reduced, st = gpu.render(ds, rd)
            assert np.array_equal(plain, reduced) and st["samples"] == 48 * 48 * 4
            rd.shard_index, rd.shard_count, rd.tile_chunk = 0, 2, 1
            with pytest.raises(RsptError) as e:  # shard_count must equal the communicator's world size
                gpu.render(ds, rd)
            assert e.value.code == abi.E_INVALID and "shard" in str(e.value)   # ... and the failed call took part in the status agreement
            # (a rank that fails before the reduce must not leave the others waiting in it) keeping its own error; the next frame runs
            rd.shard_index, rd.shard_count = 0, 1
            again, _ = gpu.render(ds, rd)
            assert np.array_equal(again, plain)
            bad = scenes.cornell_render_desc(res=48, spp=4, integrator="directlighting", max_depth=33)   # (past the recursion stack of the per-lane form)
            bad.film_reduce = 1
            with pytest.raises(RsptError) as e:
                gpu.render(ds, bad)
            assert e.value.code == abi.E_UNSUPPORTED
            assert np.array_equal(gpu.render(ds, rd)[0], plain)
        finally:
            gpu.comm_destroy()


def test_film_reduce_with_two_ranks(gpu, tmp_path):
    """X1 with a world of two: two processes (one device each where the box has two devices, both on device 0 otherwise) join the library's RCCL communicator, render the shards
    (0, 2, 1) / (1, 2, 1) with film_reduce = 1; rank 0's buffer must hold the whole frame (= the single-rank render: weights bit for
    bit, radiance up to the order of the sum).  RCCL refuses two ranks on one device ("Duplicate GPU detected"): on a one-GPU box the test checks that both ranks come back
    with an error code and a message (no hang) and then reports XFAIL — the reduce itself did not run there, and the report must say so."""
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_rccl_worker.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=240)[0].decode("utf-8", "replace"))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("two-rank reduce hung")
    errs = [open(os.path.join(tmp_path, f)).read() for f in os.listdir(tmp_path) if f.startswith("init_error")]
    if errs:
        # a one-device box: RCCL takes one rank per device.  What is checked here is then the error path of rspt_comm_init — BOTH ranks come back with an
        # error code and a message (no hang, no crash); the two-rank control flow itself runs over gloo in test_bench_two_ranks_share_one_device below
        import torch
        assert torch.cuda.device_count() < 2, "RCCL refused two ranks although the box has %d devices: %s" % (torch.cuda.device_count(), errs[0])
        assert len(errs) == 2 and [p.returncode for p in procs] == [2, 2], (errs, outs)
        assert all("librspt error" in e for e in errs), errs
        pytest.xfail("one device: RCCL refuses two ranks on it (%s) — ncclReduce with a world of two did NOT run; only rspt_comm_init's error path was checked" % errs[0].strip().splitlines()[-1][:120])
    assert [p.returncode for p in procs] == [0, 0], outs
    sc = scenes.cornell_box(gpu.bvh_build)
    with gpu.DeviceScene(sc) as ds:
        whole, st = gpu.render(ds, scenes.cornell_render_desc(res=80, spp=4))
    reduced = np.load(os.path.join(tmp_path, "film_0.npy"))
    part1 = np.load(os.path.join(tmp_path, "film_1.npy"))
    n = int(np.load(os.path.join(tmp_path, "samples_0.npy"))[0]) + int(np.load(os.path.join(tmp_path, "samples_1.npy"))[0])
    assert n == st["samples"] == 80 * 80 * 4
    assert np.array_equal(reduced[:, 3], whole[:, 3]) and np.allclose(reduced, whole, rtol=1e-6, atol=1e-7)
    assert 0 < part1[:, 3].sum() < whole[:, 3].sum()   # the other rank keeps its partial film
    codes = [int(np.load(os.path.join(tmp_path, "samples_%d.npy" % r))[1]) for r in range(2)]
    assert codes == [abi.E_PEER, abi.E_UNSUPPORTED], codes   # the failed rank keeps its own error, the healthy one is told
    assert np.array_equal(np.load(os.path.join(tmp_path, "film3_0.npy")), reduced)


def test_round5_schedule_switches_leave_the_film_bit_identical(gpu, monkeypatch):
    """RSPT_FRESH=0 (k_raygen writes L / beta, the first shade launch reads them back), RSPT_XCD_DEAL=1 and RSPT_W4_SHAPE=1 change which bytes move and which wave
    traces which ray, never a sample: the film of a gallery frame (every material recipe, four light kinds) is bit-identical under each"""
    sc = gallery(gpu.bvh_build)
    rd = scenes.make_render_desc(96, 72, 16, GALLERY_LOOK_AT, 60.0, max_depth=5)
    with gpu.DeviceScene(sc) as ds:
        base, _ = gpu.render(ds, rd)
        for env in (dict(RSPT_FRESH="0"), dict(RSPT_XCD_DEAL="1"), dict(RSPT_W4_SHAPE="1")):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            film, _ = gpu.render(ds, rd)
            for k in env:
                monkeypatch.delenv(k)
            assert np.array_equal(film, base), env


def test_bench_two_ranks_share_one_device(tmp_path):
    """`python bench.py --gpus 2` end to end on whatever the box has — self_spawn (torch.distributed.run, two ranks), the unique-id broadcast,
    the Morton tile deal (rank, 2, 1), the film sum onto rank 0 (the library's ncclReduce where there are two devices; gloo through host memory where the
    ranks share one), max-over-ranks time, summed samples, ONE JSON line — and the summed frame against the one-rank frame: filter weights bit for bit,
    radiance to 1e-6 (the order of the sum at shared pixels).  The collector it replaces: integrator.rs:209-215."""
    import json
    import subprocess
    import sys
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    common = ["--workload", "cornell", "--res", "96", "--spp", "4", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-extra", "--no-count", "--watchdog", "400"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    lines = {}
    for n in (1, 2):
        r = subprocess.run([sys.executable, bench, "--gpus", str(n), "--dump-film", str(tmp_path / ("film%d.npy" % n))] + common, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        js = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(js) == 1, r.stdout
        lines[n] = json.loads(js[0])
    one, two = lines[1], lines[2]
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["config"]["samples_per_step"] == one["config"]["samples_per_step"] == 96 * 96 * 4
    assert two["scaling"] == "strong" and two["value"] > 0 and two["config"]["librccl"] and "librccl" in two["config"]["librccl"]
    f1, f2 = np.load(tmp_path / "film1.npy"), np.load(tmp_path / "film2.npy")
    assert f1.shape == f2.shape == (96 * 96, 4)
    assert np.array_equal(f1[:, 3], f2[:, 3]) and np.allclose(f1, f2, rtol=1e-6, atol=1e-7)
    assert f1[:, 3].min() > 0


def test_spatial_light_distribution_on_demand_voxels(gpu, oracle):
    """the spatial distribution is built voxel by voxel as paths look voxels up (what the reference's lazily filled hash
    table does, lightdistrib.rs:297-384) once the full table would be large; a voxel's distribution is a pure function of
    (voxel, lights) (Q18), so the film must equal the eager build's bit for bit.  A pool that is too small fails loudly."""
    import os
    from rs_pbrt_amd.lib import RsptError
    sc = gallery(gpu.bvh_build)
    rd = scenes.make_render_desc(64, 48, 8, GALLERY_LOOK_AT, 60.0, max_depth=5)
    with gpu.DeviceScene(sc) as ds:
        eager, _ = gpu.render(ds, rd)
    try:
        os.environ["RSPT_LIGHT_TABLE_EAGER_BYTES"] = "0"
        with gpu.DeviceScene(sc) as ds:
            lazy, st = gpu.render(ds, rd)
            again, _ = gpu.render(ds, rd)    # second render: every voxel it needs is already there
        assert np.array_equal(eager, lazy) and np.array_equal(lazy, again)
        ref = oracle.render(sc, rd, threads=8)
        assert np.array_equal(lazy[:, 3], ref["film"][:, 3]) and film_rmse(lazy, ref["film"]) < 1e-5
        os.environ["RSPT_LIGHT_TABLE_POOL_BYTES"] = "64"   # room for a single row
        with gpu.DeviceScene(sc) as ds:
            with pytest.raises(RsptError) as e:
                gpu.render(ds, rd)
            assert e.value.code == abi.E_NOMEM
    finally:
        os.environ.pop("RSPT_LIGHT_TABLE_EAGER_BYTES", None); os.environ.pop("RSPT_LIGHT_TABLE_POOL_BYTES", None)


@pytest.mark.parametrize("kw", [dict(integrator="volpath"), dict(integrator="volpath", sampler="halton"), dict(sampler="02sequence"), dict(sampler="random"),
                                dict(integrator="volpath", sampler="stratified")])
def test_on_demand_light_voxels_under_volpath_and_the_pixel_samplers(gpu, kw):
    """VERDICT r3 missing #8: the kernels that meet their lookup points only while they run — volpath (a point in a medium is sampled inside
    the kernel) and the pixel samplers (a tile is one serial chain) — claim missing voxels themselves and the step runs again once the rows
    are built (lightdistrib.rs:276-384 builds on first touch too).  A row is a pure function of (voxel, lights), so the film must equal the
    eager build's bit for bit; the pool limit fails loudly here too."""
    import os
    from rs_pbrt_amd.lib import RsptError
    fog = kw.get("integrator") == "volpath"
    sc = scenes.cornell_box(gpu.bvh_build, fog=scenes.CORNELL_FOG) if fog else gallery(gpu.bvh_build)
    spp = 16 if kw.get("sampler") == "stratified" else 8
    rd = scenes.cornell_render_desc(res=48, spp=spp, **kw) if fog else scenes.make_render_desc(64, 48, spp, GALLERY_LOOK_AT, 60.0, max_depth=5, **kw)
    rd.allow_slow_paths = 1
    with gpu.DeviceScene(sc) as ds:
        eager, _ = gpu.render(ds, rd)
    try:
        os.environ["RSPT_LIGHT_TABLE_EAGER_BYTES"] = "0"
        with gpu.DeviceScene(sc) as ds:
            lazy, st = gpu.render(ds, rd)
            again, _ = gpu.render(ds, rd)
        assert np.array_equal(eager, lazy) and np.array_equal(lazy, again) and eager[:, :3].sum() > 0
        os.environ["RSPT_LIGHT_TABLE_POOL_BYTES"] = "64"   # room for a single row
        with gpu.DeviceScene(sc) as ds:
            with pytest.raises(RsptError) as e:
                gpu.render(ds, rd)
            assert e.value.code == abi.E_NOMEM
    finally:
        os.environ.pop("RSPT_LIGHT_TABLE_EAGER_BYTES", None); os.environ.pop("RSPT_LIGHT_TABLE_POOL_BYTES", None)


def test_light_distribution_hook_equals_oracle_voxel_by_voxel(gpu, oracle):
    """rspt_light_distribution = LightDistribution::lookup(p) (lightdistrib.rs:33-39): func / cdf of p's voxel, bit for bit the
    oracle's spatial_compute (128 Halton points per voxel x every light, :297-384), from the eager table and from on-demand rows;
    uniform and power are one row for every p"""
    import ctypes as C, os
    sc = gallery(gpu.bvh_build)
    rd = scenes.make_render_desc(64, 48, 1, GALLERY_LOOK_AT, 60.0)
    nl = int(sc.desc.n_lights)
    lo, hi = sc.nodes["bmin"][0], sc.nodes["bmax"][0]   # BVHAccel::world_bound (bvh.rs:394-400)
    pts = np.random.default_rng(4).uniform(lo - 0.5, hi + 0.5, (60, 3)).astype(np.float32)   # some outside: clamped to the border voxels

    def check(ds):
        for p in pts:
            f, c, nv, vx = gpu.light_distribution(ds, abi.LIGHTS_SPATIAL, p)
            of, oc, onv = np.zeros(nl, np.float32), np.zeros(nl + 1, np.float32), (C.c_int32 * 3)()
            oracle.lib().orc_spatial_voxel(C.addressof(sc.desc), C.addressof(rd), (C.c_int32 * 3)(*map(int, vx)), of.ctypes.data, oc.ctypes.data, onv)
            assert list(nv) == list(onv) and np.array_equal(f, of) and np.array_equal(c, oc)
    with gpu.DeviceScene(sc) as ds:
        check(ds)
        f, c, nv, _ = gpu.light_distribution(ds, abi.LIGHTS_UNIFORM, pts[0])
        assert list(nv) == [1, 1, 1] and np.all(f == 1.0) and np.allclose(c, np.arange(nl + 1) / nl, atol=1e-6)
        f, c, _, _ = gpu.light_distribution(ds, abi.LIGHTS_POWER, pts[0])
        assert f.min() > 0 and c[0] == 0 and abs(c[-1] - 1) < 1e-6 and np.all(np.diff(c) > 0)
    try:
        os.environ["RSPT_LIGHT_TABLE_EAGER_BYTES"] = "0"
        with gpu.DeviceScene(sc) as ds:
            check(ds)   # every voxel is built by the hook itself
            check(ds)   # and found the second time
    finally:
        os.environ.pop("RSPT_LIGHT_TABLE_EAGER_BYTES", None)


def test_scene_with_ten_thousand_emissive_triangles_renders(gpu, oracle):
    """the case the eager table refused in round 1 (64^3 voxels x 10 082 lights): an emissive 71 x 71 grid (10 082 light triangles)
    over a room; only the voxels paths actually reach are built"""
    sb = scenes.SceneBuilder()
    grey = sb.add_material(scenes.matte((0.6, 0.6, 0.6)))
    sb.add_quad([(-5, 0, -5), (-5, 0, 5), (5, 0, 5), (5, 0, -5)], grey)
    sb.add_quad([(-5, 0, 5), (-5, 6, 5), (5, 6, 5), (5, 0, 5)], grey)
    sb.add_quad([(-1, 0.0, 1), (1, 0.0, 1), (1, 2.0, 2), (-1, 2.0, 2)], sb.add_material(scenes.plastic((0.5, 0.3, 0.2), (0.3, 0.3, 0.3), 0.1)))
    k = 71
    g = np.linspace(-2, 2, k + 1, dtype=np.float32)
    X, Z = np.meshgrid(g, g, indexing="xy")
    P = np.stack([X, np.full_like(X, 5.9), Z], -1).reshape(-1, 3)
    i, j = np.meshgrid(np.arange(k), np.arange(k), indexing="ij")
    a = (i * (k + 1) + j).reshape(-1); b = a + 1; c = a + (k + 1); e = c + 1
    sb.add_mesh(P, np.concatenate([np.stack([a, b, c], 1), np.stack([b, e, c], 1)]), grey, emit=(6, 6, 6))
    sc = sb.finish(gpu.bvh_build)
    assert sc.desc.n_lights == 2 * k * k
    rd = scenes.make_render_desc(32, 24, 4, GALLERY_LOOK_AT, 60.0, max_depth=3)
    with gpu.DeviceScene(sc) as ds:
        film, st = gpu.render(ds, rd)
    assert st["nan_samples"] == 0 and np.isfinite(film).all() and film[:, 1].mean() > 0.05
    ref = oracle.render(sc, rd, threads=THREADS_ALL)
    assert np.array_equal(film[:, 3], ref["film"][:, 3]) and film_rmse(film, ref["film"]) < 1e-4


@pytest.mark.parametrize("sampler,integrator", [("sobol", "path"), ("halton", "path"), ("02sequence", "path"), ("sobol", "directlighting"), ("stratified", "volpath"), ("sobol", "ao")])
def test_moving_camera(gpu, oracle, sampler, integrator):
    """SURVEY a4 / VERDICT r2 missing #7: CameraBase.camera_to_world as an AnimatedTransform (transform.rs:894-2124).  The camera turns and
    travels during the exposure: every camera ray — and the differentials the textures are filtered with — goes through the matrix
    interpolated at its own time sample (translation and scale linearly, rotation by slerp); times before / after the key times use the
    key matrices.  Thin lens on, so that the lens and the time values of a sample are both in play.  Per-sample radiance bit-identical."""
    from tests.util import TEXTURED_LOOK_AT, textured_room
    sc = textured_room(gpu.bvh_build, specular=(integrator == "directlighting"))
    la1 = ((1.2, 3.1, -4.2), (0.4, 1.3, 2.0), (0.2, 1.0, 0.05))
    rd = scenes.make_render_desc(48, 36, 8 if sampler != "stratified" else 16, TEXTURED_LOOK_AT, 55.0, max_depth=3, sampler=sampler, integrator=integrator, ao_samples=4,
                                 light_samples=[1] * sc.desc.n_lights, look_at_end=la1, camera_times=(0.2, 0.85), shutter=(0.0, 1.0), lens_radius=0.03, focal_distance=6.0)
    if integrator == "directlighting":   # (the oracle's DirectLightingIntegrator has its own entry point)
        from tests.test_gpu_directlighting import check
        film = check(gpu, oracle, sc, rd, "all", [1] * sc.desc.n_lights)
    else:
        film, li, st, ref = _render_pair(gpu, oracle, sc, rd)
        assert np.array_equal(film[:, 3], ref["film"][:, 3]) and film_rmse(film, ref["film"]) < 1e-6
    rd_static = scenes.make_render_desc(48, 36, 8 if sampler != "stratified" else 16, TEXTURED_LOOK_AT, 55.0, max_depth=3, sampler=sampler, integrator=integrator, ao_samples=4,
                                        light_samples=[1] * sc.desc.n_lights, lens_radius=0.03, focal_distance=6.0)
    with gpu.DeviceScene(sc) as ds:
        still = gpu.render(ds, rd_static)[0]
        rd_static.camera_animated = 1          # equal key matrices: the camera does not move (actually_animated = false)
        rd_static.camera_to_world_end[:] = list(rd_static.camera_to_world)
        assert np.array_equal(gpu.render(ds, rd_static)[0], still)
    assert film_rmse(film, still) > 1e-3       # the motion is in the picture
