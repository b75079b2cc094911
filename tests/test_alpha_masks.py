"""TriangleMesh.alpha_mask / shadow_alpha_mask (the "alpha" / "shadowalpha" shape parameters, api.rs:1920-1965): a candidate hit where
the float texture evaluates to exactly 0 is no hit — Triangle::intersect tests alpha_mask (triangle.rs:313-330), Triangle::intersect_p
both masks (:593-655).  CPU: the oracle's restatement against geometry with the holes cut out for real.  -m gpu: librspt's trace
kernels (plain and instance-aware variants, reference-order loop) and renders against the oracle."""
import numpy as np
import pytest

from rs_pbrt_amd import abi, scenes
from tests.util import film_rmse, random_rays

LOOK = ((0, 2.0, -5.0), (0, 1.0, 0), (0, 1, 0))


def masked_scene(builder, cut=False, shadow_only=False, instanced=False, mode="fixed", plain=False, moving=False):
    """a wall, a floor, a light, and a 2 x 2 m panel in front of the wall whose 4 x 4 checker alpha texture (constants 0 / 1) removes every
    other cell.  (No surface lies on a boundary of the 64-voxel light-distribution grid: chosen when the device library's sinf / cosf still moved bounce-ray
    hit points by an ulp, which on a boundary picks the neighbouring voxel's distribution — DESIGN.md §3; no longer needed.)  cut=True: the same panel modelled as the eight remaining cells (no texture) — the first-principles twin.
    shadow_only: the mask sits in shadow_alpha_mask (camera rays see the whole panel, shadow rays go through the holes)."""
    sb = scenes.SceneBuilder()
    grey = sb.add_material(scenes.matte((0.6, 0.6, 0.6)))
    red = sb.add_material(scenes.matte((0.7, 0.2, 0.15)))
    sb.add_quad([(-4, 0, -4), (-4, 0, 4), (4, 0, 4), (4, 0, -4)], grey)
    sb.add_quad([(-4, 0, 2.97), (-4, 5, 2.97), (4, 5, 2.97), (4, 0, 2.97)], grey)
    sb.add_quad([(-1, 4.47, -1), (1, 4.47, -1), (1, 4.47, 1), (-1, 4.47, 1)], grey, emit=(12, 12, 12))
    sb.add_point_light((0, 2, -4), (25, 25, 25))
    x0, y0, z = -1.0, 0.2, 0.97
    if cut:
        for i in range(4):
            for j in range(4):
                if (i + j) % 2 == 1:   # checkerboard.rs:32-42: (floor(s) + floor(t)) % 2 == 0 -> tex1 (alpha 0 here)
                    a, b = x0 + 0.5 * i, y0 + 0.5 * j
                    sb.add_quad([(a, b, z), (a + 0.5, b, z), (a + 0.5, b + 0.5, z), (a, b + 0.5, z)], red,
                                UV=[[i / 4, j / 4], [(i + 1) / 4, j / 4], [(i + 1) / 4, (j + 1) / 4], [i / 4, (j + 1) / 4]])
        return sb.finish(builder)
    mask = sb.checkerboard_texture(sb.constant_texture(0.0), sb.constant_texture(1.0), su=4.0, sv=4.0)
    kw = {} if plain else (dict(shadow_alpha=mask) if shadow_only else dict(alpha=mask))
    panel = [(x0, y0, z), (x0 + 2, y0, z), (x0 + 2, y0 + 2, z), (x0, y0 + 2, z)]
    uv = [[0, 0], [1, 0], [1, 1], [0, 1]]
    if instanced:
        sb.begin_object("panel")
        sb.add_mesh(np.array(panel, np.float32) - np.array([0, 0, z], np.float32), [[0, 1, 2], [0, 2, 3]], red, UV=uv, **kw)
        sb.add_mesh(np.array([(-0.2, 0, 0.3), (0.2, 0, 0.3), (0, 0.4, 0.3)], np.float32), [[0, 1, 2]], red)
        sb.end_object()
        if moving:   # (round 5) the masked panels MOVE over the shutter: one turns (slerp), one slides and grows
            T = scenes.Transform
            sb.add_instance("panel", T.translate((0, 0, z)) * T.rotate_y(8.0), T.translate((0.3, 0.1, z)) * T.rotate_y(40.0))
            sb.add_instance("panel", T.translate((2.6, 0.3, 0.5)) * T.scale(0.6, 0.8, 1.0), T.translate((2.2, 0.5, 0.7)) * T.scale(0.8, 0.9, 1.0))
            return sb.finish(builder, instancing=mode)
        sb.add_instance("panel", scenes.Transform.translate((0, 0, z)) * scenes.Transform.rotate_y(8.0))
        sb.add_instance("panel", scenes.Transform.translate((2.6, 0.3, 0.5)) * scenes.Transform.scale(0.6, 0.8, 1.0))
        return sb.finish(builder, instancing=mode)
    sb.add_quad(panel, red, UV=uv, **kw)
    return sb.finish(builder)


def rays_at_panel(n=20000, seed=3):
    rng = np.random.default_rng(seed)
    rays = np.zeros(n, abi.RAY_DT)
    rays["o"] = np.stack([rng.uniform(-1.5, 1.5, n), rng.uniform(0.0, 2.6, n), np.full(n, -2.0)], 1).astype(np.float32)
    tgt = np.stack([rng.uniform(-1.4, 1.4, n), rng.uniform(0.1, 2.4, n), np.full(n, 0.97)], 1)
    d = tgt - rays["o"]
    rays["d"] = (d / np.linalg.norm(d, axis=1)[:, None]).astype(np.float32)
    rays["t_max"] = np.inf
    return rays


def test_oracle_alpha_mask_equals_cut_out_geometry(oracle):
    from rs_pbrt_amd import lib
    rays = rays_at_panel()
    masked, cut = masked_scene(lib.bvh_build), masked_scene(lib.bvh_build, cut=True)
    hm, hc = oracle.trace(masked, rays), oracle.trace(cut, rays)
    # the same distances except within an ulp of a cell border; a hole lets the ray reach the wall at z = 2.97
    same = np.isclose(hm["t"], hc["t"], rtol=1e-5)
    assert same.mean() > 0.995
    on_panel = np.isclose((rays["o"][:, 2] + hm["t"] * rays["d"][:, 2]), 0.97, atol=1e-4)
    assert 0.3 < on_panel.mean() < 0.6 and (~on_panel).sum() > 1000
    om, oc = oracle.trace(masked, rays, any_hit=True), oracle.trace(cut, rays, any_hit=True)
    assert (om["prim"] == oc["prim"]).mean() > 0.995
    # shadow_alpha_mask: camera rays see the whole panel, shadow rays the holes
    sh = masked_scene(lib.bvh_build, shadow_only=True)
    hs = oracle.trace(sh, rays)
    assert hs.tobytes() == oracle.trace(masked_scene(lib.bvh_build, plain=True), rays).tobytes()
    short = rays.copy(); short["t_max"] = 3.5   # ends in front of the wall: occluded only by panel cells that are there
    assert np.array_equal(oracle.trace(sh, short, any_hit=True)["prim"], oracle.trace(masked, short, any_hit=True)["prim"])
    # and the pictures agree
    rd = scenes.make_render_desc(64, 48, 16, LOOK, 50.0)
    a = scenes.film_to_rgb(oracle.render(masked, rd, threads=4)["film"]); b = scenes.film_to_rgb(oracle.render(cut, rd, threads=4)["film"])
    assert np.sqrt(np.mean((a - b) ** 2)) < 0.02 and abs(a.mean() - b.mean()) < 3e-3 * b.mean()


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["alpha", "shadow", "instanced-fixed", "instanced-reference"])
def test_gpu_alpha_masks_match_oracle(gpu, oracle, variant):
    import os
    sc = masked_scene(gpu.bvh_build, shadow_only=variant == "shadow", instanced=variant.startswith("instanced"), mode=variant.split("-")[-1] if "-" in variant else "fixed")
    rays = np.concatenate([rays_at_panel(30000, 5), random_rays(30000, 6, -3.0, 3.0)])
    rd = scenes.make_render_desc(64, 48, 16, LOOK, 50.0)
    ref = oracle.render(sc, rd, threads=8, want_li=True)
    for kernel in ("2", "0"):   # the persistent four-box kernel and the reference-order loop
        os.environ["RSPT_TRACE_KERNEL"] = kernel
        try:
            with gpu.DeviceScene(sc) as ds:
                for any_hit in (False, True):
                    assert gpu.trace(ds, rays, any_hit=any_hit).tobytes() == oracle.trace(sc, rays, any_hit=any_hit).tobytes()
                film, st = gpu.render(ds, rd)
        finally:
            os.environ.pop("RSPT_TRACE_KERNEL", None)
        assert st["samples"] == ref["counters"]["samples"] and np.array_equal(film[:, 3], ref["film"][:, 3])
        assert film_rmse(film, ref["film"]) < 1e-5


@pytest.mark.gpu
def test_gpu_image_alpha_mask_and_refusal(gpu, oracle):
    """an image texture as mask (bilinear lookup at level 0: no differentials at the alpha test, mipmap.rs:253-262) under both integrators
    that trace shadow rays in bulk; an emissive mesh with a mask is refused"""
    from rs_pbrt_amd.lib import RsptError
    sb = scenes.SceneBuilder()
    grey = sb.add_material(scenes.matte((0.6, 0.6, 0.6)))
    img = (np.random.default_rng(2).random((16, 16, 3)) > 0.45).astype(np.float32)
    mask = sb.image_texture(img, channels=1, su=2.0, sv=2.0, trilinear=True)
    sb.add_quad([(-4, 0, -4), (-4, 0, 4), (4, 0, 4), (4, 0, -4)], grey)
    sb.add_quad([(-2, 1.5, -2), (2, 1.5, -2), (2, 1.5, 2), (-2, 1.5, 2)], sb.add_material(scenes.matte((0.2, 0.5, 0.2))), UV=[[0, 0], [1, 0], [1, 1], [0, 1]], alpha=mask)
    sb.add_quad([(-1, 4.47, -1), (1, 4.47, -1), (1, 4.47, 1), (-1, 4.47, 1)], grey, emit=(12, 12, 12))
    sc = sb.finish(gpu.bvh_build)
    rd = scenes.make_render_desc(64, 48, 8, ((0, 3.0, -5.0), (0, 1.0, 0), (0, 1, 0)), 50.0)
    with gpu.DeviceScene(sc) as ds:
        film, _ = gpu.render(ds, rd)
        ref = oracle.render(sc, rd, threads=8)
        assert np.array_equal(film[:, 3], ref["film"][:, 3]) and film_rmse(film, ref["film"]) < 1e-5
        rda = scenes.make_render_desc(64, 48, 4, ((0, 3.0, -5.0), (0, 1.0, 0), (0, 1, 0)), 50.0, integrator="ao", ao_samples=8)
        fa, _ = gpu.render(ds, rda)
        ra = oracle.render(sc, rda, threads=8)
        assert np.array_equal(fa[:, 3], ra["film"][:, 3]) and film_rmse(fa, ra["film"]) < 1e-5
    sb.mesh_emit[-1] = None
    sb2 = scenes.SceneBuilder()
    m = sb2.add_material(scenes.matte((0.5, 0.5, 0.5)))
    t = sb2.constant_texture(1.0)
    sb2.add_quad([(-1, 0, -1), (1, 0, -1), (1, 0, 1), (-1, 0, 1)], m, emit=(1, 1, 1))
    sc2 = sb2.finish(gpu.bvh_build)
    sc2.meshes["alpha_tex"][0] = t.index + 1   # behind the builder's back
    with pytest.raises(RsptError) as e:
        gpu.DeviceScene(sc2)
    assert e.value.code == abi.E_UNSUPPORTED


def image_masked_scene(builder, form):
    """the panel scene with masks in the forms rspt_scene_create turns into in-line records (dev_scene.h AlphaMask): "float imagemap" cut-outs under a
    UVMapping2D and constants — and one form it must leave to the general evaluator (a checkerboard next to image masks)"""
    rng = np.random.default_rng(11)
    sb = scenes.SceneBuilder()
    grey = sb.add_material(scenes.matte((0.6, 0.6, 0.6)))
    green = sb.add_material(scenes.matte((0.2, 0.55, 0.2)))
    sb.add_quad([(-4, 0, -4), (-4, 0, 4), (4, 0, 4), (4, 0, -4)], grey)
    sb.add_quad([(-4, 0, 2.97), (-4, 5, 2.97), (4, 5, 2.97), (4, 0, 2.97)], grey)
    sb.add_quad([(-1, 4.47, -1), (1, 4.47, -1), (1, 4.47, 1), (-1, 4.47, 1)], grey, emit=(12, 12, 12))
    sb.add_point_light((0, 2, -4), (25, 25, 25))
    img1 = (rng.random((16, 16, 3)) > 0.45).astype(np.float32)
    img3 = (rng.random((8, 12, 3)) > 0.4).astype(np.float32) * rng.random((8, 12, 3)).astype(np.float32)   # three channels: only the first one counts
    panel = np.array([(-1.0, 0.2, 0.97), (1.0, 0.2, 0.97), (1.0, 2.2, 0.97), (-1.0, 2.2, 0.97)], np.float32)
    uv = [[0, 0], [1, 0], [1, 1], [0, 1]]
    leaf = np.array([(-2.6, 0.3, 0.2), (-1.4, 0.5, 0.4), (-2.0, 1.9, 0.3)], np.float32)   # a mesh without uvs: (0,0) (1,0) (1,1), triangle.rs:97-112
    if form == "repeat":       # EWA filter (the default), repeat wrap, scaled and shifted uvs
        m = sb.image_texture(img1, channels=1, su=2.0, sv=3.0, du=0.25, dv=-0.4)
        sb.add_quad(panel, green, UV=uv, alpha=m)
        sb.add_mesh(leaf, [[0, 1, 2]], green, alpha=m)
    elif form == "clamp-trilinear":
        m = sb.image_texture(img1, channels=1, su=1.5, sv=1.5, du=-0.2, dv=-0.2, wrap="clamp", trilinear=True)
        sb.add_quad(panel, green, UV=uv, alpha=m)
        sb.add_mesh(leaf, [[0, 1, 2]], green, alpha=sb.image_texture(img3, channels=3, wrap="black", su=1.3, sv=0.8))
    elif form == "shadow":     # alpha and shadowalpha are different textures; a second mesh has the shadow mask only
        m = sb.image_texture(img1, channels=1, su=2.0, sv=2.0)
        ms = sb.image_texture(img3, channels=3, su=1.0, sv=2.0)
        sb.add_quad(panel, green, UV=uv, alpha=m, shadow_alpha=ms)
        sb.add_mesh(leaf, [[0, 1, 2]], green, shadow_alpha=m)
    elif form == "constants":  # "float alpha 0" removes a mesh, 1 keeps it; a degenerate uv set under a shadow mask (triangle.rs:611-621)
        sb.add_quad(panel, green, UV=uv, alpha=sb.constant_texture(0.0))
        sb.add_mesh(leaf, [[0, 1, 2]], green, alpha=sb.constant_texture(1.0))
        sb.add_mesh(leaf + np.float32(0.8), [[0, 1, 2]], green, UV=[[0.5, 0.5], [0.5, 0.5], [0.5, 0.5]], shadow_alpha=sb.constant_texture(1.0))
    elif form in ("instanced", "instanced-reference"):
        m = sb.image_texture(img1, channels=1, su=2.0, sv=2.0)
        sb.begin_object("panel")
        sb.add_mesh(panel - np.array([0, 0, 0.97], np.float32), [[0, 1, 2], [0, 2, 3]], green, UV=uv, alpha=m)
        sb.add_mesh(np.array([(-0.2, 0, 0.3), (0.2, 0, 0.3), (0, 0.4, 0.3)], np.float32), [[0, 1, 2]], green, shadow_alpha=m)
        sb.end_object()
        sb.add_instance("panel", scenes.Transform.translate((0, 0, 0.97)) * scenes.Transform.rotate_y(8.0))
        sb.add_instance("panel", scenes.Transform.translate((2.6, 0.3, 0.5)) * scenes.Transform.scale(0.6, 0.8, 1.0))
        return sb.finish(builder, instancing="reference" if form.endswith("reference") else "fixed")
    elif form == "mixed-with-a-graph":   # one mask is a checkerboard: the whole scene stays with the general evaluator
        sb.add_quad(panel, green, UV=uv, alpha=sb.image_texture(img1, channels=1, su=2.0, sv=2.0))
        sb.add_mesh(leaf, [[0, 1, 2]], green, alpha=sb.checkerboard_texture(sb.constant_texture(0.0), sb.constant_texture(1.0), su=3.0, sv=3.0))
    else:
        raise ValueError(form)
    return sb.finish(builder)


def test_oracle_image_alpha_masks_remove_light_blockers(oracle):
    """CPU: the forms above change what the oracle renders (masks are really applied), "float alpha 0" equals leaving the mesh out"""
    sc = image_masked_scene(oracle.bvh_build, "constants")
    rays = rays_at_panel(4000, 9)
    hit = oracle.trace(sc, rays, any_hit=False)
    assert (hit["prim"][np.abs(rays["o"][:, 0] + rays["d"][:, 0] * 2.97 / rays["d"][:, 2]) < 0.9] != 0xffffffff).all()   # they reach the wall behind the removed panel
    sb = image_masked_scene(oracle.bvh_build, "repeat")
    a = oracle.render(sb, scenes.make_render_desc(32, 24, 4, LOOK, 50.0), threads=8)["film"]
    b = oracle.render(masked_scene(oracle.bvh_build, plain=True), scenes.make_render_desc(32, 24, 4, LOOK, 50.0), threads=8)["film"]
    assert not np.array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["repeat", "clamp-trilinear", "shadow", "constants", "instanced", "instanced-reference", "mixed-with-a-graph"])
def test_gpu_inline_alpha_masks_match_oracle_and_the_general_evaluator(gpu, oracle, form, monkeypatch):
    """k_trace_w4<.., ALPHA = 2> (kernels.h alpha_simple: masks as in-line records, no texture-graph interpreter in the traversal) against the oracle and
    against the general path (RSPT_ALPHA_SIMPLE=0: alpha_pass -> tex_eval): hit records byte for byte, every sample's radiance bit for bit"""
    sc = image_masked_scene(gpu.bvh_build, form)
    rays = np.concatenate([rays_at_panel(30000, 5), random_rays(30000, 6, -3.0, 3.0)])
    rd = scenes.make_render_desc(64, 48, 8, LOOK, 50.0)
    ref = oracle.render(sc, rd, threads=8, want_li=True)
    ref_hits = {a: oracle.trace(sc, rays, any_hit=a).tobytes() for a in (False, True)}
    for simple in ("1", "0"):
        monkeypatch.setenv("RSPT_ALPHA_SIMPLE", simple)
        with gpu.DeviceScene(sc) as ds:
            for any_hit in (False, True):
                assert gpu.trace(ds, rays, any_hit=any_hit).tobytes() == ref_hits[any_hit], (simple, any_hit)
            li = gpu.render_samples(ds, rd)[0]
            film, st = gpu.render(ds, rd)
        assert np.array_equal(li, ref["li"]), (simple, int((li != ref["li"]).any(axis=2).sum()))
        assert st["samples"] == ref["counters"]["samples"] and np.array_equal(film[:, 3], ref["film"][:, 3])


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(sampler="02sequence"), dict(sampler="random", integrator="directlighting", max_depth=3),
                                dict(sampler="stratified", strat=(2, 2), integrator="ao", ao_samples=4), dict(integrator="directlighting", max_depth=12)],
                         ids=["02sequence-path", "random-directlighting", "stratified-ao", "sobol-directlighting-per-lane"])
def test_gpu_per_lane_kernels_with_inline_alpha_masks(gpu, oracle, kw, monkeypatch):
    """the per-lane kernels (k_tile_serial, k_lane_dl) on a scene whose masks have the in-line form: they take the four-box traversal
    (trace_serial.h traverse_w4<.., ALPHA>) instead of the reference-order loop — same films as the oracle's and as the general evaluator's"""
    films = []
    for form in ("repeat", "shadow"):
        sc = image_masked_scene(gpu.bvh_build, form)
        rd = scenes.make_render_desc(64, 48, 4, LOOK, 50.0, **kw)
        ref = oracle.render_integrator(sc, rd, "direct", threads=8) if kw.get("integrator") == "directlighting" else oracle.render(sc, rd, threads=8)
        for simple in ("1", "0"):
            monkeypatch.setenv("RSPT_ALPHA_SIMPLE", simple)
            with gpu.DeviceScene(sc) as ds:
                film, st = gpu.render(ds, rd)
            assert st["samples"] == ref["counters"]["samples"] and np.array_equal(film[:, 3], ref["film"][:, 3])
            assert film_rmse(film, ref["film"]) < 2e-5, (form, simple)
            films.append(film)
        assert np.array_equal(films[-1], films[-2]), form   # one lane per tile / per camera sample: the accumulation order is fixed, the two evaluators agree bit for bit


@pytest.mark.gpu
@pytest.mark.parametrize("mode,integrator", [("fixed", "path"), ("reference", "path"), ("fixed", "volpath"), ("fixed", "directlighting")])
def test_moving_instances_of_alpha_masked_meshes(gpu, oracle, mode, integrator):
    """round 5: a moving TransformedPrimitive whose object carries an alpha mask (primitive.rs:216-265 around triangle.rs:313-330): k_trace_w4<INST, ALPHA, ANIM> — the mask
    test is a function of the hit's uv, whatever Transform the instance was entered with.  Per-sample radiance bit for bit; rspt_scene_create used to refuse the scene."""
    sc = masked_scene(gpu.bvh_build, instanced=True, mode=mode, moving=True)
    assert int(sc.instances["animated"].sum()) == 2
    kw = dict(integrator=integrator)
    if integrator == "directlighting":
        kw.update(direct_strategy="all", light_samples=[1] * sc.desc.n_lights)
    rd = scenes.make_render_desc(80, 60, 8, LOOK, 45.0, shutter=(0.0, 1.0), **kw)
    with gpu.DeviceScene(sc) as ds:
        film, st = gpu.render(ds, rd)
        li, _ = gpu.render_samples(ds, rd)
    ref = (oracle.render_integrator(sc, rd, "direct", strategy="all", light_samples=kw["light_samples"], threads=8, want_li=True) if integrator == "directlighting"
           else oracle.render(sc, rd, threads=8, want_li=True))
    assert st["samples"] == ref["counters"]["samples"] and st["nan_samples"] == 0
    assert np.array_equal(film[:, 3], ref["film"][:, 3]) and np.array_equal(li, ref["li"])
    still = oracle.render(sc, scenes.make_render_desc(80, 60, 8, LOOK, 45.0, shutter=(0.0, 0.0)), threads=8) if integrator == "path" else None
    if still is not None:
        assert np.abs(scenes.film_to_rgb(film) - scenes.film_to_rgb(still["film"])).mean() > 1e-4   # the motion is in the picture


@pytest.mark.gpu
@pytest.mark.parametrize("switch", [{"RSPT_TRACE_KERNEL": "0"}, {"RSPT_INSTANCE_KERNEL": "0"}, {"RSPT_ANIM_W4": "0"}, {"RSPT_COUNTERS": "1"}])
def test_moving_masked_instances_under_every_trace_switch(gpu, oracle, monkeypatch, switch):
    """round 6: with RSPT_TRACE_KERNEL=0 / RSPT_INSTANCE_KERNEL=0 / RSPT_COUNTERS=1 a scene with moving instances AND alpha masks used to fall to
    k_trace<.., INST, ALPHA> without the interpolation (the start key's Transform in the traversal, inst_at(time) in the shade stage): a silently wrong picture.  The
    reference-order loop now has its <INST, ALPHA, ANIM> instantiation; every switch gives the oracle's samples bit for bit."""
    sc = masked_scene(gpu.bvh_build, instanced=True, mode="fixed", moving=True)
    rd = scenes.make_render_desc(80, 60, 4, LOOK, 45.0, shutter=(0.0, 1.0))
    for k, v in switch.items():
        monkeypatch.setenv(k, v)
    with gpu.DeviceScene(sc) as ds:
        li, st = gpu.render_samples(ds, rd)
    ref = oracle.render(sc, rd, threads=8, want_li=True)
    assert st["samples"] == ref["counters"]["samples"] and np.array_equal(li, ref["li"])
