"""VolPathIntegrator::li on the GPU (rs_pbrt_amd/csrc/vol.h) against the oracle (volpath.rs:60-347, homogeneous.rs, medium.rs).
Radiance goes through expf / logf (the medium) and sinf / cosf (phase function and BSDF sampling), which glibc_libm.h evaluates exactly as
the host libm does: per-sample radiance is bit-identical (tests/test_gpu_render.py); the bars here are the film's (weights exact, RMSE 1e-5)."""
import numpy as np
import pytest

from rs_pbrt_amd import abi, scenes
from tests.test_volpath import slab_scene
from tests.util import film_rmse

pytestmark = pytest.mark.gpu
LOOK = ((0, 1.0, -6.0), (0, 1.0, 0), (0, 1, 0))


def fog_room(builder, g=0.4, area_light=True, glass=False, thin=1.0):
    """a room with a box of fog in it (a medium boundary without material), a diffuse block inside the fog, a point light outside it and
    an area light above: paths scatter in the medium, on surfaces inside and outside it, and shadow rays cross its boundary"""
    sb = scenes.SceneBuilder()
    grey = sb.add_material(scenes.matte((0.6, 0.6, 0.6)))
    red = sb.add_material(scenes.matte((0.7, 0.25, 0.2)))
    fog = sb.add_medium(sigma_a=(0.02, 0.03, 0.05), sigma_s=(0.35, 0.3, 0.25), g=g, scale=thin)
    sb.add_quad([(-4, 0, -4.03), (-4, 0, 4.03), (4, 0, 4.03), (4, 0, -4.03)], grey)
    sb.add_quad([(-4, 0, 2.97), (-4, 5.03, 2.97), (4, 5.03, 2.97), (4, 0, 2.97)], grey)
    sb.add_box((-1.53, 0.07, -1.51), (1.49, 2.53, 1.52), None, medium=(fog, None))
    sb.add_box((-0.52, 0.31, -0.49), (0.51, 1.27, 0.53), red, medium=(fog, fog))            # inside the fog: not a transition, rays keep their medium
    if glass:
        sb.add_box((1.93, 0.09, -0.8), (2.87, 1.1, 0.1), sb.add_material(scenes.glass((1, 1, 1), (1, 1, 1), 1.5)))
    if area_light:
        sb.add_quad([(-1, 4.47, -1), (1, 4.47, -1), (1, 4.47, 1), (-1, 4.47, 1)], grey, emit=(14, 13, 12))
    sb.add_point_light((2.5, 3.1, -3.2), (22, 22, 25))
    return sb.finish(builder)


@pytest.mark.parametrize("case", ["fog", "fog-isotropic", "fog-glass", "fog-delta-only", "fog-dense", "fog-halton", "fog-uniform-lights"])
def test_gpu_volpath_matches_oracle(gpu, oracle, case):
    sc = fog_room(gpu.bvh_build, g=0.0 if case == "fog-isotropic" else (-0.3 if case == "fog-dense" else 0.4), glass=case == "fog-glass",
                  area_light=case != "fog-delta-only", thin=6.0 if case == "fog-dense" else 1.0)
    kw = dict(sampler="halton") if case == "fog-halton" else {}
    if case == "fog-uniform-lights":
        kw["light_strategy"] = abi.LIGHTS_UNIFORM
    rd = scenes.make_render_desc(64, 48, 16, LOOK, 55.0, integrator="volpath", max_depth=7 if case == "fog-dense" else 5, **kw)
    ref = oracle.render(sc, rd, threads=8)
    with gpu.DeviceScene(sc) as ds:
        film, st = gpu.render(ds, rd)
    assert st["samples"] == ref["counters"]["samples"] and st["truncated_paths"] == 0
    assert np.array_equal(film[:, 3], ref["film"][:, 3])
    assert film_rmse(film, ref["film"]) < 1e-5
    assert scenes.film_to_rgb(film).mean() > 0.02


def test_gpu_beer_lambert_and_media_free_scenes(gpu, oracle):
    """the absorbing slab of tests/test_volpath.py (closed form) and a scene without media, where volpath and path differ only by the
    estimate's BSDF-sampled half"""
    sc = slab_scene(gpu.bvh_build, (0.7, 0.7, 0.7), (0, 0, 0))
    rd = scenes.make_render_desc(48, 48, 64, ((0, 0, -6.0), (0, 0, 0), (0, 1, 0)), 12.0, integrator="volpath")
    with gpu.DeviceScene(sc) as ds:
        film, _ = gpu.render(ds, rd)
        rdp = scenes.make_render_desc(48, 48, 4, ((0, 0, -6.0), (0, 0, 0), (0, 1, 0)), 12.0)
        plain, _ = gpu.render(ds, rdp)   # the path integrator ignores the medium
    assert np.allclose(scenes.film_to_rgb(film).reshape(-1, 3).mean(0), np.array([3.0, 2.0, 1.0]) * np.exp(-0.7), rtol=0.01)
    assert np.allclose(scenes.film_to_rgb(plain).reshape(-1, 3).mean(0), [3.0, 2.0, 1.0], rtol=1e-5)
    ref = oracle.render(sc, rd, threads=8)
    assert np.array_equal(film[:, 3], ref["film"][:, 3]) and film_rmse(film, ref["film"]) < 1e-5
    cb = scenes.cornell_box(gpu.bvh_build)
    rdc = scenes.cornell_render_desc(res=64, spp=16)
    rdc.integrator = abi.INTEGRATOR_VOLPATH
    refc = oracle.render(cb, rdc, threads=8)
    with gpu.DeviceScene(cb) as ds:
        fc, _ = gpu.render(ds, rdc)
    assert np.array_equal(fc[:, 3], refc["film"][:, 3]) and film_rmse(fc, refc["film"]) < 1e-5


def test_gpu_volpath_escaping_paths_and_infinite_light(gpu, oracle):
    """no walls: most scattered rays leave the scene, which ends the path (volpath.rs:338-339) after the infinite light was added for
    bounces == 0 along the SCATTERED ray (:332-337)"""
    sb = scenes.SceneBuilder()
    fog = sb.add_medium(sigma_a=(0.05, 0.05, 0.05), sigma_s=(0.5, 0.6, 0.7), g=0.6)
    sb.add_box((-1.53, -1.07, -1.51), (1.49, 1.53, 1.52), None, medium=(fog, None))
    sb.add_box((-0.4, -0.4, -0.4), (0.45, 0.35, 0.4), sb.add_material(scenes.matte((0.5, 0.5, 0.5))), medium=(fog, fog))
    sb.add_infinite_light(L=(0.6, 0.7, 0.9))
    sb.add_point_light((3, 3, -3), (30, 30, 30))
    sc = sb.finish(gpu.bvh_build)
    rd = scenes.make_render_desc(64, 48, 16, ((0, 0.5, -6.0), (0, 0, 0), (0, 1, 0)), 40.0, integrator="volpath")
    ref = oracle.render(sc, rd, threads=8)
    with gpu.DeviceScene(sc) as ds:
        film, st = gpu.render(ds, rd)
    assert np.array_equal(film[:, 3], ref["film"][:, 3]) and film_rmse(film, ref["film"]) < 1e-5 and st["truncated_paths"] == 0


def test_gpu_volpath_with_textured_materials(gpu, oracle):
    """the texture stage in front of k_vol_shade (image textures with camera-ray differentials, scale textures, dropped lobes): the room of
    test_textured_materials_match_oracle with a box of fog in front of the slabs; the camera ray loses its differentials when it
    crosses the fog's boundary (isect.spawn_ray, volpath.rs:141-145), so the slabs behind the fog are filtered without footprint"""
    from tests.util import TEXTURED_LOOK_AT, textured_room
    sb = textured_room(gpu.bvh_build, bump=False).builder
    fog = sb.add_medium(sigma_a=(0.02, 0.02, 0.03), sigma_s=(0.12, 0.1, 0.08), g=0.2)
    sb.add_box((-1.03, 0.21, -0.97), (2.51, 3.07, 1.03), None, medium=(fog, None))
    sb.add_point_light((-3.1, 4.2, -2.2), (20, 20, 20))
    sc = sb.finish(gpu.bvh_build)
    rd = scenes.make_render_desc(96, 72, 16, TEXTURED_LOOK_AT, 45, max_depth=4, integrator="volpath")
    ref = oracle.render(sc, rd, threads=8)
    with gpu.DeviceScene(sc) as ds:
        film, st = gpu.render(ds, rd)
    assert np.array_equal(film[:, 3], ref["film"][:, 3]) and st["truncated_paths"] == 0
    assert film_rmse(film, ref["film"]) < 2e-5


@pytest.mark.parametrize("seed", list(range(201, 213)))
def test_volpath_random_scenes_fuzz(gpu, oracle, seed):
    """the random rooms of test_random_scenes_fuzz (every material recipe, textures, bump maps, null surfaces, area / delta / sometimes
    infinite lights) with one or two overlapping boxes of fog, both samplers, all light strategies.  Where two media overlap, the
    boundary a ray crossed last decides its medium (Interaction::get_medium), whatever that means physically."""
    from tests.util import GALLERY_LOOK_AT, random_scene
    sb = random_scene(gpu.bvh_build, seed).builder
    rng = np.random.default_rng(seed)
    for k in range(1 + seed % 2):
        fog = sb.add_medium(sigma_a=tuple(rng.uniform(0.0, 0.08, 3)), sigma_s=tuple(rng.uniform(0.05, 0.5, 3)), g=float(rng.uniform(-0.5, 0.8)))
        lo = rng.uniform([-3.5, 0.3, -2.5], [-0.5, 1.5, 0.5])
        sb.add_box(tuple(lo), tuple(lo + rng.uniform([1.5, 1.5, 1.5], [4.0, 3.5, 4.0])), None, medium=(fog, None))
    sc = sb.finish(gpu.bvh_build)
    rd = scenes.make_render_desc(56, 40, 8, GALLERY_LOOK_AT, 55, max_depth=2 + seed % 5, sampler="halton" if seed % 2 else "sobol", integrator="volpath",
                                 light_strategy=[abi.LIGHTS_SPATIAL, abi.LIGHTS_POWER, abi.LIGHTS_UNIFORM][seed % 3])
    ref = oracle.render(sc, rd, threads=8)
    with gpu.DeviceScene(sc) as ds:
        film, st = gpu.render(ds, rd)
    assert np.array_equal(film[:, 3], ref["film"][:, 3]) and st["truncated_paths"] == 0
    assert st["nan_samples"] == ref["counters"]["nan_samples"]
    assert film_rmse(film, ref["film"]) < 3e-4


def test_gpu_volpath_refusals(gpu):
    from rs_pbrt_amd.lib import RsptError
    sb = scenes.SceneBuilder()
    m = sb.add_material(scenes.matte((0.5, 0.5, 0.5)))
    sb.add_quad([(-1, 0, -1), (1, 0, -1), (1, 0, 1), (-1, 0, 1)], m)
    sc = sb.finish(gpu.bvh_build)
    sc.meshes["medium_inside"][0] = 3   # no such medium
    with pytest.raises(RsptError) as e:
        gpu.DeviceScene(sc)
    assert e.value.code == abi.E_INVALID


@pytest.mark.parametrize("mode", ["reference", "fixed"])
def test_gpu_volpath_with_object_instances(gpu, oracle, mode):
    """VERDICT r2 missing #5: fog + ObjectInstances.  Transform::transform_surface_interaction builds the instanced hit from
    SurfaceInteraction::default(): it has no medium interface — a ray that leaves an instanced surface travels in no medium, although it
    may sit in the middle of the fog — and in v0.9.12 no primitive: `li` passes through it (no BSDF) and VisibilityTester::tr neither
    blocks nor attenuates at it (light.rs:216-229).  Pyramids inside and outside a box of fog (scaled, rotated, one identity instance,
    one single-triangle object), both instancing behaviours, every camera sample against the oracle."""
    from tests.test_instancing import PYR, PYR_IDX
    sb = scenes.SceneBuilder()
    grey = sb.add_material(scenes.matte((0.6, 0.6, 0.6)))
    red = sb.add_material(scenes.plastic((0.6, 0.2, 0.15), (0.3, 0.3, 0.3), 0.15))
    fog = sb.add_medium(sigma_a=(0.02, 0.03, 0.05), sigma_s=(0.3, 0.28, 0.25), g=0.3)
    sb.add_quad([(-4, 0, -4.03), (-4, 0, 4.03), (4, 0, 4.03), (4, 0, -4.03)], grey)
    sb.add_quad([(-4, 0, 2.97), (-4, 5.03, 2.97), (4, 5.03, 2.97), (4, 0, 2.97)], grey)
    sb.add_box((-1.83, 0.07, -1.51), (1.79, 2.53, 1.52), None, medium=(fog, None))
    sb.add_quad([(-1, 4.47, -1), (1, 4.47, -1), (1, 4.47, 1), (-1, 4.47, 1)], grey, emit=(14, 13, 12))
    sb.add_point_light((2.5, 3.1, -3.2), (22, 22, 25))
    sb.begin_object("pyr")
    sb.add_mesh(PYR * np.float32(0.6), PYR_IDX, red, medium=(fog, fog))
    sb.end_object()
    sb.begin_object("one")
    sb.add_mesh(PYR[:3] * np.float32(0.8), [[0, 1, 2]], red)
    sb.end_object()
    T = scenes.Transform
    sb.add_instance("pyr", T.translate((-0.6, 0.4, 0.1)) * T.rotate_y(30.0))                  # inside the fog
    sb.add_instance("pyr", T.translate((0.7, 0.5, -0.3)) * T.scale(1.3, 0.8, 1.1))            # inside the fog
    sb.add_instance("pyr", T.translate((2.9, 0.3, 0.4)))                                      # outside
    sb.add_instance("one", T.translate((-2.6, 1.2, 0.0)))
    sb.add_instance("pyr", T.identity())                                                       # Q10
    sc = sb.finish(gpu.bvh_build, instancing=mode)
    rd = scenes.make_render_desc(64, 48, 16, LOOK, 55.0, integrator="volpath", max_depth=5)
    ref = oracle.render(sc, rd, threads=8, want_li=True)
    with gpu.DeviceScene(sc) as ds:
        film, st = gpu.render(ds, rd)
        li, _ = gpu.render_samples(ds, rd)
    assert np.array_equal(film[:, 3], ref["film"][:, 3]) and st["truncated_paths"] == 0 and st["nan_samples"] == 0
    assert np.array_equal(li, ref["li"]), "%d of %d camera samples differ" % (int((li != ref["li"]).any(axis=2).sum()), li.shape[0] * li.shape[1])
    assert film_rmse(film, ref["film"]) < 1e-6


def grid_room(builder, kind="cloud", instances=False):
    """the fog room with a GridDensityMedium ("heterogeneous") in the box: a 6 x 5 x 4 density with a dense core, empty corners and a
    maximum in one voxel (ratio / delta tracking against the majorant), or a uniform grid; optionally object instances in and around it"""
    sb = scenes.SceneBuilder()
    grey = sb.add_material(scenes.matte((0.6, 0.6, 0.6)))
    red = sb.add_material(scenes.plastic((0.6, 0.2, 0.15), (0.3, 0.3, 0.3), 0.15))
    rng = np.random.default_rng(17)
    if kind == "uniform":
        dens = np.ones((4, 5, 6), np.float32)
    else:
        z, y, x = np.mgrid[0:4, 0:5, 0:6]
        dens = np.exp(-(((x - 2.5) / 2.0) ** 2 + ((y - 2.0) / 1.5) ** 2 + ((z - 1.5) / 1.2) ** 2)).astype(np.float32) * 2.0 + rng.uniform(0, 0.3, (4, 5, 6)).astype(np.float32)
        dens[0, 0, :2] = 0.0; dens[3, 4, 4:] = 0.0; dens[2, 2, 3] = 3.5
    fog = sb.add_grid_medium(dens, p0=(-1.53, 0.07, -1.51), p1=(1.49, 2.53, 1.52), sigma_a=(0.03, 0.04, 0.06), sigma_s=(0.5, 0.45, 0.4), g=0.35)
    sb.add_quad([(-4, 0, -4.03), (-4, 0, 4.03), (4, 0, 4.03), (4, 0, -4.03)], grey)
    sb.add_quad([(-4, 0, 2.97), (-4, 5.03, 2.97), (4, 5.03, 2.97), (4, 0, 2.97)], grey)
    sb.add_box((-1.53, 0.07, -1.51), (1.49, 2.53, 1.52), None, medium=(fog, None))
    sb.add_box((-0.52, 0.31, -0.49), (0.51, 1.27, 0.53), red, medium=(fog, fog))
    sb.add_quad([(-1, 4.47, -1), (1, 4.47, -1), (1, 4.47, 1), (-1, 4.47, 1)], grey, emit=(14, 13, 12))
    sb.add_point_light((2.5, 3.1, -3.2), (22, 22, 25))
    if instances:
        from tests.test_instancing import PYR, PYR_IDX
        sb.begin_object("pyr"); sb.add_mesh(PYR * np.float32(0.5), PYR_IDX, red, medium=(fog, fog)); sb.end_object()
        T = scenes.Transform
        sb.add_instance("pyr", T.translate((-0.9, 1.4, 0.2)) * T.rotate_y(40.0))
        sb.add_instance("pyr", T.translate((2.7, 0.3, 0.4)))
    return sb.finish(builder)


@pytest.mark.parametrize("name", ["random", "02sequence", "stratified", "maxmindist"])
def test_gpu_volpath_under_the_pixel_samplers(gpu, oracle, name):
    """VERDICT r2 #5 / missing #4: VolPathIntegrator::li per lane (vol_serial.h) with the tile's PCG stream read in program order — the
    homogeneous fog room, every camera sample against the oracle"""
    sc = fog_room(gpu.bvh_build)
    rd = scenes.make_render_desc(48, 40, 16, LOOK, 55.0, integrator="volpath", max_depth=5, sampler=name, strat=(4, 4))
    ref = oracle.render(sc, rd, threads=8, want_li=True)
    with gpu.DeviceScene(sc) as ds:
        film, st = gpu.render(ds, rd)
        li, _ = gpu.render_samples(ds, rd)
    assert np.array_equal(film[:, 3], ref["film"][:, 3]) and st["truncated_paths"] == 0 and st["nan_samples"] == 0
    assert np.array_equal(li, ref["li"]), "%s: %d of %d camera samples differ" % (name, int((li != ref["li"]).any(axis=2).sum()), li.shape[0] * li.shape[1])
    assert film_rmse(film, ref["film"]) < 1e-6


@pytest.mark.parametrize("case", ["cloud-random", "cloud-02sequence", "uniform-stratified", "cloud-instances-maxmindist"])
def test_gpu_grid_density_medium(gpu, oracle, case):
    """VERDICT r2 missing #2: GridDensityMedium — trilinear density, ratio-tracking tr with its roulette, delta-tracking sample, the medium
    interaction at the unnormalised ray's parameter, the draws of estimate_direct's second half (its intersect_tr walk moves the stream
    although the half adds nothing) — under the pixel samplers, every camera sample against the oracle; refused with Sobol' / Halton"""
    from rs_pbrt_amd.lib import RsptError
    kind, *rest = case.split("-")
    name = rest[-1]
    sc = grid_room(gpu.bvh_build, kind=kind, instances="instances" in rest)
    rd = scenes.make_render_desc(48, 40, 16, LOOK, 55.0, integrator="volpath", max_depth=5, sampler=name, strat=(4, 4))
    ref = oracle.render(sc, rd, threads=8, want_li=True)
    with gpu.DeviceScene(sc) as ds:
        film, st = gpu.render(ds, rd)
        li, _ = gpu.render_samples(ds, rd)
        with pytest.raises(RsptError) as e:
            gpu.render(ds, scenes.make_render_desc(16, 16, 4, LOOK, 55.0, integrator="volpath"))
        assert e.value.code == abi.E_UNSUPPORTED
        plain, _ = gpu.render(ds, scenes.make_render_desc(16, 16, 4, LOOK, 55.0))   # `path` ignores media: unaffected
    assert np.array_equal(film[:, 3], ref["film"][:, 3]) and st["truncated_paths"] == 0 and st["nan_samples"] == 0
    assert np.array_equal(li, ref["li"]), "%s: %d of %d camera samples differ" % (case, int((li != ref["li"]).any(axis=2).sum()), li.shape[0] * li.shape[1])
    assert film_rmse(film, ref["film"]) < 1e-6
    assert scenes.film_to_rgb(film).mean() > 0.02 and np.isfinite(plain).all()
