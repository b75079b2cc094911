"""Cross-feature fuzz: random rooms (tests/util.random_scene: every material recipe, textures, bump maps, null surfaces, all light
kinds) with fog boxes, object instances (both behaviours), alpha-masked quads, crop windows, shards and sample ranges thrown in, under
every integrator x sampler combination the library accepts.  Film weights bit for bit; radiance within the bump-map bar of
tests/test_gpu_render.py::test_random_scenes_fuzz."""
import numpy as np
import pytest

from rs_pbrt_amd import abi, scenes
from tests.util import GALLERY_LOOK_AT, film_rmse, random_scene

pytestmark = pytest.mark.gpu
PIXEL = ("random", "02sequence", "stratified", "maxmindist")


def build(builder, seed):
    rng = np.random.default_rng(1000 + seed)
    sb = random_scene(builder, seed).builder
    integrator = ["path", "path", "volpath", "ao"][seed % 4]
    feats = []
    if integrator == "volpath" or rng.random() < 0.3:   # media are ignored by path / ao, as in the reference
        fog = sb.add_medium(sigma_a=tuple(rng.uniform(0.0, 0.05, 3)), sigma_s=tuple(rng.uniform(0.05, 0.4, 3)), g=float(rng.uniform(-0.4, 0.7)))
        lo = rng.uniform([-3.5, 0.3, -2.5], [-0.5, 1.2, 0.5])
        sb.add_box(tuple(lo), tuple(lo + rng.uniform([1.5, 1.5, 1.5], [4.0, 3.5, 4.0])), None, medium=(fog, None))
        feats.append("fog")
    mode = "fixed"
    if integrator != "volpath" and rng.random() < 0.6:   # volpath refuses instances
        mode = ["fixed", "reference"][int(rng.integers(2))]
        sb.begin_object("thing")
        m = sb.add_material(scenes.plastic(tuple(rng.uniform(0.1, 0.8, 3)), (0.3, 0.3, 0.3), 0.15))
        sb.add_box((-0.3, 0.0, -0.3), (0.3, 0.7, 0.3), m)
        sb.add_mesh(np.array([(-0.5, 0.8, 0), (0.5, 0.8, 0), (0, 1.3, 0.2)], np.float32), [[0, 1, 2]], m)
        sb.end_object()
        for _ in range(int(rng.integers(1, 4))):
            t = scenes.Transform.translate(tuple(rng.uniform([-3.5, 0.0, -2.0], [3.5, 2.5, 3.5]))) * scenes.Transform.rotate_y(float(rng.uniform(0, 360))) * \
                scenes.Transform.scale(*(float(v) for v in rng.uniform(0.6, 1.6, 3)))
            sb.add_instance("thing", t)
        feats.append("instances-" + mode)
    if rng.random() < 0.5:
        mask = sb.checkerboard_texture(sb.constant_texture(0.0), sb.constant_texture(1.0), su=float(rng.integers(2, 7)), sv=float(rng.integers(2, 7)))
        c = rng.uniform([-3, 0.5, 0.0], [3, 3.0, 3.0])
        kw = dict(alpha=mask) if rng.random() < 0.6 else dict(shadow_alpha=mask)
        sb.add_quad([tuple(c + (-0.8, -0.5, 0)), tuple(c + (0.8, -0.5, 0.1)), tuple(c + (0.8, 0.5, 0.1)), tuple(c + (-0.8, 0.5, 0))],
                    sb.add_material(scenes.matte(tuple(rng.uniform(0.2, 0.8, 3)))), UV=[[0, 0], [1, 0], [1, 1], [0, 1]], **kw)
        feats.append("alpha")
    sc = sb.finish(builder, instancing=mode)
    sampler = ["sobol", "halton"][int(rng.integers(2))]
    if integrator == "path" and (seed // 2) % 2 == 0:
        sampler = PIXEL[(seed // 4) % 4]
    kw = dict(sampler=sampler, integrator=integrator, max_depth=1 + seed % 5, strat=(2, 2), dimensions=int(rng.integers(1, 6)),
              light_strategy=[abi.LIGHTS_SPATIAL, abi.LIGHTS_POWER, abi.LIGHTS_UNIFORM][seed % 3], ao_samples=int(rng.integers(1, 6)))
    if rng.random() < 0.4:
        kw["crop"] = (0.1, 0.85, 0.2, 0.95)
    if rng.random() < 0.4:
        kw["shard"] = (int(rng.integers(3)), 3, int(rng.integers(1, 4)))
    if sampler in ("sobol", "halton") and rng.random() < 0.4:
        kw["sample_range"] = (1, 2)
    if sampler not in PIXEL and rng.random() < 0.3:
        kw["lens_radius"], kw["focal_distance"] = 0.03, 6.0
    return sc, kw, integrator, feats


@pytest.mark.parametrize("seed", list(range(401, 433)))
def test_cross_feature_fuzz(gpu, oracle, seed):
    sc, kw, integrator, feats = build(gpu.bvh_build, seed)
    rd = scenes.make_render_desc(56, 40, 4, GALLERY_LOOK_AT, 55, **kw)
    ref = oracle.render(sc, rd, threads=8)
    with gpu.DeviceScene(sc) as ds:
        film, st = gpu.render(ds, rd)
    what = (integrator, kw["sampler"], feats, {k: v for k, v in kw.items() if k in ("crop", "shard", "sample_range", "max_depth")})
    assert st["samples"] == ref["counters"]["samples"], what
    assert np.array_equal(film[:, 3], ref["film"][:, 3]), what
    assert st["nan_samples"] == ref["counters"]["nan_samples"], what
    assert film_rmse(film, ref["film"]) < 3e-4, (what, film_rmse(film, ref["film"]))
