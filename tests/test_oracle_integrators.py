"""CPU: the oracle's restatement of DirectLightingIntegrator (directlighting.rs) and WhittedIntegrator (whitted.rs) —
SURVEY §8(f) #4 groundwork; the GPU library does not build these integrators yet (rspt_render* rejects them), so
these tests pin the oracle only:

  * UniformSampleOne with max_depth 1 is, sample for sample, PathIntegrator with max_depth 1 and the uniform light
    strategy (same sampler dimensions: light choice, u_light, u_scattering; same estimate_direct) — that ties the new
    code to the path code the GPU is checked against;
  * UniformSampleAll (sample arrays in dimensions 5.., get_2d skipping them), UniformSampleOne and Whitted estimate
    the same direct illumination: their images agree in the mean;
  * a mirror facing the camera shows the emitter behind the camera: exactly L at max_depth >= 2, 0 at max_depth 1
    (specular_reflect recursion and its depth test); a two-lobe glass pane in front of the emitter transmits
    (1 - F)^2 L at normal incidence and needs max_depth >= 3."""
import numpy as np
import pytest

from rs_pbrt_amd import abi, scenes
from tests.util import film_rmse


def _rgb(film):
    w = np.maximum(film[:, 3:4], 1e-30)
    return film[:, :3] / w


def test_sample_one_depth1_is_path_depth1(oracle):
    sc = scenes.cornell_box(oracle.bvh_build)
    rd = scenes.cornell_render_desc(res=40, spp=8, max_depth=1, light_strategy=abi.LIGHTS_UNIFORM)
    path = oracle.render(sc, rd, threads=4, want_li=True)
    direct = oracle.render_integrator(sc, rd, "direct", strategy="one", threads=4, want_li=True)
    assert np.array_equal(direct["li"], path["li"])
    assert np.array_equal(direct["film"], path["film"])
    assert direct["li"].max() > 1.0  # the ceiling light is in view
    halton = scenes.cornell_render_desc(res=24, spp=8, max_depth=1, light_strategy=abi.LIGHTS_UNIFORM, sampler="halton")
    assert np.array_equal(oracle.render_integrator(sc, halton, "direct", strategy="one", threads=4, want_li=True)["li"],
                          oracle.render(sc, halton, threads=4, want_li=True)["li"])


@pytest.mark.parametrize("sampler", ["sobol", "halton"])
def test_direct_strategies_and_whitted_agree_in_the_mean(oracle, sampler):
    sc = scenes.cornell_box(oracle.bvh_build)
    rd = scenes.cornell_render_desc(res=32, spp=64, max_depth=5, sampler=sampler)
    one = oracle.render_integrator(sc, rd, "direct", strategy="one", threads=8)
    all1 = oracle.render_integrator(sc, rd, "direct", strategy="all", threads=8)
    all4 = oracle.render_integrator(sc, rd, "direct", strategy="all", light_samples=[4, 4], threads=8)
    whitted = oracle.render_integrator(sc, rd, "whitted", threads=8)
    ref = _rgb(all4["film"])
    assert ref.mean() > 0.05
    for r in (one, all1, whitted):
        img = _rgb(r["film"])
        assert abs(img.mean() - ref.mean()) < 0.02 * ref.mean()
        assert film_rmse(r["film"], all4["film"]) < 0.25 * ref.mean()  # noise only: no structural difference
    # shadow rays: sample-all casts one per light and array sample (2 lights), sample-one one per hit
    hits = one["counters"]["bounces"]
    assert all1["counters"]["bounces"] == hits and all4["counters"]["bounces"] == hits
    assert all4["counters"]["rays_any"] > 3 * all1["counters"]["rays_any"] > 4 * one["counters"]["rays_any"] * 0.9
    # more light samples, less noise
    assert film_rmse(all4["film"], one["film"]) > 0  # different estimators


def _mirror_scene(oracle, pane=None):
    """camera at the origin looking down +z at a mirror in the plane z = 4; a one-sided emitter behind the camera (z = -2)
    faces +z.  Optionally a glass pane (two specular lobes, allow_multiple_lobes = false) at z = 2."""
    sb = scenes.SceneBuilder()
    mir = sb.add_material(scenes.mirror((1.0, 1.0, 1.0)))
    dark = sb.add_material(scenes.matte((0.0, 0.0, 0.0)))
    sb.add_quad([(-50, -50, 4), (-50, 50, 4), (50, 50, 4), (50, -50, 4)], mir)  # normal towards -z
    sb.add_quad([(-50, -50, -2), (50, -50, -2), (50, 50, -2), (-50, 50, -2)], dark, emit=(3.0, 2.0, 1.0))  # normal towards +z
    if pane is not None:
        g = sb.add_material(scenes.glass(index=pane, multiple_lobes=False))
        sb.add_quad([(-50, -50, 2), (-50, 50, 2), (50, 50, 2), (50, -50, 2)], g)
    return sb.finish(oracle.bvh_build)


@pytest.mark.parametrize("kind", ["direct", "whitted"])
def test_mirror_shows_the_emitter_behind_the_camera(oracle, kind):
    sc = _mirror_scene(oracle)
    look = ((0, 0, 0), (0, 0, 1), (0, 1, 0))
    for depth, expect in ((1, 0.0), (2, 1.0), (5, 1.0)):
        rd = scenes.make_render_desc(8, 8, 4, look, 20.0, max_depth=depth)
        r = oracle.render_integrator(sc, rd, kind, strategy="one", threads=2, want_li=True)
        assert np.allclose(r["li"], np.array([3.0, 2.0, 1.0]) * expect, rtol=1e-6, atol=0)


@pytest.mark.parametrize("kind", ["direct", "whitted"])
def test_glass_pane_transmits_twice(oracle, kind):
    """camera -> pane (T) -> mirror -> pane (T) -> emitter: (1 - F)^2 L with F = ((n - 1) / (n + 1))^2 at normal incidence,
    radiance scaling 1/eta^2 in and eta^2 out cancelling; the pane's own reflection shows the emitter too: + F L.
    Depth: pane 0, mirror 1, pane 2, emitter hit at depth 3 -> max_depth >= 4 for the transmitted term."""
    n = 1.5
    sc = _mirror_scene(oracle, pane=n)
    look = ((0, 0, 0), (0, 0, 1), (0, 1, 0))
    F = ((n - 1) / (n + 1)) ** 2
    L = np.array([3.0, 2.0, 1.0])
    rd = scenes.make_render_desc(4, 4, 2, look, 2.0, max_depth=2)  # narrow view: normal incidence to ~1e-4
    r = oracle.render_integrator(sc, rd, kind, strategy="one", threads=2, want_li=True)
    assert np.allclose(r["li"], F * L, rtol=2e-3)  # only the pane's reflection
    rd = scenes.make_render_desc(4, 4, 2, look, 2.0, max_depth=4)
    r = oracle.render_integrator(sc, rd, kind, strategy="one", threads=2, want_li=True)
    assert np.allclose(r["li"], (F + (1 - F) ** 2) * L, rtol=2e-3)
    rd = scenes.make_render_desc(4, 4, 2, look, 2.0, max_depth=6)  # + the path that bounces once more between pane and mirror
    r = oracle.render_integrator(sc, rd, kind, strategy="one", threads=2, want_li=True)
    assert np.allclose(r["li"], (F + (1 - F) ** 2 * (1 + F)) * L, rtol=2e-3)
