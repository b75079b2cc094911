"""The CONTROL FLOW of the path integrator held to the REFERENCE'S OWN TEXT (round 6, third session).

oracle/make_flow_fixtures.py compiles `PathIntegrator::li` (integrators/path.rs:59-282) and `uniform_sample_one_light` (core/integrator.rs:359-403) from the Rust text —
syntax rewritten by committed rules, the subsurface block dropped by rule — over carriers that hand the reference's method names to the oracle's leaf functions, and
renders through the oracle's tile loop with that li.  The oracle's own restatement of li (oracle/orc_render.hpp path_li, which every GPU test is held to sample for
sample) must give the same radiance for EVERY camera sample, bit for bit: which terms a path adds in which order, when it stops (max_depth, a black f, a zero pdf),
what it draws from the sampler and when, the null-material `continue` that skips `bounces += 1`, the eta_scale of refraction, Russian roulette from bounce 4 on.
Needs /root/reference (the text is compiled here; nothing of it is committed): skipped elsewhere — the GPU box holds the HIP path to the oracle, this file holds the
oracle to the reference."""
import os
import sys

import numpy as np
import pytest

from rs_pbrt_amd import abi, scenes

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HAVE_REF = os.path.exists("/root/reference/src/integrators/path.rs")
pytestmark = pytest.mark.skipif(not HAVE_REF, reason="needs /root/reference to compile the reference's text")


@pytest.fixture(scope="module")
def flow():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import make_flow_fixtures as mk
    L, where = mk.convert()
    for w in ("estimate_direct core/integrator.rs:406-570", "uniform_sample_one_light core/integrator.rs:359-403", "PathIntegrator::li integrators/path.rs:59-282", "Bsdf::sample_f core/reflection.rs:298-420",
              "uniform_sample_all_lights core/integrator.rs:300-355", "DirectLightingIntegrator::li integrators/directlighting.rs:71-123", "recursive_build accelerators/bvh.rs:178-357"):
        assert w in where, (w, [x for x in where if w.split(" ")[0] in x])
    return mk, L


def both(flow, sc, rd):
    mk, L = flow
    film_t, li_t = mk.render(L, sc, rd, True)
    film_o, li_o = mk.render(L, sc, rd, False)
    return film_t, li_t, film_o, li_o


def assert_same(li_t, li_o):
    a, b = li_t.view(np.uint32), li_o.view(np.uint32)
    nan = np.isnan(li_t) & np.isnan(li_o)
    assert np.all((a == b) | nan), "%d of %d camera samples differ" % (int(((a != b) & ~nan).any(axis=-1).sum()), li_t.shape[0] * li_t.shape[1])


def test_li_text_equals_the_oracles_li_on_the_cornell_box_past_the_roulette_threshold(flow, oracle):
    sc = scenes.cornell_box(oracle.bvh_build)
    rd = scenes.cornell_render_desc(res=48, spp=16, max_depth=12)          # roulette from bounce 4 on (rr_threshold 1)
    film_t, li_t, film_o, li_o = both(flow, sc, rd)
    assert_same(li_t, li_o)
    assert np.array_equal(film_t, film_o) and li_t.mean() > 0.05
    ref = oracle.render(sc, rd, threads=4, want_li=True)                   # and the library every other test uses gives the same samples
    assert np.array_equal(ref["li"].reshape(li_o.shape).view(np.uint32), li_o.view(np.uint32))


@pytest.mark.parametrize("lights", ["all", "delta", "area"])
def test_li_text_equals_the_oracles_li_on_the_gallery(flow, oracle, lights):
    """every material recipe (specular and rough transmission: eta_scale; mixes), area + point + spot + distant lights, all three light strategies"""
    from tests.util import GALLERY_LOOK_AT, gallery
    sc = gallery(oracle.bvh_build, lights)
    for strategy in (abi.LIGHTS_SPATIAL, abi.LIGHTS_POWER, abi.LIGHTS_UNIFORM):
        rd = scenes.make_render_desc(64, 48, 8, GALLERY_LOOK_AT, 60, max_depth=7, light_strategy=strategy)
        _, li_t, _, li_o = both(flow, sc, rd)
        assert_same(li_t, li_o)


def test_li_text_equals_the_oracles_li_with_null_surfaces_and_an_infinite_light(flow, oracle):
    from tests.util import sky_scene
    cb = scenes.cornell_box(oracle.bvh_build)
    cb.prims["material"][cb.prims["material"] == 1] = abi.NO_MATERIAL     # a null boundary: passes do not count as bounces (path.rs:109-116)
    rd = scenes.cornell_render_desc(res=40, spp=8, max_depth=3)
    _, li_t, _, li_o = both(flow, cb, rd)
    assert_same(li_t, li_o)
    for kind in ("constant", "image"):
        sc = sky_scene(oracle.bvh_build, kind, with_area=True)             # escaped rays add the environment on bounce 0 / after a specular bounce only
        from tests.util import GALLERY_LOOK_AT
        rd = scenes.make_render_desc(48, 36, 8, GALLERY_LOOK_AT, 60, max_depth=5)
        _, li_t, _, li_o = both(flow, sc, rd)
        assert_same(li_t, li_o)


@pytest.mark.parametrize("seed", list(range(101, 117)))
def test_li_text_equals_the_oracles_li_on_random_scenes(flow, oracle, seed):
    from tests.util import GALLERY_LOOK_AT, random_scene
    sc = random_scene(oracle.bvh_build, seed)
    rd = scenes.make_render_desc(48, 36, 8, GALLERY_LOOK_AT, 55, max_depth=2 + seed % 9, sampler="halton" if seed % 2 else "sobol",
                                 light_strategy=[abi.LIGHTS_SPATIAL, abi.LIGHTS_POWER, abi.LIGHTS_UNIFORM][seed % 3])
    _, li_t, _, li_o = both(flow, sc, rd)
    assert_same(li_t, li_o)


def test_every_lobes_f_pdf_and_sample_f_text_equals_the_oracles(flow):
    """LambertianReflection / Transmission, OrenNayar, SpecularReflection / Transmission, FresnelSpecular, MicrofacetReflection / Transmission and FresnelBlend (reflection.rs:711-1478:
    f, pdf, sample_f, get_type of each, over Fresnel::evaluate, TrowbridgeReitzDistribution::sample_wh / d / g / pdf, reflect, refract, cosine_sample_hemisphere — all
    compiled from the reference's text) against the oracle's lobe evaluator on the same records: 2^17 cases, every bit.  Includes the reference's own quirk that a
    MixMaterial scale enters sample_f's value twice where a lobe's sample_f calls its f."""
    mk, L = flow
    n = 1 << 17
    b, wo, wi, u = mk.lobe_cases(n, 0x10BE5)
    t, o = mk.run_lobes(L, b, wo, wi, u)
    names = ["f.r", "f.g", "f.b", "pdf", "sample_f.r", "sample_f.g", "sample_f.b", "wi.x", "wi.y", "wi.z", "sample pdf", "sampled_type", "get_type"]
    bad = (t.view(np.uint32) != o.view(np.uint32)) & ~(np.isnan(t) & np.isnan(o))
    for kind in np.unique(b["type"]):
        sel = b["type"] == kind
        assert not bad[sel].any(), "lobe kind %d: %s differ in %d of %d cases" % (kind, [names[c] for c in np.nonzero(bad[sel].any(axis=0))[0]], int(bad[sel].any(axis=1).sum()), int(sel.sum()))
        assert (t[sel, 12] != 0).all() and (t[sel, 10] > 0).mean() > 0.25
    r = __import__("subprocess").run([sys.executable, os.path.join(ROOT, "oracle", "make_flow_fixtures.py"), "--check"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]          # tests/golden/lobe_functions.npz (what the GPU test holds the device's lobes to) is what the text gives         # every kind was evaluated and mostly sampled with a positive pdf


@pytest.mark.parametrize("shape,kind", [((7, 1), "lights"), ((64, 32), "image"), ((5, 3), "zeros"), ((1, 1), "single")])
def test_distribution_text_equals_the_oracles(flow, shape, kind):
    """Distribution1D::new / sample_discrete / sample_continuous / discrete_pdf (sampling.rs:24-147: the table behind every light choice) and Distribution2D's
    sample_continuous / pdf (the environment light's image, sampling.rs:172-198) from the reference's text against the oracle's: cdf, integral, and 2^15 samples each —
    over a handful of light powers, a 64 x 32 image with black rows, an all-zero function (the uniform fall-back), a single entry"""
    import ctypes as C
    mk, L = flow
    rng = np.random.default_rng(hash(kind) & 0xffff)
    nu, nv = shape
    f = np.exp(rng.uniform(-3, 3, (nv, nu))).astype(np.float32)
    if kind == "image":
        f[5] = 0.0; f[:, 7] = 0.0
    if kind == "zeros":
        f[:] = 0.0
    n = 1 << 15
    u = rng.uniform(0, 1, (n, 2)).astype(np.float32).clip(0, np.nextafter(np.float32(1), np.float32(0)))
    u[:4] = [[0, 0], [0.99999994, 0.99999994], [0.5, 0.5], [0, 0.99999994]]
    ht, ho = np.zeros(nu + 2, np.float32), np.zeros(nu + 2, np.float32)
    t, o = np.zeros((n, 10), np.float32), np.zeros((n, 10), np.float32)
    L.flow_distributions.restype = None
    L.flow_distributions.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64] + [C.c_void_p] * 4
    L.flow_distributions(f.ctypes.data, nu, nv, u.ctypes.data, n, ht.ctypes.data, ho.ctypes.data, t.ctypes.data, o.ctypes.data)
    assert np.array_equal(ht.view(np.uint32), ho.view(np.uint32)), "cdf / integral differ"
    bad = (t.view(np.uint32) != o.view(np.uint32)) & ~(np.isnan(t) & np.isnan(o))
    assert not bad.any(), "columns %s differ in %d samples" % (np.nonzero(bad.any(axis=0))[0], int(bad.any(axis=1).sum()))


@pytest.mark.parametrize("lens", [0.0, 0.05])
def test_camera_text_equals_the_oracles_camera_ray(flow, lens):
    """PerspectiveCamera::generate_ray_differential (perspective.rs:190-280) over Transform::{transform_point, transform_vector, transform_point_with_error, transform_ray}
    (transform.rs:490-595, 662-708), Ray::position and lerp — the reference's text — against the oracle's camera_ray: origin, direction, t_max, time and the two offset rays
    of 2^15 camera samples per camera (pinhole and thin lens; a camera that does not move), bit for bit"""
    import ctypes as C
    mk, L = flow
    rng = np.random.default_rng(77 + int(lens * 100))
    L.flow_camera.restype = None
    L.flow_camera.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    n = 1 << 15
    for k in range(6):
        eye = rng.uniform(-20, 20, 3); look = eye + rng.normal(size=3) * 5; up = [(0, 1, 0), (0, 0, 1), (1, 0, 0)][k % 3]
        res = [(400, 400), (1920, 1080), (64, 48)][k % 3]
        rd = scenes.make_render_desc(res[0], res[1], 4, (tuple(eye), tuple(look), up), float(rng.uniform(20, 90)), lens_radius=lens, focal_distance=float(rng.uniform(1, 20)),
                                     shutter=(0.0, 1.0) if k % 2 else (0.25, 0.75))
        smp = np.zeros((n, 5), np.float32)
        smp[:, 0] = rng.uniform(0, res[0], n); smp[:, 1] = rng.uniform(0, res[1], n); smp[:, 2:] = rng.uniform(0, 1, (n, 3))
        smp[:4, :2] = [[0, 0], [res[0], res[1]], [res[0] / 2, res[1] / 2], [0.5, 0.5]]; smp[:8, 3:] = 0.5
        t, o = np.zeros((n, 20), np.float32), np.zeros((n, 20), np.float32)
        L.flow_camera(C.addressof(rd), smp.ctypes.data, n, t.ctypes.data, o.ctypes.data)
        bad = (t.view(np.uint32) != o.view(np.uint32)) & ~(np.isnan(t) & np.isnan(o))
        assert not bad.any(), "camera %d: columns %s differ in %d samples" % (k, np.nonzero(bad.any(axis=0))[0], int(bad.any(axis=1).sum()))
        assert np.abs(np.linalg.norm(t[:, 3:6], axis=1) - 1).max() < 1e-5 and (t[:, 6] > 1e30).all()


def test_delta_light_text_equals_the_oracles_light_sample_li(flow, oracle):
    """PointLight / SpotLight (with falloff through world_to_light) / DistantLight::sample_li (lights/point.rs, spot.rs, distant.rs) from the reference's text against the
    oracle's light_sample_li: wi, pdf, radiance, the light-side point and its time for 2^15 reference points per light kind"""
    import ctypes as C
    from tests.util import gallery
    mk, L = flow
    sc = gallery(oracle.bvh_build, "delta")
    lights = sc.lights[np.isin(sc.lights["kind"], [abi.LIGHT_POINT, abi.LIGHT_SPOT, abi.LIGHT_DISTANT])]
    assert set(lights["kind"]) == {abi.LIGHT_POINT, abi.LIGHT_SPOT, abi.LIGHT_DISTANT}
    rng = np.random.default_rng(5)
    n = 1 << 15
    lt = np.ascontiguousarray(lights[rng.integers(0, len(lights), n)])
    ref = rng.uniform(-6, 6, (n, 3)).astype(np.float32)
    t, o = np.zeros((n, 11), np.float32), np.zeros((n, 11), np.float32)
    L.flow_delta_lights.restype = None
    L.flow_delta_lights.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    L.flow_delta_lights(C.addressof(sc.desc), lt.ctypes.data, ref.ctypes.data, n, t.ctypes.data, o.ctypes.data)
    bad = (t.view(np.uint32) != o.view(np.uint32)) & ~(np.isnan(t) & np.isnan(o))
    for kind in (abi.LIGHT_POINT, abi.LIGHT_SPOT, abi.LIGHT_DISTANT):
        sel = lt["kind"] == kind
        assert not bad[sel].any(), "light kind %d: columns %s differ in %d cases" % (kind, np.nonzero(bad[sel].any(axis=0))[0], int(bad[sel].any(axis=1).sum()))
    spot = lt["kind"] == abi.LIGHT_SPOT
    assert (t[spot, 4:7] == 0).all(axis=1).any() and (t[spot, 4:7] > 0).all(axis=1).any()     # outside and inside the cone


@pytest.mark.parametrize("lights", ["all", "area"])
def test_spatial_light_distribution_text_equals_the_oracles(flow, oracle, lights):
    """SpatialLightDistribution::compute_distribution (lightdistrib.rs:180-275: the voxel's bounds, 128 Halton points, every light sampled at each, the floor of a
    thousandth of the average, Distribution1D::new) from the reference's text against the oracle's spatial_compute: the weights of every light in 64 voxels"""
    import ctypes as C
    from tests.util import GALLERY_LOOK_AT, gallery
    mk, L = flow
    sc = gallery(oracle.bvh_build, lights)
    rd = scenes.make_render_desc(32, 24, 1, GALLERY_LOOK_AT, 60, light_strategy=abi.LIGHTS_SPATIAL)
    rng = np.random.default_rng(3)
    pi = rng.integers(0, 6, (64, 3)).astype(np.int32)
    pi[:2] = [[0, 0, 0], [1, 0, 2]]
    nl = int(sc.desc.n_lights)
    t, o = np.zeros((64, nl), np.float32), np.zeros((64, nl), np.float32)
    L.flow_spatial.restype = C.c_int
    L.flow_spatial.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    assert L.flow_spatial(C.addressof(sc.desc), C.addressof(rd), pi.ctypes.data, 64, t.ctypes.data, o.ctypes.data) == nl
    assert np.array_equal(t.view(np.uint32), o.view(np.uint32)), "%d of %d weights differ" % (int((t.view(np.uint32) != o.view(np.uint32)).sum()), t.size)
    assert (t > 0).all() and len(np.unique(t)) > nl


@pytest.mark.parametrize("kind", ["soup", "dense", "coincident", "grid", "tiny"])
def test_bvh_builder_text_equals_the_oracles_tree(flow, oracle, kind):
    """BVHAccel::recursive_build (SAH with 12 buckets, the two-primitive case, coincident centroids, the leaf rule, the stable partition; the RIGHT child is built first) and
    flatten_bvh_tree (bvh.rs:171-392) with BVHPrimitiveInfo::new, BVHBuildNode::init_leaf / init_interior, Bounds3f::{default, diagonal, surface_area, maximum_extent, offset},
    bnd3_union_* from the reference's text: the flattened nodes (bounds, offsets, counts, axes) and the primitive order equal the oracle's builder byte for byte"""
    import ctypes as C
    mk, L = flow
    rng = np.random.default_rng({"soup": 1, "dense": 2, "coincident": 3, "grid": 4, "tiny": 5}[kind])
    n = {"soup": 50000, "dense": 20000, "coincident": 3000, "grid": 4096, "tiny": 3}[kind]
    c = rng.uniform(-1, 1, (n, 3)); h = rng.uniform(0, 0.01 if kind != "dense" else 0.3, (n, 3))
    if kind == "coincident":
        c[: n // 2] = c[0]; c[n // 2:, 1:] = 0.25                     # half of the centroids in one point, the rest on a line
    if kind == "grid":
        g = np.stack(np.meshgrid(*[np.arange(16)] * 3, indexing="ij"), -1).reshape(-1, 3); c = g / 16.0; h[:] = 0.03     # equal costs: ties in the bucket search
    b6 = np.concatenate([c - h, c + h], 1).astype(np.float32)
    L.flow_bvh_build.restype = C.c_int64
    L.flow_bvh_build.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p]
    for max_prims in (1, 4, 255):
        nodes_t = np.zeros(2 * n, abi.NODE_DT); order_t = np.zeros(n, np.uint32)
        nn = L.flow_bvh_build(b6.ctypes.data, n, max_prims, nodes_t.ctypes.data, len(nodes_t), order_t.ctypes.data)
        nodes_o, order_o = oracle.bvh_build_bounds(b6, max_prims)
        assert nn == len(nodes_o) > 0
        assert nodes_t[:nn].tobytes() == nodes_o.tobytes(), "%s, max_prims %d: %d of %d nodes differ" % (kind, max_prims, int((nodes_t[:nn] != nodes_o).sum()), nn)
        assert np.array_equal(order_t, order_o) and sorted(order_t.tolist()) == list(range(n))


def render_direct(flow, sc, rd, strategy, use_text, n_light_samples=None):
    import ctypes as C
    mk, L = flow
    cw, ch = rd.crop_px[2] - rd.crop_px[0], rd.crop_px[3] - rd.crop_px[1]
    film = np.zeros((cw * ch, 4), np.float32)
    li = np.zeros((cw * ch, int(rd.spp), 3), np.float32)
    L.flow_render_direct.restype = C.c_int
    L.flow_render_direct.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    ns = None if n_light_samples is None else np.ascontiguousarray(n_light_samples, np.int32)
    assert L.flow_render_direct(C.addressof(sc.desc), C.addressof(rd), 4, film.ctypes.data, li.ctypes.data, strategy, None if ns is None else ns.ctypes.data, int(use_text)) == 0
    return li


@pytest.mark.parametrize("strategy", [0, 1])
def test_directlighting_li_text_equals_the_oracles(flow, oracle, strategy):
    """DirectLightingIntegrator::li with specular_reflect / specular_transmit (the reflected and refracted rays' differentials; directlighting.rs:71-258) and
    uniform_sample_all_lights (the per-light 2-D sample arrays; integrator.rs:299-354) / uniform_sample_one_light without a distribution, compiled from the reference's
    text, against the oracle's recursive_li for every camera sample: the gallery (mirror, glass, every light kind), several samples per light, both samplers, a
    null-material wall, an infinite light"""
    from tests.util import GALLERY_LOOK_AT, gallery, sky_scene
    sc = gallery(oracle.bvh_build, "all")
    nl = int(sc.desc.n_lights)
    for sampler, samples in (("sobol", None), ("halton", [1 + (j % 3) * 2 for j in range(nl)]), ("sobol", [4] * nl)):
        rd = scenes.make_render_desc(48, 36, 4, GALLERY_LOOK_AT, 60, max_depth=5, sampler=sampler, integrator="directlighting", direct_strategy="all" if strategy == 0 else "one", light_samples=samples)
        t = render_direct(flow, sc, rd, strategy, True, samples); o = render_direct(flow, sc, rd, strategy, False, samples)
        assert_same(t, o)
        assert t.mean() > 0.01
    cb = scenes.cornell_box(oracle.bvh_build)
    cb.prims["material"][cb.prims["material"] == 1] = abi.NO_MATERIAL
    rd = scenes.cornell_render_desc(res=32, spp=4, integrator="directlighting", direct_strategy="all" if strategy == 0 else "one")
    assert_same(render_direct(flow, cb, rd, strategy, True), render_direct(flow, cb, rd, strategy, False))
    sky = sky_scene(oracle.bvh_build, "image", with_area=True)
    rd = scenes.make_render_desc(40, 30, 4, GALLERY_LOOK_AT, 60, max_depth=4, integrator="directlighting", direct_strategy="all" if strategy == 0 else "one")
    assert_same(render_direct(flow, sky, rd, strategy, True), render_direct(flow, sky, rd, strategy, False))


@pytest.mark.parametrize("cos_sample,sampler,n", [(True, "sobol", 16), (False, "sobol", 8), (True, "halton", 5)])
def test_ao_li_text_equals_the_oracles(flow, oracle, cos_sample, sampler, n):
    """AOIntegrator::li (ao.rs:53-110: the frame from the true geometry, this pixel sample's slice of the 2-D array, cosine / uniform hemisphere directions, the unoccluded
    terms summed in array order) from the reference's text against the oracle's ao_li, every camera sample"""
    sc = scenes.cornell_box(oracle.bvh_build)
    rd = scenes.cornell_render_desc(res=40, spp=4, integrator="ao", ao_samples=n, ao_cos_sample=cos_sample, sampler=sampler)
    _, li_t, _, li_o = both(flow, sc, rd)
    assert_same(li_t, li_o)
    assert 0.3 < li_t.mean() < 3.2


def test_material_recipes_text_equals_the_oracles_lobe_lists(flow, oracle):
    """MatteMaterial / PlasticMaterial / MirrorMaterial / GlassMaterial / MetalMaterial / SubstrateMaterial / UberMaterial / TranslucentMaterial::compute_scattering_functions (materials/*.rs) over Bsdf::new / add, OrenNayar::new,
    SpecularTransmission::new, MicrofacetTransmission::new, TrowbridgeReitzDistribution::new / roughness_to_alpha and Spectrum::clamp — the reference's text, with constant
    parameters — against the oracle's material assembly (which the library's rspt_material_lobes is held to): the same lobes in the same order with the same parameters,
    with both values of allow_multiple_lobes and remaproughness"""
    import ctypes as C
    mk, L = flow
    rng = np.random.default_rng(17)
    L.flow_material.restype = C.c_int
    L.flow_material.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    frame = np.array([0, 0, 1, 0, 0, 1, 1, 0, 0], np.float32); sc3 = np.zeros(3, np.float32)
    fields = ["type", "fresnel", "r", "t", "eta_a", "eta_b", "alpha_x", "alpha_y", "c1", "c2", "on_a", "on_b", "has_sc"]
    n_checked = 0
    for trial in range(400):
        kind = trial % 8
        remap = bool(rng.integers(0, 2)); allow = bool(rng.integers(0, 2))
        c = lambda: tuple(float(np.float32(x)) for x in rng.uniform(0, 1, 3) * (rng.uniform(size=3) > 0.15))      # noqa: E731  (black channels / black colours drop lobes)
        f = lambda lo, hi: float(np.float32(rng.uniform(lo, hi)))                                                  # noqa: E731
        p = np.zeros(24, np.float32)
        if kind == 0:
            kd, sigma = c(), (0.0 if trial % 10 < 5 else f(0, 40)); m = scenes.matte(kd, sigma); p[:3] = kd; p[3] = sigma
        elif kind == 1:
            kd, ks, ro = c(), c(), f(0.001, 1.0); m = scenes.plastic(kd, ks, ro, remap); p[:3] = kd; p[3:6] = ks; p[6] = ro
        elif kind == 2:
            kr = c(); m = scenes.mirror(kr); p[:3] = kr
        elif kind == 3:
            kr, kt, idx = c(), c(), f(1.1, 2.5); ur, vr = (0.0, 0.0) if trial % 10 < 5 else (f(0.01, 0.8), f(0.01, 0.8))
            m = scenes.glass(kr, kt, idx, ur, vr, remap); p[:3] = kr; p[3:6] = kt; p[6] = ur; p[7] = vr; p[8] = idx
        elif kind == 5:
            kd, ks, nu, nv = c(), c(), f(0.001, 1.0), f(0.001, 1.0); m = scenes.substrate(kd, ks, nu, nv, remap); p[:3] = kd; p[3:6] = ks; p[6] = nu; p[7] = nv
        elif kind == 6:
            kd, ks, kr, kt, ro, idx = c(), c(), c(), c(), f(0.001, 1.0), f(1.1, 2.5)
            op = (1.0, 1.0, 1.0) if trial % 16 < 8 else tuple(float(np.float32(x)) for x in rng.uniform(0, 1, 3))
            uv = (None, None) if trial % 3 else (f(0.01, 0.5), f(0.01, 0.5))
            m = scenes.uber(kd, ks, kr, kt, ro, uv[0], uv[1], op, idx, remap)
            p[:3] = kd; p[3:6] = ks; p[6:9] = kr; p[9:12] = kt; p[12:15] = op; p[15] = ro; p[16] = -1 if uv[0] is None else uv[0]; p[17] = -1 if uv[1] is None else uv[1]; p[18] = idx
        elif kind == 7:
            kd, ks, refl, tran, ro = c(), c(), c(), c(), f(0.001, 1.0); m = scenes.translucent(kd, ks, refl, tran, ro, remap); p[:3] = kd; p[3:6] = ks; p[6] = ro; p[7:10] = refl; p[10:13] = tran
        else:
            eta, k, ro = tuple(f(0.1, 3) for _ in range(3)), tuple(f(0, 6) for _ in range(3)), f(0.001, 0.8)
            uv = (None, None) if trial % 10 < 5 else (f(0.01, 0.5), f(0.01, 0.5))
            m = scenes.metal(eta, k, ro, remap, uv[0], uv[1]); p[:3] = eta; p[3:6] = k; p[6] = ro; p[7] = -1 if uv[0] is None else uv[0]; p[8] = -1 if uv[1] is None else uv[1]
        sc, mi = scenes.material_scene(m)
        eta_o, lob_o = oracle.material_lobes(sc, mi, allow)
        out = np.zeros(8, abi.BXDF_DT); eta_t = C.c_float(0)
        n = L.flow_material(kind, p.ctypes.data, (1 if remap else 0) | (2 if allow else 0), sc3.ctypes.data, frame.ctypes.data, out.ctypes.data, C.addressof(eta_t))
        assert n == len(lob_o), (kind, trial, n, len(lob_o))
        assert np.float32(eta_t.value) == np.float32(eta_o)
        for fld in fields:
            assert np.array_equal(np.ascontiguousarray(out[:n][fld]).view(np.uint32), np.ascontiguousarray(lob_o[fld]).view(np.uint32)), (kind, trial, fld, out[:n][fld], lob_o[fld])
        n_checked += n
    assert n_checked > 500


# ---- SamplerIntegrator::render: the tile loop's own text over the text of every stage it calls ----
@pytest.mark.parametrize("sampler", ["sobol", "halton", "02sequence", "maxmindist", "stratified", "random"])
def test_render_with_every_stage_from_the_references_text_equals_the_oracles_film(flow, oracle, sampler):
    """the worker closure of SamplerIntegrator::render (integrator.rs:109-200: tile bounds, the per-tile seed, start_pixel, get_camera_sample, generate_ray_differential,
    scale_differentials, li, the NaN test, add_sample, start_next_sample) compiled from the text and run over the TEXT of the sampler, the camera, PathIntegrator::li and the film,
    tiles in BlockQueue's Morton order, merged by the text's Film::merge_film_tile: the oracle's render() must produce the same Film.pixels, bit for bit.
    56 x 40 pixels = 3.5 x 2.5 tiles (partial tiles on two sides; the filter reaches across tile borders)."""
    mk, L = flow
    assert "SamplerIntegrator::render (tile loop) core/integrator.rs:109-200" in mk.convert_parts()[1]
    sc = scenes.cornell_box(oracle.bvh_build, variant="mixed")
    rd = scenes.make_render_desc(56, 40, 16 if sampler != "halton" else 5, scenes.CORNELL_LOOK_AT, scenes.CORNELL_FOV, max_depth=6, sampler=sampler, strat=(4, 4), filter_radius=(1.5, 1.5))
    film_t = mk.render_tiles(L, sc, rd)
    ref = oracle.render(sc, rd, threads=4)
    a, b = film_t.view(np.uint32), np.ascontiguousarray(ref["film"], np.float32).view(np.uint32)
    assert np.array_equal(a, b), "%d of %d film words differ" % (int((a != b).sum()), a.size)
    assert film_t[:, 3].min() > 0 and film_t[:, 1].mean() > 0.05


def test_render_text_on_the_gallery_with_cropped_sample_bounds(flow, oracle):
    from tests.util import GALLERY_LOOK_AT, gallery
    mk, L = flow
    sc = gallery(oracle.bvh_build, "all")
    rd = scenes.make_render_desc(64, 48, 4, GALLERY_LOOK_AT, 60, max_depth=5, crop=(0.25, 0.9, 0.1, 0.8))
    film_t = mk.render_tiles(L, sc, rd)
    ref = oracle.render(sc, rd, threads=4)
    assert np.array_equal(film_t.view(np.uint32), np.ascontiguousarray(ref["film"], np.float32).view(np.uint32)) and (np.array(rd.sample_bounds[:2]) > 0).all()


@pytest.mark.parametrize("kind", ["constant", "image"])
def test_infinite_light_text_equals_the_oracle(flow, oracle, kind):
    """InfiniteAreaLight::{sample_li, le, pdf_li} (lights/infinite.rs:298-392) over MipMap::{lookup_pnt_flt, triangle, texel} (core/mipmap.rs, the Repeat arm), spherical_theta / _phi and the text's
    Distribution2D, against the oracle's light_sample_li / infinite_le / infinite_pdf_li on the sky scenes (a constant map = one texel; an image map with its pyramid and a rotated light frame)"""
    import ctypes as C
    from tests.util import sky_scene
    mk, L = flow
    sc = sky_scene(oracle.bvh_build, kind, with_area=False)
    rng = np.random.default_rng(23)
    n = 1 << 15
    ref = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    u = rng.uniform(0, 1, (n, 2)).astype(np.float32).clip(0, np.nextafter(np.float32(1), np.float32(0))); u[:64, 1] = 0.0; u[64:128, 0] = 0.0
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1)[:, None]; d = d.astype(np.float32); d[:32] = [0, 0, 1]; d[32:64] = [0, 0, -1]; d[64:96] = [1, 0, 0]      # the poles (sin theta = 0), the seam
    t, q = np.zeros((n, 17), np.float32), np.zeros((n, 17), np.float32)
    L.flow_infinite.restype = C.c_int
    L.flow_infinite.argtypes = [C.c_void_p] * 4 + [C.c_uint64, C.c_void_p, C.c_void_p]
    levels = L.flow_infinite(C.addressof(sc.desc), ref.ctypes.data, u.ctypes.data, d.ctypes.data, n, t.ctypes.data, q.ctypes.data)
    assert levels >= 1 and (kind == "constant" or levels > 3)
    bad = (t.view(np.uint32) != q.view(np.uint32)) & ~(np.isnan(t) & np.isnan(q))
    assert not bad.any(), "%d of %d values differ (columns %s)" % (int(bad.sum()), bad.size, sorted(set(np.where(bad)[1].tolist())))
    assert (t[:, 0] > 0).mean() > 0.9 and (t[:, 13] > 0).mean() > 0.9 and t[:, 10:13].max() > 0


@pytest.mark.parametrize("wrap", [abi.WRAP_REPEAT, abi.WRAP_BLACK, abi.WRAP_CLAMP])
@pytest.mark.parametrize("trilinear", [0, 1])
def test_mipmap_lookup_text_equals_the_oracle(flow, oracle, wrap, trilinear):
    """MipMap::lookup_pnt_vec_vec (mipmap.rs:253-297) — the trilinear width or the EWA path with the eccentricity clamp, the level of detail, the two-level blend — over ewa (:337-400, with the
    weight table of MipMap::new :186-193), triangle, lookup_pnt_flt and texel in all three wrap modes, against the oracle's img_lookup (which the device's texture fetch is held to): footprints from a
    fraction of a texel to tens of texels, degenerate (zero) axes, lookups across the image border"""
    import ctypes as C
    mk, L = flow
    rng = np.random.default_rng(31 + wrap * 2 + trilinear)
    w, h = 32, 16
    levels = []
    lw, lh = w, h
    base = rng.uniform(0, 2, (h, w, 3)).astype(np.float32)
    while True:
        levels.append(rng.uniform(0, 2, (lh, lw, 3)).astype(np.float32) if levels else base)      # (any numbers: the lookups' arithmetic is what is compared)
        if lw == 1 and lh == 1:
            break
        lw, lh = max(1, lw // 2), max(1, lh // 2)
    tex = np.concatenate([l.reshape(-1) for l in levels])
    img = abi.Image(); img.width, img.height, img.n_levels, img.channels = w, h, len(levels), 3; img.texels = tex.ctypes.data
    tx = abi.Texture(); tx.wrap, tx.trilinear, tx.max_aniso = wrap, trilinear, 8.0
    n = 1 << 14
    st = rng.uniform(-0.5, 1.5, (n, 2)).astype(np.float32)
    scale = np.exp(rng.uniform(np.log(1e-4), np.log(0.2), (n, 1)))
    d0 = (rng.normal(size=(n, 2)) * scale).astype(np.float32); d1 = (rng.normal(size=(n, 2)) * scale * np.exp(rng.uniform(-3, 0, (n, 1)))).astype(np.float32)
    d1[:64] = 0.0; d0[64:96] = 0.0; d1[64:96] = 0.0
    t, q = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    L.flow_mipmap.restype = None
    L.flow_mipmap.argtypes = [C.c_void_p] * 5 + [C.c_uint64, C.c_void_p, C.c_void_p]
    L.flow_mipmap(C.addressof(img), C.addressof(tx), st.ctypes.data, d0.ctypes.data, d1.ctypes.data, n, t.ctypes.data, q.ctypes.data)
    bad = (t.view(np.uint32) != q.view(np.uint32)) & ~(np.isnan(t) & np.isnan(q))
    assert not bad.any(), "%d of %d lookups differ" % (int(bad.any(axis=1).sum()), n)
    assert np.isfinite(t).mean() > 0.99 and t.std() > 0.1


def test_noise_fbm_turbulence_text_equals_the_oracle(flow, oracle):
    """noise_flt / noise_pnt3 / grad / noise_weight over the permutation table NOISE_PERM, fbm, turbulence (the octave count from the footprint through log_2), smooth_step and lanczos
    (core/texture.rs:21-48, 289-439; pbrt.rs:153-156) against the oracle's (the marble / wrinkled / windy / fbm textures sit on them): points over many noise cells incl. negative
    coordinates and exact lattice points, footprints from sub-octave to none"""
    import ctypes as C
    mk, L = flow
    rng = np.random.default_rng(41)
    n = 1 << 16
    p = (rng.uniform(-300, 300, (n, 3)) * np.exp(rng.uniform(-4, 0, (n, 1)))).astype(np.float32); p[:256] = np.round(p[:256])
    dx = (rng.normal(size=(n, 3)) * np.exp(rng.uniform(-9, 1, (n, 1)))).astype(np.float32); dy = (rng.normal(size=(n, 3)) * np.exp(rng.uniform(-9, 1, (n, 1)))).astype(np.float32)
    dx[:32] = 0; dy[:32] = 0
    par = np.stack([rng.uniform(0.2, 0.9, n), rng.integers(1, 9, n), rng.uniform(-0.5, 1.5, n)], 1).astype(np.float32)
    t, q = np.zeros((n, 5), np.float32), np.zeros((n, 5), np.float32)
    L.flow_noise.restype = None
    L.flow_noise.argtypes = [C.c_void_p] * 4 + [C.c_uint64, C.c_void_p, C.c_void_p]
    L.flow_noise(p.ctypes.data, dx.ctypes.data, dy.ctypes.data, par.ctypes.data, n, t.ctypes.data, q.ctypes.data)
    bad = (t.view(np.uint32) != q.view(np.uint32)) & ~(np.isnan(t) & np.isnan(q))
    assert not bad.any(), "%d of %d values differ (columns %s)" % (int(bad.sum()), bad.size, sorted(set(np.where(bad)[1].tolist())))
    assert np.abs(t[:, 0]).max() > 0.5 and (t[:256, 0] == 0).all() and t[:, 2].min() > 0         # gradient noise vanishes on the lattice; turbulence is a sum of magnitudes
    x = np.abs(par[:, 2].astype(np.float64)); ref = np.where(x < 1e-5, 1.0, np.where(x > 1, 0.0, np.sin(x * np.pi * 2) / np.maximum(x * np.pi * 2, 1e-30) * np.sin(x * np.pi) / np.maximum(x * np.pi, 1e-30)))
    assert np.abs(t[:, 4] - ref).max() < 1e-5


@pytest.mark.parametrize("kind,mapping", [(abi.TEX_CHECKERBOARD, abi.MAP_UV), (abi.TEX_CHECKERBOARD, abi.MAP_PLANAR), (abi.TEX_DOTS, abi.MAP_SPHERICAL), (abi.TEX_DOTS, abi.MAP_CYLINDRICAL),
                                          (abi.TEX_MARBLE, abi.MAP_IDENTITY3D), (abi.TEX_WINDY, abi.MAP_IDENTITY3D), (abi.TEX_WRINKLED, abi.MAP_IDENTITY3D), (abi.TEX_FBM, abi.MAP_IDENTITY3D),
                                          (abi.TEX_SCALE, abi.MAP_UV), (abi.TEX_MIX, abi.MAP_UV), (abi.TEX_IMAGE, abi.MAP_UV), (abi.TEX_IMAGE, abi.MAP_SPHERICAL)])
def test_texture_mappings_and_procedural_textures_text_equals_the_oracle(flow, oracle, kind, mapping):
    """UVMapping2D / SphericalMapping2D / CylindricalMapping2D / PlanarMapping2D / IdentityMapping3D::map (texture.rs:101-283, with the finite differences and the wrap fix-ups of the two
    angular mappings) and Checkerboard2DTexture / DotsTexture / MarbleTexture / WindyTexture / WrinkledTexture / FBmTexture::evaluate (textures/*.rs) against the oracle's tex_map2d / tex_map3d /
    tex_eval (which the device's texture code is held to)"""
    import ctypes as C
    mk, L = flow
    rng = np.random.default_rng(kind * 16 + mapping)
    tx = (abi.Texture * 3)()
    tx[0].kind, tx[0].mapping, tx[0].tex1, tx[0].tex2, tx[0].tex3 = kind, mapping, 1, 2, 2
    mp = rng.uniform(-2, 2, 8).astype(np.float32) if mapping == abi.MAP_PLANAR else np.array([3.0, 2.5, 0.25, -0.5, 0, 0, 0, 0], np.float32)
    for k in range(8):
        tx[0].map[k] = float(mp[k])
    rot = np.linalg.qr(rng.normal(size=(3, 3)))[0]
    w2t = np.eye(4); w2t[:3, :3] = rot * 1.7; w2t[:3, 3] = [0.3, -0.2, 0.1]
    for k in range(16):
        tx[0].world_to_texture[k] = float(np.float32(w2t.reshape(-1)[k]))
    tx[0].octaves, tx[0].omega, tx[0].scale, tx[0].variation = 6, 0.5, 1.3, 0.2
    for j, v in ((1, (1.0, 0.5, 0.25)), (2, (0.1, 0.2, 0.9))):
        tx[j].kind = abi.TEX_CONSTANT
        for k in range(3):
            tx[j].value[k] = v[k]
    n = 1 << 14
    si = np.concatenate([rng.uniform(-4, 4, (n, 3)), rng.uniform(-1, 2, (n, 2)), rng.normal(size=(n, 6)) * np.exp(rng.uniform(-8, -1, (n, 1))), rng.normal(size=(n, 4)) * 0.01], 1).astype(np.float32)
    si[:16, 5:] = 0.0
    t, q = np.zeros((n, 12), np.float32), np.zeros((n, 12), np.float32)
    L.flow_textures.restype = C.c_int
    L.flow_textures.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    img = None
    if kind == abi.TEX_IMAGE:                                                    # a 16 x 8 map with its pyramid, EWA, wrap mode Clamp
        levels, lw, lh = [], 16, 8
        while True:
            levels.append(rng.uniform(0, 2, (lh, lw, 3)).astype(np.float32))
            if lw == 1 and lh == 1:
                break
            lw, lh = max(1, lw // 2), max(1, lh // 2)
        texels = np.concatenate([l.reshape(-1) for l in levels])
        img = abi.Image(); img.width, img.height, img.n_levels, img.channels = 16, 8, len(levels), 3; img.texels = texels.ctypes.data
        tx[0].image, tx[0].trilinear, tx[0].max_aniso, tx[0].wrap = 0, 0, 8.0, abi.WRAP_CLAMP
        si[:, 11:15] *= 3
    assert L.flow_textures(C.addressof(tx), C.addressof(img) if img is not None else None, si.ctypes.data, n, t.ctypes.data, q.ctypes.data) == 0
    bad = (t.view(np.uint32) != q.view(np.uint32)) & ~(np.isnan(t) & np.isnan(q))
    assert not bad.any(), "%d of %d values differ (columns %s)" % (int(bad.sum()), bad.size, sorted(set(np.where(bad)[1].tolist())))
    assert t[:, :3].std() > 0.01 or kind in (abi.TEX_SCALE, abi.TEX_MIX)


def test_bump_mapping_text_equals_the_oracle(flow, oracle):
    """Material::bump (material.rs:116-219: the two shifted evaluations of the displacement, the step sizes from the uv differentials with the 0.0005 fall-back, the displaced dpdu / dpdv) and
    SurfaceInteraction::set_shading_geometry (interaction.rs:345-370: the new shading normal, turned to the geometric side), over a WrinkledTexture displacement, against the oracle's bump"""
    import ctypes as C
    mk, L = flow
    rng = np.random.default_rng(77)
    tx = (abi.Texture * 1)()
    tx[0].kind, tx[0].mapping = abi.TEX_WRINKLED, abi.MAP_IDENTITY3D
    w2t = np.eye(4) * 2.5; w2t[3, 3] = 1
    for k in range(16):
        tx[0].world_to_texture[k] = float(w2t.reshape(-1)[k])
    tx[0].octaves, tx[0].omega = 5, 0.6
    n = 1 << 14
    def unit(k):
        v = rng.normal(size=(k, 3)); return v / np.linalg.norm(v, axis=1)[:, None]
    nn = unit(n); sn = nn + rng.normal(size=(n, 3)) * 0.2; sn /= np.linalg.norm(sn, axis=1)[:, None]; sn[: n // 4] *= -1      # shading normals on either side of the geometric one
    du = np.cross(sn, unit(n)); dv = np.cross(sn, du) * rng.uniform(0.5, 2, (n, 1))
    si = np.concatenate([rng.uniform(-4, 4, (n, 3)), rng.uniform(0, 1, (n, 2)), rng.normal(size=(n, 6)) * np.exp(rng.uniform(-8, -2, (n, 1))), rng.normal(size=(n, 4)) * np.exp(rng.uniform(-8, -2, (n, 1))),
                         nn, sn, du, dv, rng.normal(size=(n, 6)) * 0.1], 1).astype(np.float32)
    si[:64, 11:15] = 0.0                                                           # no differentials: the 0.0005 step
    t, q = np.zeros((n, 9), np.float32), np.zeros((n, 9), np.float32)
    L.flow_bump.restype = C.c_int
    L.flow_bump.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    assert L.flow_bump(C.addressof(tx), si.ctypes.data, n, t.ctypes.data, q.ctypes.data) == 0
    bad = (t.view(np.uint32) != q.view(np.uint32)) & ~(np.isnan(t) & np.isnan(q))
    assert not bad.any(), "%d of %d values differ (columns %s)" % (int(bad.sum()), bad.size, sorted(set(np.where(bad)[1].tolist())))
    assert (np.abs(t[:, :3] - si[:, 18:21]).max(axis=1) > 1e-3).mean() > 0.5          # the bump turns the shading normal


@pytest.mark.parametrize("g", [0.0, 0.0005, 0.6, -0.85])
def test_homogeneous_medium_and_henyey_greenstein_text_equals_the_oracle(flow, oracle, g):
    """HomogeneousMedium::{tr, sample} (homogeneous.rs:33-91: the channel choice, the sampled distance, the medium interaction, transmittance / density / the throughput factor) and
    HenyeyGreenstein::{p, sample_p} (medium.rs:301-328, both branches of |g| < 1e-3) with spherical_direction_vec3 and Spectrum::exp, against the oracle's (which VolPath on the device is held to)"""
    import ctypes as C
    mk, L = flow
    rng = np.random.default_rng(int(abs(g) * 1000) + 3)
    m = abi.Medium()
    for k in range(3):
        m.sigma_a[k] = float(np.float32(rng.uniform(0.0, 2.0))); m.sigma_s[k] = float(np.float32(rng.uniform(0.05, 4.0)))
    m.sigma_a[1] = 0.0
    m.g = g
    n = 1 << 15
    def unit(k):
        v = rng.normal(size=(k, 3)); return v / np.linalg.norm(v, axis=1)[:, None]
    d = unit(n) * np.exp(rng.uniform(-2, 2, (n, 1)))
    tmax = np.exp(rng.uniform(-4, 3, n)); tmax[:64] = np.inf
    u = rng.uniform(0, 1, (n, 2)).astype(np.float32).clip(0, np.nextafter(np.float32(1), np.float32(0))); u[:16, 1] = 0.0; u[16:32, 0] = 0.0
    x = np.concatenate([rng.uniform(-3, 3, (n, 3)), d, tmax[:, None], u, unit(n), unit(n)], 1).astype(np.float32)
    t, q = np.zeros((n, 18), np.float32), np.zeros((n, 18), np.float32)
    L.flow_media.restype = None
    L.flow_media.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    L.flow_media(C.addressof(m), x.ctypes.data, n, t.ctypes.data, q.ctypes.data)
    bad = (t.view(np.uint32) != q.view(np.uint32)) & ~(np.isnan(t) & np.isnan(q))
    assert not bad.any(), "%d of %d values differ (columns %s)" % (int(bad.sum()), bad.size, sorted(set(np.where(bad)[1].tolist())))
    assert 0.1 < t[:, 6].mean() < 0.95 and t[:, 13].min() > 0


def test_animated_transform_decomposition_and_interpolation_text_equals_the_oracle(flow, oracle):
    """AnimatedTransform::decompose (transform.rs:2032-2080: the polar iteration over Matrix4x4::inverse / transpose, Quaternion::new on either branch of the trace, the scale), the quaternion
    flip and has_rotation of AnimatedTransform::new, and AnimatedTransform::interpolate (:2081-2113) over quat_slerp (both branches), Quaternion::to_transform, Transform::translate, the
    Transform product and mtx_mul — against the oracle's AnimatedTransform (which the moving camera and the moving instances on the device are held to): the interpolated matrix AND its inverse"""
    import ctypes as C
    mk, L = flow
    rng = np.random.default_rng(91)
    L.flow_animated.restype = None
    L.flow_animated.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float] + [C.c_void_p, C.c_uint64] + [C.c_void_p] * 4

    def rot(axis, ang):
        axis = axis / np.linalg.norm(axis); x, y, z = axis; c, s_ = np.cos(ang), np.sin(ang); k = np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]])
        return np.eye(3) + s_ * k + (1 - c) * (k @ k)
    n_pairs, n_rot, n_small = 0, 0, 0
    for trial in range(200):
        def key(small):
            m = np.eye(4)
            ang = rng.uniform(0, 0.02) if small else rng.uniform(0, 3.1)
            m[:3, :3] = rot(rng.normal(size=3), ang) @ np.diag(rng.uniform(0.3, 3, 3) if trial % 3 else np.ones(3) * rng.uniform(0.5, 2))
            if trial % 7 == 0:
                m[:3, :3] = m[:3, :3] @ (np.eye(3) + rng.normal(size=(3, 3)) * 0.1)          # shear: S is not diagonal
            m[:3, 3] = rng.uniform(-5, 5, 3)
            return m.astype(np.float32)
        a = key(False); b = key(trial % 5 == 0) if trial % 11 else a.copy()
        if trial % 5 == 0:
            b[:3, :3] = (a[:3, :3].astype(np.float64) @ rot(rng.normal(size=3), rng.uniform(0, 0.02))).astype(np.float32)          # nearly the same rotation: the nlerp branch of quat_slerp
        t0, t1 = 0.25, 1.5
        times = np.concatenate([[0.0, 0.25, 1.5, 2.0], rng.uniform(t0, t1, 60)]).astype(np.float32)
        n = len(times)
        ot, oo = np.zeros((n, 32), np.float32), np.zeros((n, 32), np.float32); dt, do = np.zeros(33, np.float32), np.zeros(33, np.float32)
        L.flow_animated(a.ctypes.data, b.ctypes.data, t0, t1, times.ctypes.data, n, ot.ctypes.data, oo.ctypes.data, dt.ctypes.data, do.ctypes.data)
        assert np.array_equal(dt.view(np.uint32), do.view(np.uint32)), (trial, dt, do)
        bad = (ot.view(np.uint32) != oo.view(np.uint32)) & ~(np.isnan(ot) & np.isnan(oo))
        assert not bad.any(), (trial, int(bad.sum()))
        n_pairs += 1; n_rot += int(dt[32]); n_small += int(dt[32] == 0)
    assert n_rot > 100 and n_small > 20


def test_transform_surface_interaction_text_equals_the_oracle(flow, oracle):
    """Transform::transform_surface_interaction (transform.rs:815-860: what a TransformedPrimitive does to an instance's hit) over transform_point_with_abs_error (:709-760, the error bound
    and the homogeneous divide), transform_normal (:528-537, the transposed inverse), transform_vector and nrm_faceforward_nrm, against the oracle's (which instancing on the device is held to)"""
    import ctypes as C
    mk, L = flow
    rng = np.random.default_rng(57)
    L.flow_instance.restype = None
    L.flow_instance.argtypes = [C.c_void_p] * 3 + [C.c_uint64, C.c_void_p, C.c_void_p]
    n = 1 << 12
    for trial in range(12):
        a = np.eye(4); a[:3, :3] = np.linalg.qr(rng.normal(size=(3, 3)))[0] @ np.diag(rng.uniform(0.3, 3, 3) * (1 if trial % 3 else -1)); a[:3, 3] = rng.uniform(-10, 10, 3)
        if trial == 5:
            a[3, :] = [0.01, -0.02, 0.005, 1.1]                                       # a projective row: the homogeneous divide
        m = a.astype(np.float32); mi = np.linalg.inv(m.astype(np.float64)).astype(np.float32)
        def unit(k):
            v = rng.normal(size=(k, 3)); return v / np.linalg.norm(v, axis=1)[:, None]
        nn = unit(n); sn = nn + rng.normal(size=(n, 3)) * 0.3; sn[: n // 3] *= -1
        si = np.concatenate([rng.uniform(-5, 5, (n, 3)), np.abs(rng.normal(size=(n, 3))) * 1e-6, nn, unit(n), rng.uniform(0, 1, (n, 1)), rng.uniform(0, 1, (n, 2)), rng.normal(size=(n, 6)),
                             sn, rng.normal(size=(n, 6)), rng.normal(size=(n, 6)) * 0.1, rng.normal(size=(n, 4)) * 0.01, rng.normal(size=(n, 6)) * 0.01], 1).astype(np.float32)
        t, q = np.zeros((n, 33), np.float32), np.zeros((n, 33), np.float32)
        L.flow_instance(m.ctypes.data, mi.ctypes.data, si.ctypes.data, n, t.ctypes.data, q.ctypes.data)
        bad = (t.view(np.uint32) != q.view(np.uint32)) & ~(np.isnan(t) & np.isnan(q))
        assert not bad.any(), (trial, int(bad.sum()), sorted(set(np.where(bad)[1].tolist())))


@pytest.mark.parametrize("moving", [False, True])
def test_transformed_primitive_text_equals_the_oracle(flow, oracle, moving):
    """TransformedPrimitive::intersect / intersect_p (primitive.rs:215-265: the interpolated primitive_to_world, Transform::inverse, the object-space ray, the t_max hand-back, the
    identity test, the hit taken to world space) against the oracle's transformed_intersect / prim_intersect_p on the instanced landscape stand-in, static and moving trees"""
    import ctypes as C
    mk, L = flow
    sc = scenes.landscape_standin(oracle.bvh_build, n_side=8, terrain=16, tree_grid=(4, 3), moving=moving)
    assert sc.desc.n_instances >= 12
    L.flow_transformed.restype = C.c_int
    L.flow_transformed.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(5 + moving)
    n = 1 << 12
    n_hit = 0
    for k in (0, 5, 11):
        tw = np.asarray(sc.instances[k]["to_world"], np.float64).reshape(-1)
        centre = np.array([tw[3], tw[7], tw[11]], np.float64)
        o = centre + rng.normal(size=(n, 3)) * 6 + [0, 4, 0]
        tgt = centre + rng.normal(size=(n, 3)) * 1.0 + [0, 1.0, 0]
        d = tgt - o; d /= np.linalg.norm(d, axis=1)[:, None]
        tmax = np.where(rng.uniform(size=n) < 0.3, rng.uniform(1, 12, n), np.inf)
        rays = np.concatenate([o, d, tmax[:, None], rng.uniform(-0.2, 1.2, (n, 1))], 1).astype(np.float32)
        t, q = np.zeros((n, 15), np.float32), np.zeros((n, 15), np.float32)
        assert L.flow_transformed(C.addressof(sc.desc), k, rays.ctypes.data, n, t.ctypes.data, q.ctypes.data) == 0
        bad = (t.view(np.uint32) != q.view(np.uint32)) & ~(np.isnan(t) & np.isnan(q))
        assert not bad.any(), (k, int(bad.any(axis=1).sum()), sorted(set(np.where(bad)[1].tolist())))
        n_hit += int(t[:, 0].sum())
    assert n_hit > 300


def test_light_power_and_the_power_distribution_text_equals_the_oracle(flow, oracle):
    """Light::power of the five light kinds over Bounds3f::bounding_sphere (the scene's radius: distant and infinite lights), Triangle::area and MipMap::lookup_pnt_flt, and the Distribution1D
    compute_light_power_distribution builds from their luminances, against the oracle's light_power (the `power` light strategy)"""
    import ctypes as C
    from tests.util import gallery, sky_scene
    mk, L = flow
    L.flow_light_power.restype = C.c_int
    L.flow_light_power.argtypes = [C.c_void_p] * 4
    kinds = set()
    for sc in (gallery(oracle.bvh_build, "all"), sky_scene(oracle.bvh_build, "image", with_area=True), scenes.cornell_box(oracle.bvh_build)):
        n = int(sc.desc.n_lights)
        t, q, r = np.zeros((n, 2), np.float32), np.zeros((n, 2), np.float32), np.zeros(2, np.float32)
        assert L.flow_light_power(C.addressof(sc.desc), t.ctypes.data, q.ctypes.data, r.ctypes.data) == n
        assert np.array_equal(r[:1].view(np.uint32), r[1:].view(np.uint32)) and r[0] > 0
        assert np.array_equal(t.view(np.uint32), q.view(np.uint32)), (t, q)
        assert (t[:, 0] > 0).all() and t[-1, 1] == 1.0
        kinds |= {int(k) for k in sc.lights["kind"]}
    assert kinds >= {abi.LIGHT_DIFFUSE_AREA, abi.LIGHT_POINT, abi.LIGHT_SPOT, abi.LIGHT_DISTANT, abi.LIGHT_INFINITE}


def test_film_setup_text_equals_the_host(flow):
    """Film::new's cropped pixel bounds and filter weight table (film.rs:187-214), GaussianFilter::evaluate / gaussian, and Film::get_sample_bounds (:266-292) against what the host side puts into
    the render description (scenes.make_render_desc, scenes.gaussian_filter_table): resolutions, crop windows and filter radii incl. fractional ones"""
    import ctypes as C
    mk, L = flow
    L.flow_film_setup.restype = None
    L.flow_film_setup.argtypes = [C.c_void_p] * 3
    rng = np.random.default_rng(3)
    for trial in range(300):
        xres, yres = int(rng.integers(1, 2000)), int(rng.integers(1, 1200))
        crop = (0.0, 1.0, 0.0, 1.0) if trial % 3 == 0 else tuple(float(np.float32(v)) for v in (rng.uniform(0, 0.5), rng.uniform(0.5, 1), rng.uniform(0, 0.5), rng.uniform(0.5, 1)))
        gauss = trial % 2 == 1
        radius = (2.0, 2.0) if trial % 5 == 0 else tuple(float(np.float32(v)) for v in rng.uniform(0.3, 3.5, 2))
        if not gauss and trial % 5 == 0:
            radius = (0.5, 0.5)
        alpha = 2.0 if trial % 4 else float(np.float32(rng.uniform(0.5, 3)))
        inp = np.array([xres, yres, *crop, 1 if gauss else 0, *radius, alpha], np.float32)
        b, t = np.zeros(8, np.int32), np.zeros(256, np.float32)
        L.flow_film_setup(inp.ctypes.data, b.ctypes.data, t.ctypes.data)
        table = scenes.gaussian_filter_table(radius, alpha) if gauss else None
        rd = scenes.make_render_desc(xres, yres, 1, scenes.CORNELL_LOOK_AT, 40.0, crop=crop, filter_radius=radius, filter_table=table)
        assert tuple(b[:4]) == tuple(rd.crop_px) and tuple(b[4:]) == tuple(rd.sample_bounds), (trial, xres, yres, crop, radius, b, tuple(rd.crop_px), tuple(rd.sample_bounds))
        host = np.array(rd.filter_table[:], np.float32)
        assert np.array_equal(t.view(np.uint32), host.view(np.uint32)), (trial, radius, alpha, int((t != host).sum()), np.abs(t - host).max())


def test_camera_setup_text_equals_the_host(flow):
    """Transform::look_at / perspective / scale / translate, the Transform product and inverse, and the projective chain of PerspectiveCamera::new (perspective.rs:59-79) against the matrices the
    host side puts into the render description (scenes.make_render_desc: raster_to_camera, camera_to_world)"""
    import ctypes as C
    mk, L = flow
    L.flow_camera_setup.restype = None
    L.flow_camera_setup.argtypes = [C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(8)
    worst = 0
    for trial in range(300):
        xres, yres = (400, 400) if trial == 0 else (int(rng.integers(16, 2000)), int(rng.integers(16, 1200)))
        fov = 40.0 if trial == 0 else float(np.float32(rng.uniform(10, 100)))
        look = scenes.CORNELL_LOOK_AT if trial == 0 else (tuple(float(np.float32(v)) for v in rng.uniform(-10, 10, 3)), tuple(float(np.float32(v)) for v in rng.uniform(-10, 10, 3)),
                                                            tuple(float(np.float32(v)) for v in rng.normal(size=3)))
        inp = np.array([xres, yres, fov, *look[0], *look[1], *look[2]], np.float32)
        out = np.zeros(32, np.float32)
        L.flow_camera_setup(inp.ctypes.data, out.ctypes.data)
        rd = scenes.make_render_desc(xres, yres, 1, look, fov)
        host = np.array(list(rd.raster_to_camera) + list(rd.camera_to_world), np.float32)
        bad = out.view(np.uint32) != host.view(np.uint32)
        worst = max(worst, int(bad.sum()))
        assert not bad.any(), (trial, xres, yres, fov, np.where(bad)[0], out[bad], host[bad])


def test_spot_and_distant_light_setup_text_equals_the_host(flow):
    """what api.rs makes of LightSource "spot" / "distant" (the direction, vec3_coordinate_system's frame, the cone cosines) against the host side's light records (scenes.SceneBuilder)"""
    import ctypes as C
    mk, L = flow
    L.flow_light_setup.restype = None
    L.flow_light_setup.argtypes = [C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(12)
    for trial in range(300):
        f = tuple(float(np.float32(v)) for v in rng.uniform(-10, 10, 3)); t = tuple(float(np.float32(v)) for v in rng.uniform(-10, 10, 3))
        cone, delta = float(np.float32(rng.uniform(5, 80))), float(np.float32(rng.uniform(0, 5)))
        out = np.zeros(14, np.float32)
        inp = np.array([*f, *t, cone, delta], np.float32)
        L.flow_light_setup(inp.ctypes.data, out.ctypes.data)
        sb = scenes.SceneBuilder()
        sb.add_spot_light(f, t, (1, 1, 1), cone, delta); sb.add_distant_light(f, t, (1, 1, 1))
        spot, dist = sb.delta_lights[0], sb.delta_lights[1]
        host = np.concatenate([spot["p"][3:14], dist["p"][:3]]).astype(np.float32)
        assert np.array_equal(out.view(np.uint32), host.view(np.uint32)), (trial, out, host)
    L.flow_rotate_y.restype = None
    L.flow_rotate_y.argtypes = [C.c_float, C.c_void_p]
    for theta in [0.0, 6.0, 90.0, 180.0] + [float(np.float32(v)) for v in rng.uniform(0, 360, 300)]:      # Transform::rotate_y (the instances of the landscape stand-in)
        out = np.zeros(32, np.float32)
        L.flow_rotate_y(theta, out.ctypes.data)
        h = scenes.Transform.rotate_y(theta)
        assert np.array_equal(out.view(np.uint32), np.concatenate([h.m.reshape(-1), h.m_inv.reshape(-1)]).astype(np.float32).view(np.uint32)), theta


def test_envmap_distribution_image_text_equals_the_host(flow):
    """the scalar image InfiniteAreaLight::new hands to Distribution2D::new (infinite.rs:133-146: a trilinear lookup at half a texel's width, luminance times sin theta) against the host's
    scenes.build_envmap, on the host's own pyramid"""
    import ctypes as C
    mk, L = flow
    L.flow_envmap_image.restype = None
    L.flow_envmap_image.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    rng = np.random.default_rng(2)
    for (h, w) in ((4, 8), (16, 32), (1, 1), (64, 64)):
        img = rng.uniform(0.01, 3, (h, w, 3)).astype(np.float32); img[0, 0] = 40.0
        e = scenes.build_envmap(img)
        out = np.zeros(e["dist_nu"] * e["dist_nv"], np.float32)
        L.flow_envmap_image(e["texels"].ctypes.data, w, h, e["n_levels"], out.ctypes.data)
        host = np.ascontiguousarray(e["dist_func"], np.float32).reshape(-1)
        bad = out.view(np.uint32) != host.view(np.uint32)
        assert not bad.any(), ((h, w), int(bad.sum()), out[bad][:4], host[bad][:4])


@pytest.mark.parametrize("wrap", [abi.WRAP_REPEAT, abi.WRAP_BLACK, abi.WRAP_CLAMP])
def test_mipmap_pyramid_text_equals_the_host(flow, wrap):
    """the level-by-level box filter of MipMap::new (mipmap.rs:166-184, four texels of the finer level through texel()'s wrap mode) against the pyramids the host builds
    (scenes.build_image / build_envmap)"""
    import ctypes as C
    mk, L = flow
    L.flow_pyramid.restype = C.c_int
    L.flow_pyramid.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    rng = np.random.default_rng(6 + wrap)
    for (h, w) in ((8, 8), (4, 32), (64, 16), (1, 1), (2, 1)):
        img = rng.uniform(0, 2, (h, w, 3)).astype(np.float32)
        e = scenes.build_image(img, wrap=wrap)
        host = np.ascontiguousarray(e["texels"], np.float32).reshape(-1)
        out = np.zeros(len(host), np.float32)
        level0 = host[: img.size].copy()                                        # (the host's level 0: ImageTexture::new has flipped / scaled the file's texels)
        n = L.flow_pyramid(level0.ctypes.data, w, h, wrap, out.ctypes.data)
        assert n == e["n_levels"], ((h, w), n, e["n_levels"])
        got = np.concatenate([level0, out[: len(host) - img.size]])
        assert np.array_equal(got.view(np.uint32), host.view(np.uint32)), ((h, w), int((got != host).sum()))
