"""The DEVICE's traversal and shading geometry against the REFERENCE'S OWN TEXT, with no oracle in between (round 6, third session).

tests/golden/geom_functions.npz holds what the Rust text of Bounds3f::intersect_p, Triangle::intersect's watertight test, pnt3_offset_ray_origin, the Trowbridge-Reitz
terms, vec3_cross_vec3, vec3_coordinate_system, refract and cosine_sample_hemisphere computes on 2^12 seeded cases each (oracle/make_geom_fixtures.py compiles that text
by committed rewrite rules).  rspt_libm's codes 8 .. 12 run the functions the kernels call (dev_scene.h tri_test, kernels.h box_hit, trace_w4.h box_pair_hit_m /
box_hit6_m, dev_math.h, dev_bsdf.h) over the same inputs: every output must be the same BITS."""
import os

import numpy as np
import pytest

from rs_pbrt_amd import lib

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def g():
    lib.init(0)
    return np.load(os.path.join(HERE, "golden", "geom_functions.npz"))


def pack(n, *cols):
    x = np.zeros((n, 16), np.float32)
    k = 0
    for c in cols:
        c = np.asarray(c, np.float32).reshape(n, -1)
        x[:, k:k + c.shape[1]] = c
        k += c.shape[1]
    return x


def same_bits(a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


def test_the_watertight_triangle_test_is_the_references_text(g):
    n = len(g["tri_p"])
    x = pack(n, g["tri_p"], g["tri_o"], g["tri_d"])
    x[:, 15] = g["tri_tmax"]
    out = lib.leaf_geom("triangle", x)
    assert same_bits(out[:, :5], g["tri_out"]), "Triangle::intersect: %d of %d cases differ" % (int((out[:, :5].view(np.uint32) != g["tri_out"].view(np.uint32)).any(axis=1).sum()), n)
    assert same_bits(out[:, :5], g["trp_out"])          # Triangle::intersect_p repeats the test
    assert (out[:, 5:] == 0).all() and 0.3 < out[:, 0].mean() < 0.8


def test_the_box_test_is_the_references_text_in_every_form_the_kernels_use(g):
    n = len(g["box_b"])
    x = pack(n, g["box_b"], g["box_o"], g["box_inv"], g["box_neg"].astype(np.float32))
    x[:, 15] = g["box_tmax"]
    out = lib.leaf_geom("box", x)
    ref = g["box_out"]
    assert np.array_equal(out[:, 0], ref), "k_trace's box_hit: %d of %d differ" % (int((out[:, 0] != ref).sum()), n)
    assert np.array_equal(out[:, 1], ref) and np.array_equal(out[:, 2], ref), "k_trace_w4's forms: %d / %d of %d differ" % (int((out[:, 1] != ref).sum()), int((out[:, 2] != ref).sum()), n)
    finite = np.isfinite(g["box_inv"]).all(axis=1)
    assert np.array_equal(out[:, 3] == 1, finite) and 100 < (~finite).sum() < n - 100      # both of the w4 kernel's forms were exercised


def test_offset_ray_origin_and_the_microfacet_terms_are_the_references_text(g):
    n = len(g["oro_p"])
    out = lib.leaf_geom("offset_ray_origin", pack(n, g["oro_p"], g["oro_e"], g["oro_n"], g["oro_w"]))
    assert same_bits(out[:, :3], g["oro_out"])
    out = lib.leaf_geom("microfacet", pack(n, g["mf_wo"], g["mf_wh"], g["mf_ax"], g["mf_ay"]))
    assert same_bits(out[:, :5], g["mf_out"]), "Trowbridge-Reitz d / lambda / g1 / g / pdf: columns differing %s" % (out[:, :5].view(np.uint32) != g["mf_out"].view(np.uint32)).sum(axis=0)


def test_cross_coordinate_system_refract_and_the_cosine_hemisphere_are_the_references_text(g):
    n = len(g["vec_a"])
    out = lib.leaf_geom("vectors", pack(n, g["vec_a"], g["vec_b"], g["rfr_eta"], g["smp_u"]))
    assert same_bits(out[:, 0:3], g["crs_out"])
    assert same_bits(out[:, 3:9], g["cs_out"])
    ok = g["rfr_out"][:, 3] == 1
    assert np.array_equal(out[:, 12] == 1, ok) and same_bits(out[ok, 9:12], g["rfr_out"][ok, :3])
    assert same_bits(out[:, 13:16], g["csh_out"])


def test_area_light_sampling_is_the_references_text(g):
    """DiffuseAreaLight::sample_li over Triangle::sample / sample_with_ref_point as the shade and directlighting kernels call it (dev_scene.h light_sample_li): pdf, wi,
    radiance, the sampled point, its normal and error bound — the cases of the fixture without vertex normals (the hook carries one triangle, no scene)"""
    sel = (g["al_flags"] & 1) == 0
    n = int(sel.sum())
    assert n > 1500
    x = pack(n, g["al_tri"][sel], g["al_ref"][sel], g["al_u"][sel], g["al_flags"][sel].astype(np.float32))
    out = lib.leaf_geom("area_light", x)
    ref = g["al_out"][sel]
    lit = ref[:, 4:7].sum(axis=1) > 0                                  # the fixture's radiance is L or 0; the hook's light has L = 1
    assert same_bits(out[:, 0:4], ref[:, 0:4]) and np.array_equal(out[:, 4] == 1, lit)
    assert same_bits(out[:, 5:14], ref[:, 7:16])


def test_the_traversal_kernels_walk_the_tree_as_the_references_text_does(g):
    """rspt_trace (k_trace_w4 closest / any, and with it the four-box records, the persistent waves, the deferred leaf phase) against BVHAccel::intersect / intersect_p
    compiled from the reference's text (bvh.rs:401-514 over primitive.rs:150-156, geometry.rs:2211-2268, triangle.rs:134-273): the hit record and the occlusion flag of
    every committed ray — incoherent, axis-parallel, leaving a triangle along its own edge, segments ending on a surface — bit for bit."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
    import make_geom_fixtures as mk
    from rs_pbrt_amd import abi
    sc = mk.traversal_scene(lib.bvh_build)
    assert np.array_equal(mk.tree_digest(sc), g["trv_tree"]), "the library's host builder gives another tree than the one the fixture was walked on"
    rays = np.zeros(len(g["trv_o"]), abi.RAY_DT)
    rays["o"], rays["d"], rays["t_max"] = g["trv_o"], g["trv_d"], g["trv_tmax"]
    with lib.DeviceScene(sc) as ds:
        h = lib.trace(ds, rays)
        a = lib.trace(ds, rays, any_hit=True)
    assert np.array_equal(h["prim"], g["trv_prim"]), "%d of %d rays end on another primitive" % (int((h["prim"] != g["trv_prim"]).sum()), len(rays))
    assert same_bits(np.stack([h["t"], h["b0"], h["b1"], h["b2"]], 1), g["trv_tb"])
    assert np.array_equal((a["prim"] != abi.MISS).astype(np.uint8), g["trv_any"])


def test_the_device_lobes_are_the_references_text():
    """lobe_f / lobe_pdf / lobe_sample_f (dev_bsdf.h, every one of the nine lobe kinds, with and without a MixMaterial scale) as the shade kernels call them against
    reflection.rs:711-1478 compiled from the reference's text (tests/golden/lobe_functions.npz, oracle/make_flow_fixtures.py): f, pdf, the sampled direction, its pdf,
    sampled_type and get_type of every case; the sampled VALUE for the specular lobes (a non-specular lobe's own value is never read: Bsdf::sample_f re-sums f over the
    matching lobes, and the reference's text carries a MixMaterial scale twice there — DESIGN.md section 3a)"""
    lib.init(0)
    g = np.load(os.path.join(HERE, "golden", "lobe_functions.npz"))
    from rs_pbrt_amd import abi
    rec = g["records"].view(abi.BXDF_DT).reshape(-1) if g["records"].dtype != abi.BXDF_DT else g["records"]
    out = lib.leaf_lobe(rec, g["wo"], g["wi"], g["u"])
    ref = g["text"]
    for cols, name in (((0, 1, 2), "f"), ((3,), "pdf"), ((7, 8, 9), "sampled wi"), ((10,), "sampled pdf"), ((11,), "sampled_type"), ((12,), "get_type")):
        c = list(cols)
        live = np.ones(len(rec), bool) if name not in ("sampled wi",) else ref[:, 10] > 0          # (a direction is only defined where the sample has a pdf)
        assert same_bits(out[live][:, c], ref[live][:, c]), "%s: %d of %d cases differ" % (name, int((out[live][:, c].view(np.uint32) != ref[live][:, c].view(np.uint32)).any(axis=1).sum()), int(live.sum()))
    spec = np.isin(rec["type"], [abi.BXDF_SPECULAR_R, abi.BXDF_SPECULAR_T, abi.BXDF_FRESNEL_SPEC])
    assert same_bits(out[spec][:, 4:7], ref[spec][:, 4:7])
    plain = ~spec & (rec["has_sc"] == 0)
    assert same_bits(out[plain][:, 4:7], ref[plain][:, 4:7])              # without a scale the value of a non-specular lobe's own sample_f agrees too
