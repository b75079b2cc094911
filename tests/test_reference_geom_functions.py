"""The geometry of a traversal step, the sampling helpers, the Trowbridge-Reitz terms and the PCG32 generator held to the REFERENCE'S OWN TEXT (round 6, third session;
the route of tests/test_reference_leaf_functions.py, second batch).

oracle/make_geom_fixtures.py compiles — syntax rewritten by committed rules, no hand-edited body — Bounds3f::intersect_p (geometry.rs:2211-2268), the watertight test of
Triangle::intersect / intersect_p (triangle.rs:134-273, 450-591: everything in front of the SurfaceInteraction), pnt3_offset_ray_origin with next_float_up / _down and
gamma, vec3_cross_vec3 (its f64 products), vec3_coordinate_system, reflect / refract, power_heuristic, cosine_ / uniform_sample_hemisphere,
TrowbridgeReitzDistribution::{roughness_to_alpha, d, lambda, g1, g, pdf}, phase_hg, RGBSpectrum::y and Rng::{set_sequence, uniform_uint32, uniform_uint32_bounded,
uniform_float} from the Rust text where it lies; tests/golden/geom_functions.npz holds 2^12 seeded cases per function with that code's outputs.  The ORACLE's
restatements (and through tests/test_gpu_*.py the HIP kernels, sample for sample) must give the same BITS.  Where /root/reference exists the fixture is regenerated
and compared, and 2^17 fresh cases per function run through both side by side."""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HAVE_REF = os.path.exists("/root/reference/src/shapes/triangle.rs")
NAMES = {"gam_out": "gamma", "nfu_out": "next_float_up", "nfd_out": "next_float_down", "ph_out": "power_heuristic", "rta_out": "roughness_to_alpha", "hg_out": "phase_hg",
         "y_out": "RGBSpectrum::y", "csh_out": "cosine_sample_hemisphere", "ush_out": "uniform_sample_hemisphere", "crs_out": "vec3_cross_vec3", "cs_out": "vec3_coordinate_system",
         "rfl_out": "reflect", "rfr_out": "refract", "adt_out": "vec3_abs_dot_vec3f", "oro_out": "pnt3_offset_ray_origin", "box_out": "Bounds3f::intersect_p",
         "tri_out": "Triangle::intersect (watertight test)", "trp_out": "Triangle::intersect_p (watertight test)", "mf_out": "TrowbridgeReitzDistribution d / lambda / g1 / g / pdf",
         "rng_u_out": "Rng uniform_uint32 / _bounded", "rng_f_out": "Rng::uniform_float",
         "trf_out": "the whole Triangle::intersect: hit point, error bound, normals, uv, dpdu / dpdv, the shading frame and dndu / dndv",
         "flm_out": "Film::get_film_tile, FilmTile::add_sample, Film::merge_film_tile (box and gaussian filter tables, tiles across the frame's border, the luminance clamp)",
         "mor_out": "morton2 / part1_by1 (the tile order of BlockQueue)",
         "dif_out": "SurfaceInteraction::compute_differentials over solve_linear_system_2x2",
         "al_out": "DiffuseAreaLight::sample_li / l over Triangle::sample / sample_with_ref_point",
         "sob_out": "SobolSampler start_pixel / get_camera_sample / get_1d / get_2d / start_next_sample over sobol_interval_to_index, sobol_sample",
         "hpc_out": "compute_radical_inverse_permutations (entry count)", "hph_out": "compute_radical_inverse_permutations over shuffle and Rng (first 8192 entries)",
         "hps_out": "compute_radical_inverse_permutations (SHA-256 of all 3 682 913 entries)",
         "rad_out": "radical_inverse / scrambled_radical_inverse over all 1000 prime bases", "radi_out": "reverse_bits_32 / reverse_bits_64 / inverse_radical_inverse",
         "halm_out": "HaltonSampler::new (base scales / exponents, stride, multiplicative inverses over extended_gcd, mod_t) and get_index_for_sample per pixel sample",
         "pix_out": "ZeroTwoSequence / MaxMinDist / Stratified / RandomSampler: new, reseed, start_pixel, get_camera_sample, get_1d / get_2d, the 2-D arrays, start_next_sample over van_der_corput, sobol_2d, "
                    "gray_code_sample_1d / _2d, sample_generator_matrix, stratified_sample_1d / _2d, latin_hypercube, shuffle",
         "pixm_out": "the samplers' spp after new (MaxMinDist rounds up to a power of two) and round_count",
         "hal_out": "HaltonSampler start_pixel / get_camera_sample / get_1d / get_2d / request_2d_array / get_2d_array_idxs / get_2d_sample / start_next_sample"}


def differing(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    if a.dtype == np.float32:
        return int(((a.view(np.uint32) != b.view(np.uint32)) & ~(np.isnan(a) & np.isnan(b))).sum())   # (a NaN has many encodings)
    return int((a != b).sum())


def generator():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import make_geom_fixtures
    return make_geom_fixtures


def test_oracle_equals_the_references_text_on_the_committed_fixture(oracle):
    g = np.load(os.path.join(HERE, "golden", "geom_functions.npz"))
    assert len(g["gam_n"]) == 1 << 12
    got = oracle.geom(g)
    assert set(got) == set(NAMES)
    assert set(g.files) == set(NAMES) | {k for k in g.files if not k.endswith("_out")} and {"trv_prim", "trv_tb", "trv_any", "trv_tree"} <= set(g.files)
    for k, name in NAMES.items():
        assert differing(got[k], g[k]) == 0, "%s: the oracle's restatement differs from the reference's text in %d of %d outputs" % (name, differing(got[k], g[k]), g[k].size)
    # the fixture exercises the branches it is there for
    tri, box = g["tri_out"], g["box_out"]
    assert 0.3 < tri[:, 0].mean() < 0.8 and 0.15 < box.mean() < 0.7                       # hits and misses
    assert ((tri[:, 0] == 1) & ((tri[:, 2:] == 0).any(axis=1))).sum() > 50                # hits ON an edge / through a vertex: the f64 fall-back decides them
    assert (np.isinf(g["box_inv"]).any(axis=1) & (box == 1)).sum() > 20                   # axis-parallel rays that pass a box (infinite reciprocals)
    assert np.isfinite(g["tri_tmax"]).sum() > 1000 and ((tri[:, 0] == 0) & np.isfinite(g["tri_tmax"])).sum() > 200   # t_max in front of the hit
    assert (g["rfr_out"][:, 3] == 0).sum() > 50                                           # total internal reflection
    assert (g["mf_out"][:16, 0] > 0).all() and (g["mf_out"][16:24, 1] == 0).all()         # D at normal incidence, lambda at grazing incidence (infinite tangent)
    assert (g["oro_out"] != g["oro_p"]).any(axis=1).mean() > 0.9                          # the offset moved the origin, rounded away from it
    al = g["al_out"]
    assert (al[:, 0] == 0).sum() > 10 and (al[:, 0] > 0).sum() > 3000 and ((al[:, 0] > 0) & (al[:, 4:7] == 0).all(axis=1)).sum() > 300     # zero / infinite pdf -> 0; one-sided lights seen from behind
    dif = g["dif_out"]
    assert (dif == 0).all(axis=1).sum() > 300 and (dif[:, :4] != 0).all(axis=1).sum() > 2500 and ((dif[:, :2] == 0).all(axis=1) & (dif[:, 4:] != 0).any(axis=1)).sum() > 50   # no differential / a grazing offset ray; the regular case; a singular 2 x 2 system
    flm = g["flm_out"]
    assert (flm[:, :, 3] > 0).any(axis=1).mean() > 0.8 and (flm[:, :, 3] == 0).any()             # every tile reached the film; pixels no sample reached stay zero
    sob = g["sob_out"]
    assert ((sob[:, :, 0] >= g["sob_pixel"][:, None, 0]) & (sob[:, :, 0] < g["sob_pixel"][:, None, 0] + 1)).all()     # the film sample lies in its pixel (the remap of dimensions 0 / 1)
    assert (sob[:, :, 25] == 0).any() and (sob[:, :, 25] == 1).any() and (g["sob_bounds"][:, :2] != 0).any()           # the last sample of a pixel; cropped sample bounds
    hal, meta = g["hal_out"], g["halm_out"]
    free = g["hal_center"] == 0
    assert ((hal[free, :, 0] >= g["hal_pixel"][free, None, 0]) & (hal[free, :, 0] < g["hal_pixel"][free, None, 0] + 1)).all()       # the pixel's Halton offset puts the film sample in its pixel
    assert (hal[~free, :, 0] == g["hal_pixel"][~free, None, 0] + 0.5).all() and (~free).sum() > 100                                    # samplepixelcenter
    assert (meta[:, 0] == 128).any() and (meta[:, 1] == 243).any() and (meta[:, 4] == 1).any() and (g["hal_pixel"] < 0).any()          # saturated base scales (K_MAX_RESOLUTION), a 1 x 1 frame (stride 1), mod_t of a negative pixel
    assert (hal[:, 0, 25] >= 0).sum() > 500 and (hal[:, 0, 29] >= 0).sum() > 300 and (hal[:, :, 25:33] == -1).any()                   # one and two requested arrays, and none
    pix, kind = g["pix_out"], g["pix_kind"]
    assert all((kind == k).sum() > 100 for k in range(4)) and (g["pix_par"][kind == 2, 4] == 0).sum() > 30 and (g["pix_par"][:, 1] == 0).any()      # every sampler; unjittered strata; no sampled dimension at all
    assert (g["pixm_out"][(kind == 1) & (g["pix_par"][:, 0] == 5), 0] == 8).all() and g["pixm_out"][4, 0] == 65536 and g["pixm_out"][6, 0] == 128   # MaxMinDistSampler::new rounds 5 -> 8, 100 -> 128
    assert (g["pixm_out"][kind <= 1, 1] == 4).all() and (g["pixm_out"][kind >= 2, 1] == 3).all()                                                # round_count(3)
    assert (g["rad_bi"] == 0).any() and (g["rad_bi"] == 999).any() and int(g["hpc_out"][0]) == 3682913                                # base 2 by bit reversal, the last base, the sum of the first 1000 primes


def oracle_traversal(oracle, sc, o, d, tmax):
    from rs_pbrt_amd import abi
    rays = np.zeros(len(o), abi.RAY_DT)
    rays["o"], rays["d"], rays["t_max"] = o, d, tmax
    h = oracle.trace(sc, rays)
    a = oracle.trace(sc, rays, any_hit=True)
    return h["prim"], np.stack([h["t"], h["b0"], h["b1"], h["b2"]], 1), (a["prim"] != abi.MISS).astype(np.uint8)


def test_oracle_traversal_equals_the_references_traversal_text_on_the_committed_rays(oracle):
    """BVHAccel::intersect / intersect_p (bvh.rs:401-514) with GeometricPrimitive::intersect's t_max update (primitive.rs:150-156), Bounds3f::intersect_p and the
    watertight triangle test underneath — all compiled from the reference's text — over a 20 000-triangle soup: the nodes a ray visits, in which order, with which
    shrinking t_max.  The hit record (primitive, t, b0, b1, b2) and the occlusion flag of every ray must be the oracle's, bit for bit."""
    mk = generator()
    g = np.load(os.path.join(HERE, "golden", "geom_functions.npz"))
    sc = mk.traversal_scene(oracle.bvh_build)
    assert np.array_equal(mk.tree_digest(sc), g["trv_tree"]), "the oracle's builder no longer gives the tree the fixture was walked on"
    prim, tb, occ = oracle_traversal(oracle, sc, g["trv_o"], g["trv_d"], g["trv_tmax"])
    assert np.array_equal(prim, g["trv_prim"]) and differing(tb, g["trv_tb"]) == 0 and np.array_equal(occ, g["trv_any"])
    assert len(prim) == 1 << 13 and 0.5 < (prim != 0xffffffff).mean() < 0.9
    assert ((occ == 0) & np.isfinite(g["trv_tmax"])).sum() > 100 and ((occ == 1) & np.isfinite(g["trv_tmax"])).sum() > 200      # segments that end in front of / behind a surface
    assert (sc.nodes["n_prims"] >= 3).sum() > 100                                                                              # leaves of several triangles: the in-leaf order


@pytest.mark.skipif(not HAVE_REF, reason="needs /root/reference to compile the reference's text")
def test_oracle_traversal_equals_the_compiled_traversal_text_on_131072_fresh_rays(oracle):
    mk = generator()
    L, _ = mk.convert()
    sc = mk.traversal_scene(oracle.bvh_build)
    o, d, tmax = mk.traversal_rays(sc, 1 << 17, 0xF2E5)
    ref = mk.run_traversal(L, sc, o, d, tmax)
    prim, tb, occ = oracle_traversal(oracle, sc, o, d, tmax)
    assert np.array_equal(prim, ref["trv_prim"]), "%d rays end on another primitive" % int((prim != ref["trv_prim"]).sum())
    assert differing(tb, ref["trv_tb"]) == 0 and np.array_equal(occ, ref["trv_any"])


@pytest.mark.skipif(not HAVE_REF, reason="the reference tree is not on this machine: the committed fixture is what travels")
def test_committed_fixture_is_what_the_references_text_gives():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "make_geom_fixtures.py"), "--check"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.skipif(not HAVE_REF, reason="needs /root/reference to compile the reference's text")
def test_oracle_equals_the_compiled_reference_text_on_131072_fresh_cases_per_function(oracle):
    mk = generator()
    L, where = mk.convert()
    assert len(where) == 18 + 3 + len(mk.SOURCES)      # (the base batch; PRIMES, PRIME_SUMS, C_MAX_MIN_DIST; this batch)
    lines = dict(w.rsplit(" ", 1) for w in where)
    assert lines["Bounds3f::intersect_p"] == "core/geometry.rs:2211-2268" and lines["Triangle::intersect"] == "shapes/triangle.rs:134-273"
    assert lines["BVHAccel::intersect"] == "accelerators/bvh.rs:401-462" and lines["BVHAccel::intersect_p"] == "accelerators/bvh.rs:463-514"
    assert lines["radical_inverse"] == "core/lowdiscrepancy.rs:1126-2162" and lines["HaltonSampler::get_index_for_sample"] == "samplers/halton.rs:173-214" and lines["PRIMES"] == "core/lowdiscrepancy.rs:20-82"
    assert lines["sobol_2d"] == "core/lowdiscrepancy.rs:919-1010" and lines["MaxMinDistSampler::start_pixel"] == "samplers/maxmin.rs:116-160" and lines["latin_hypercube"] == "core/sampling.rs:273-306"
    d = mk.inputs(n=1 << 17, seed=0x5EED7)
    ref = mk.run_reference(L, d)
    got = oracle.geom(d)
    for k, name in NAMES.items():
        assert differing(got[k], ref[k]) == 0, "%s: %d of %d outputs differ" % (name, differing(got[k], ref[k]), ref[k].size)
