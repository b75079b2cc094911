"""TEST INFRASTRUCTURE — ctypes loader for the CPU oracle (oracle/liboracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The product package (rs_pbrt_amd) never does.

Parity: pinned by real rs_pbrt output on one scene family only — the reference's two documentation renders of the Cornell box
(tests/test_reference_pin.py: path, Sobol', matte, area light, spatial light distribution, film).  For everything else the
reference has no tests or golden vectors and no Rust toolchain exists here: unpinned, checked against first-principles known
answers (tests/test_oracle_*.py), not against rs_pbrt output.  Round 6 adds pins by the reference's own TEXT: seven scalar leaf functions
(fr_dielectric, fr_conductor, trowbridge_reitz_sample_11 / _sample, sobol_sample_float, concentric_sample_disk, Matrix4x4::inverse) compiled from the Rust
sources by oracle/make_leaf_fixtures.py and equal to this oracle bit for bit (tests/test_reference_leaf_functions.py), AnimatedTransform's derivative
polynomials (oracle/make_motion_fixture.py, round 5), the Sobol' / max-min-distance / prime tables (tests/test_reference_tables.py), and — second batch,
oracle/make_geom_fixtures.py / tests/test_reference_geom_functions.py — the traversal's box test (Bounds3f::intersect_p), the watertight test of Triangle::intersect /
intersect_p, pnt3_offset_ray_origin, vec3_cross_vec3, vec3_coordinate_system, reflect / refract, power_heuristic, the hemisphere samplers, the Trowbridge-Reitz terms,
phase_hg, RGBSpectrum::y and the PCG32 generator.  Control flow stays unpinned."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        so = os.environ.get("ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")   # ORACLE_LIB: the sanitizer build (make -C oracle san; tests/test_oracle_sanitizers.py)
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.orc_next_float_up.restype = C.c_float; L.orc_next_float_up.argtypes = [C.c_float]
        L.orc_next_float_down.restype = C.c_float; L.orc_next_float_down.argtypes = [C.c_float]
        L.orc_gamma.restype = C.c_float; L.orc_gamma.argtypes = [C.c_int]
        L.orc_radical_inverse.restype = C.c_float; L.orc_radical_inverse.argtypes = [C.c_int, C.c_uint64]
        L.orc_sobol_index.restype = C.c_uint64; L.orc_sobol_index.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32, C.c_int32]
        L.orc_sobol_sample.restype = C.c_float; L.orc_sobol_sample.argtypes = [C.c_void_p, C.c_int64, C.c_int]
        L.orc_fr_dielectric.restype = C.c_float; L.orc_fr_dielectric.argtypes = [C.c_float] * 3
        L.orc_distribution1d.restype = C.c_int64
        L.orc_distribution1d.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_bvh_build.restype = C.c_int64
        L.orc_bvh_build.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p]
        L.orc_bvh_build_bounds.restype = C.c_int64
        L.orc_bvh_build_bounds.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p]
        L.orc_transform_bounds.restype = None; L.orc_transform_bounds.argtypes = [C.c_void_p] * 4
        L.orc_trace.restype = None
        L.orc_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_render.restype = C.c_int
        L.orc_render.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_render_integrator.restype = C.c_int
        L.orc_render_integrator.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_camera_sample.restype = None
        L.orc_camera_sample.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p]
        L.orc_camera_ray.restype = None; L.orc_camera_ray.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_camera_matrix.restype = None; L.orc_camera_matrix.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
        L.orc_offset_ray_origin.restype = None; L.orc_offset_ray_origin.argtypes = [C.c_void_p] * 5
        L.orc_concentric_sample_disk.restype = None; L.orc_concentric_sample_disk.argtypes = [C.c_float, C.c_float, C.c_void_p]
        L.orc_bsdf_f.restype = None
        L.orc_bsdf_f.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_bsdf_sample_f.restype = None
        L.orc_bsdf_sample_f.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_material_lobes.restype = C.c_int
        L.orc_material_lobes.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_halton_permutations.restype = C.c_uint64; L.orc_halton_permutations.argtypes = [C.c_int, C.c_void_p]
        L.orc_halton_index.restype = C.c_uint64; L.orc_halton_index.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_uint64]
        L.orc_halton_sample.restype = C.c_float; L.orc_halton_sample.argtypes = [C.c_void_p, C.c_uint64, C.c_int]
        L.orc_pcg32_next.restype = C.c_uint32; L.orc_pcg32_next.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_phase_hg.restype = C.c_float; L.orc_phase_hg.argtypes = [C.c_float, C.c_float]
        L.orc_hg_sample_p.restype = C.c_float; L.orc_hg_sample_p.argtypes = [C.c_float, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        L.orc_homogeneous_sample.restype = None
        L.orc_homogeneous_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p]
        L.orc_visibility_tr.restype = None; L.orc_visibility_tr.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_grid_density.restype = None; L.orc_grid_density.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.orc_grid_tr.restype = C.c_int
        L.orc_grid_tr.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.orc_grid_sample.restype = C.c_int
        L.orc_grid_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.orc_libm.restype = None; L.orc_libm.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.orc_pixel_sampler.restype = None
        L.orc_pixel_sampler.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_pixel_sampler_arrays.restype = None
        L.orc_pixel_sampler_arrays.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_round_count.restype = C.c_int32; L.orc_round_count.argtypes = [C.c_void_p, C.c_int32]
        L.orc_spatial_voxel.restype = None
        L.orc_spatial_voxel.argtypes = [C.c_void_p] * 6
        L.orc_tex_eval.restype = None; L.orc_tex_eval.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_camera_ray_diff.restype = None; L.orc_camera_ray_diff.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_interpolate_transform.restype = None
        L.orc_interpolate_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.orc_compute_differentials.restype = None; L.orc_compute_differentials.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_bump.restype = None; L.orc_bump.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


def material_lobes(scene, material, allow_multiple_lobes=True, uv=(0.0, 0.0), p=(0.0, 0.0, 0.0), duv=(0.0, 0.0, 0.0, 0.0)):
    """Material::compute_scattering_functions of material `material` of a scenes.Scene at a surface point: (Bsdf.eta, Bsdf.bxdfs as
    BXDF_DT records with every parameter texture evaluated)"""
    from rs_pbrt_amd import abi  # layouts only
    surf = np.array(list(p) + list(uv) + list(duv) + [0.0] * 6, np.float32)
    bx = np.zeros(8, abi.BXDF_DT)
    eta = C.c_float(0)
    n = lib().orc_material_lobes(C.addressof(scene.desc), int(material), int(bool(allow_multiple_lobes)), surf.ctypes.data, C.addressof(eta), bx.ctypes.data)
    return float(eta.value), bx[:n].copy()


def tex_eval(scene, tex, uv=(0.0, 0.0), p=(0.0, 0.0, 0.0), duv=(0.0, 0.0, 0.0, 0.0), dpdx=(0.0, 0.0, 0.0), dpdy=(0.0, 0.0, 0.0)):
    """Texture::evaluate of texture `tex` (TexRef or index) at a SurfaceInteraction given by uv, p and its differentials
    (duv = dudx, dvdx, dudy, dvdy)"""
    surf = np.array(list(p) + list(uv) + list(duv) + list(dpdx) + list(dpdy), np.float32)
    out = np.zeros(3, np.float32)
    lib().orc_tex_eval(C.addressof(scene.desc), int(getattr(tex, "index", tex)), surf.ctypes.data, out.ctypes.data)
    return out


def bump(scene, tex, uv=(0.0, 0.0), p=(0.0, 0.0, 0.0), duv=(0.0, 0.0, 0.0, 0.0)):
    surf = np.array(list(p) + list(uv) + list(duv) + [0.0] * 6, np.float32)
    out = np.zeros(6, np.float32)
    lib().orc_bump(C.addressof(scene.desc), int(getattr(tex, "index", tex)), surf.ctypes.data, out.ctypes.data)
    return out[:3], out[3:]


def bvh_build(P, tri, max_prims_in_node=4):
    """BVHAccel::new restated (oracle/orc_scene.hpp).  Returns (nodes NODE_DT[], ordered u32[])."""
    from rs_pbrt_amd import abi  # layouts only
    P = np.ascontiguousarray(P, np.float32); tri = np.ascontiguousarray(tri, np.uint32)
    n = len(tri)
    nodes = np.zeros(max(2 * n, 1), abi.NODE_DT)
    ordered = np.zeros(n, np.uint32)
    k = lib().orc_bvh_build(P.ctypes.data, tri.ctypes.data, n, max_prims_in_node, nodes.ctypes.data, len(nodes), ordered.ctypes.data)
    assert k >= 0
    return nodes[:k].copy(), ordered


def bvh_build_bounds(bounds, max_prims_in_node=4):
    """BVHAccel::new over primitives given by world bounds (n, 6)"""
    from rs_pbrt_amd import abi
    bounds = np.ascontiguousarray(bounds, np.float32).reshape(-1, 6)
    n = len(bounds)
    nodes = np.zeros(max(2 * n, 1), abi.NODE_DT)
    ordered = np.zeros(n, np.uint32)
    k = lib().orc_bvh_build_bounds(bounds.ctypes.data, n, max_prims_in_node, nodes.ctypes.data, len(nodes), ordered.ctypes.data)
    assert k >= 0
    return nodes[:k].copy(), ordered


def transform_bounds(m, lo, hi):
    m = np.ascontiguousarray(m, np.float32).reshape(16); lo = np.ascontiguousarray(lo, np.float32); hi = np.ascontiguousarray(hi, np.float32)
    out = np.zeros(6, np.float32)
    lib().orc_transform_bounds(m.ctypes.data, lo.ctypes.data, hi.ctypes.data, out.ctypes.data)
    return out[:3], out[3:]


def trace(scene, rays, any_hit=False, brute=False, counters=False):
    from rs_pbrt_amd import abi
    rays = np.ascontiguousarray(rays, abi.RAY_DT)
    out = np.zeros(len(rays), abi.HIT_DT)
    cnt = np.zeros(2, np.uint64)
    lib().orc_trace(C.addressof(scene.desc), rays.ctypes.data, len(rays), out.ctypes.data, int(any_hit), int(brute), cnt.ctypes.data)
    return (out, cnt) if counters else out


COUNTER_NAMES = ("nodes_visited", "tris_tested", "rays_closest", "rays_any", "bounces", "samples", "nan_samples", "mis_rays")


def render_integrator(scene, rd, kind, strategy="all", light_samples=None, threads=1, want_li=False):
    """DirectLightingIntegrator ("direct"; strategy "all" = UniformSampleAll | "one" = UniformSampleOne) or
    WhittedIntegrator ("whitted") through the oracle's tile loop (oracle-only: no GPU counterpart yet).
    light_samples: per-light sample counts of UniformSampleAll (Light::get_n_samples), default 1 each."""
    npix = (rd.crop_px[2] - rd.crop_px[0]) * (rd.crop_px[3] - rd.crop_px[1])
    film = np.zeros((npix, 4), np.float32)
    li = np.zeros((npix, int(rd.spp), 3), np.float32) if want_li else None
    cnt = np.zeros(8, np.uint64)
    ns = None if light_samples is None else np.ascontiguousarray(light_samples, np.int32)
    assert ns is None or len(ns) == scene.desc.n_lights
    rc = lib().orc_render_integrator(C.addressof(scene.desc), C.addressof(rd), threads, film.ctypes.data,
                                     li.ctypes.data if want_li else None, cnt.ctypes.data,
                                     {"direct": 2, "whitted": 3}[kind], {"all": 0, "one": 1}[strategy],
                                     None if ns is None else ns.ctypes.data)
    assert rc == 0
    return dict(film=film, li=li, counters=dict(zip(COUNTER_NAMES, (int(x) for x in cnt))))


def render(scene, rd, threads=1, want_li=False):
    """SamplerIntegrator::render restated.  Returns dict(film (npix,4), li (npix,spp,3)|None, counters, seconds)."""
    npix = (rd.crop_px[2] - rd.crop_px[0]) * (rd.crop_px[3] - rd.crop_px[1])
    film = np.zeros((npix, 4), np.float32)
    li = np.zeros((npix, int(rd.spp), 3), np.float32) if want_li else None
    cnt = np.zeros(8, np.uint64)
    sec = C.c_double(0)
    rc = lib().orc_render(C.addressof(scene.desc), C.addressof(rd), threads, film.ctypes.data,
                          li.ctypes.data if want_li else None, cnt.ctypes.data, C.addressof(sec))
    assert rc == 0
    return dict(film=film, li=li, counters=dict(zip(COUNTER_NAMES, (int(x) for x in cnt))), seconds=sec.value)


# ---- GridDensityMedium leaf functions (src/media/grid.rs; oracle only) ----
def _f32(a):
    return np.ascontiguousarray(a, np.float32)


def grid_density(density, pts):
    """GridDensityMedium::density at medium-space points; density[z][y][x]"""
    d = _f32(density); n = np.array(d.shape[::-1], np.int32); p = _f32(pts).reshape(-1, 3); out = np.zeros(len(p), np.float32)
    lib().orc_grid_density(n.ctypes.data, d.ctypes.data, p.ctypes.data, len(p), out.ctypes.data)
    return out


def grid_tr(density, sigma_a, sigma_s, o, d, t_max, u, world_to_medium=None):
    """GridDensityMedium::tr with the 1-D sample stream u: (tr rgb, values used)"""
    dn = _f32(density); n = np.array(dn.shape[::-1], np.int32); w = _f32(np.eye(4) if world_to_medium is None else world_to_medium)
    sa, ss, oo, dd, uu = _f32(np.broadcast_to(sigma_a, 3)), _f32(np.broadcast_to(sigma_s, 3)), _f32(o), _f32(d), _f32(u)
    out = np.zeros(3, np.float32); used = C.c_uint64(0)
    rc = lib().orc_grid_tr(sa.ctypes.data, ss.ctypes.data, n.ctypes.data, w.ctypes.data, dn.ctypes.data, oo.ctypes.data, dd.ctypes.data, float(t_max), uu.ctypes.data, len(uu), out.ctypes.data, C.addressof(used))
    assert rc == 0, "sample stream too short"
    return out, used.value


def grid_sample(density, sigma_a, sigma_s, g, o, d, t_max, u, world_to_medium=None):
    """GridDensityMedium::sample: dict(beta rgb, sampled, p, wo, used)"""
    dn = _f32(density); n = np.array(dn.shape[::-1], np.int32); w = _f32(np.eye(4) if world_to_medium is None else world_to_medium)
    sa, ss, oo, dd, uu = _f32(np.broadcast_to(sigma_a, 3)), _f32(np.broadcast_to(sigma_s, 3)), _f32(o), _f32(d), _f32(u)
    out = np.zeros(10, np.float32); used = C.c_uint64(0)
    rc = lib().orc_grid_sample(sa.ctypes.data, ss.ctypes.data, float(g), n.ctypes.data, w.ctypes.data, dn.ctypes.data, oo.ctypes.data, dd.ctypes.data, float(t_max), uu.ctypes.data, len(uu),
                               out.ctypes.data, C.addressof(used))
    assert rc == 0, "sample stream too short"
    return dict(beta=out[:3].copy(), sampled=bool(out[3]), p=out[4:7].copy(), wo=out[7:10].copy(), used=used.value)


def motion_bounds(start_m, t0, end_m, t1, lo, hi, terms=None):
    """AnimatedTransform::motion_bounds (transform.rs:2147-2210): (lo, hi, terms (5, 3, 4), theta, actually_animated, has_rotation);
    terms: use this coefficient table instead of the oracle's own (orc_motion.hpp)"""
    a = np.ascontiguousarray(start_m, np.float32).reshape(16); b = np.ascontiguousarray(end_m, np.float32).reshape(16)
    l = np.ascontiguousarray(lo, np.float32).reshape(3); h = np.ascontiguousarray(hi, np.float32).reshape(3)
    ti = None if terms is None else np.ascontiguousarray(terms, np.float32).reshape(60)
    ol, oh, to, fl = np.zeros(3, np.float32), np.zeros(3, np.float32), np.zeros(61, np.float32), C.c_int32(0)
    L = lib()
    L.orc_motion_bounds.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_float] + [C.c_void_p] * 7
    rc = L.orc_motion_bounds(a.ctypes.data, float(t0), b.ctypes.data, float(t1), l.ctypes.data, h.ctypes.data, None if ti is None else ti.ctypes.data,
                             ol.ctypes.data, oh.ctypes.data, to.ctypes.data, C.addressof(fl))
    assert rc == 0, "more than eight zeros: the reference panics here"
    return ol, oh, to[:60].reshape(5, 3, 4).copy(), float(to[60]), bool(fl.value & 1), bool(fl.value & 2)


def animated_keys(start_m, t0, end_m, t1):
    """AnimatedTransform::new's decomposition (transform.rs:912-943): t (2, 3), r (2, 4) xyzw, s (2, 4, 4)"""
    a = np.ascontiguousarray(start_m, np.float32).reshape(16); b = np.ascontiguousarray(end_m, np.float32).reshape(16)
    trs = np.zeros(46, np.float32)
    L = lib()
    L.orc_animated_keys.restype = None
    L.orc_animated_keys.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_float, C.c_void_p]
    L.orc_animated_keys(a.ctypes.data, float(t0), b.ctypes.data, float(t1), trs.ctypes.data)
    return trs[:6].reshape(2, 3).copy(), trs[6:14].reshape(2, 4).copy(), trs[14:].reshape(2, 4, 4).copy()


def interpolate_transform(start_m, t0, end_m, t1, time, start_inv=None, end_inv=None, want_inverse=False):
    """AnimatedTransform::interpolate (transform.rs:2081-2113) between two key matrices (4 x 4, row major): the Transform's m (and m_inv)"""
    a = np.ascontiguousarray(start_m, np.float32).reshape(16); b = np.ascontiguousarray(end_m, np.float32).reshape(16)
    ai = np.ascontiguousarray(np.linalg.inv(a.reshape(4, 4).astype(np.float64)) if start_inv is None else start_inv, np.float32).reshape(16)
    bi = np.ascontiguousarray(np.linalg.inv(b.reshape(4, 4).astype(np.float64)) if end_inv is None else end_inv, np.float32).reshape(16)
    m, mi = np.zeros(16, np.float32), np.zeros(16, np.float32)
    lib().orc_interpolate_transform(a.ctypes.data, ai.ctypes.data, float(t0), b.ctypes.data, bi.ctypes.data, float(t1), float(time), m.ctypes.data, mi.ctypes.data)
    return (m.reshape(4, 4), mi.reshape(4, 4)) if want_inverse else m.reshape(4, 4)


def leaf(fn, n, out_shape, a=None, b=None, c=None, d=None, e=None, ia=None, ib=None, words=None):
    """the oracle's restatement of leaf function `fn` (oracle.cpp orc_leaf) over n inputs — what tests/golden/leaf_functions.npz pins by the reference's text"""
    L = lib()
    L.orc_leaf.restype = None
    L.orc_leaf.argtypes = [C.c_int] + [C.c_void_p] * 8 + [C.c_uint64, C.c_void_p]
    keep = [None if x is None else np.ascontiguousarray(x) for x in (a, b, c, d, e, ia, ib, words)]
    out = np.zeros(out_shape, np.float32)
    L.orc_leaf(fn, *[None if x is None else x.ctypes.data for x in keep], n, out.ctypes.data)
    return out


def geom(d):
    """the oracle's restatements of everything tests/golden/geom_functions.npz pins by the reference's text (oracle.cpp orc_geom_*; oracle/make_geom_fixtures.py's
    input dictionary in, its output dictionary out)"""
    L = lib()
    n = len(d["gam_n"])
    keep = []

    def call(fn, ins, shape, pre=(), dtype=np.float32):
        o = np.zeros(shape, dtype)
        f = getattr(L, fn)
        f.restype = None
        arrs = [np.ascontiguousarray(a) for a in ins]
        keep.extend(arrs)
        f.argtypes = [C.c_int] * len(pre) + [C.c_void_p] * len(arrs) + [C.c_uint64, C.c_void_p]
        f(*pre, *[a.ctypes.data for a in arrs], n, o.ctypes.data)
        return o
    z = np.zeros(n, np.float32)
    out = {
        "gam_out": call("orc_geom_scalar", [d["gam_n"], z, z], n, (0,)), "nfu_out": call("orc_geom_scalar", [d["nf_x"], z, z], n, (1,)),
        "nfd_out": call("orc_geom_scalar", [d["nf_x"], z, z], n, (2,)), "ph_out": call("orc_geom_scalar", [d["ph_nf"], d["ph_f"], d["ph_g"]], n, (3,)),
        "rta_out": call("orc_geom_scalar", [d["rta_r"], z, z], n, (5,)), "hg_out": call("orc_geom_scalar", [d["hg_c"], d["hg_g"], z], n, (6,)),
        "y_out": call("orc_geom_scalar", [d["y_rgb"][:, 0], d["y_rgb"][:, 1], d["y_rgb"][:, 2]], n, (7,)),
        "csh_out": call("orc_geom_sample", [d["smp_u"]], (n, 3), (0,)), "ush_out": call("orc_geom_sample", [d["smp_u"]], (n, 3), (1,)),
        "crs_out": call("orc_geom_vec", [d["vec_a"], d["vec_b"]], (n, 3), (0,)), "cs_out": call("orc_geom_vec", [d["vec_a"], d["vec_b"]], (n, 6), (1,)),
        "rfl_out": call("orc_geom_vec", [d["vec_a"], d["vec_b"]], (n, 3), (2,)),
        "rfr_out": call("orc_geom_vec", [d["vec_a"], np.concatenate([np.asarray(d["vec_b"]).reshape(-1), d["rfr_eta"]])], (n, 4), (3,)),
        "adt_out": call("orc_geom_vec", [d["vec_a"], d["vec_b"]], n, (4,)),
        "oro_out": call("orc_geom_offset_ray_origin", [d["oro_p"], d["oro_e"], d["oro_n"], d["oro_w"]], (n, 3)),
        "box_out": call("orc_geom_box", [d["box_b"], d["box_o"], d["box_inv"], d["box_neg"], d["box_tmax"]], n),
        "tri_out": call("orc_geom_triangle", [d["tri_p"], d["tri_o"], d["tri_d"], d["tri_tmax"]], (n, 5)),
        "mf_out": call("orc_geom_microfacet", [d["mf_wo"], d["mf_wh"], d["mf_ax"], d["mf_ay"]], (n, 5)),
        "trf_out": call("orc_geom_triangle_full", [d["tri_p"], d["trf_n"], d["trf_s"], d["trf_uv"], d["trf_flags"], d["tri_o"], d["tri_d"], d["tri_tmax"]], (n, 48)),
        "dif_out": call("orc_geom_differentials", [d["dif_x"]], (n, 10)),
        "al_out": call("orc_geom_area_light", [d["al_tri"], d["al_nrm"], d["al_flags"], d["al_L"], d["al_ref"], d["al_u"]], (n, 16)),
    }
    out["mor_out"] = call("orc_geom_morton", [np.ascontiguousarray(d["mor_xy"], np.uint32)], n, dtype=np.uint32)
    nf = len(d["flm_geo"])
    L.orc_geom_film.restype = None
    L.orc_geom_film.argtypes = [C.c_void_p] * 3 + [C.c_uint64, C.c_void_p]
    fo = np.zeros((nf, 256, 4), np.float32)
    fg, ff, fs = np.ascontiguousarray(d["flm_geo"], np.int32), np.ascontiguousarray(d["flm_flt"], np.float32), np.ascontiguousarray(d["flm_smp"], np.float32)
    L.orc_geom_film(fg.ctypes.data, ff.ctypes.data, fs.ctypes.data, nf, fo.ctypes.data)
    out["flm_out"] = fo
    from rs_pbrt_amd import scenes
    tables = scenes.sobol_tables().as_struct(None)
    L.orc_geom_sobol.restype = None
    L.orc_geom_sobol.argtypes = [C.c_void_p] * 4 + [C.c_uint64, C.c_void_p]
    ns = len(d["sob_spp"])
    so = np.zeros((ns, 4, 26), np.float32)
    spp, bounds, pixel = np.ascontiguousarray(d["sob_spp"], np.int64), np.ascontiguousarray(d["sob_bounds"], np.int32), np.ascontiguousarray(d["sob_pixel"], np.int32)
    L.orc_geom_sobol(C.addressof(tables), spp.ctypes.data, bounds.ctypes.data, pixel.ctypes.data, ns, so.ctypes.data)
    out["sob_out"] = so
    # the Halton sampler: the digit permutations (all 1000 bases), the radical inverses, the sample stream
    n_p = L.orc_halton_permutations(1000, None)
    perms = np.zeros(n_p, np.uint16)
    L.orc_halton_permutations(1000, perms.ctypes.data)
    import hashlib
    out["hpc_out"] = np.array([n_p], np.uint64); out["hph_out"] = perms[:8192].copy()
    out["hps_out"] = np.frombuffer(hashlib.sha256(perms.astype("<u2").tobytes()).digest(), np.uint64).copy()
    nr = len(d["rad_bi"])
    ro, ri = np.zeros((nr, 2), np.float32), np.zeros((nr, 4), np.uint64)
    bi, aa = np.ascontiguousarray(d["rad_bi"], np.uint16), np.ascontiguousarray(d["rad_a"], np.uint64)
    L.orc_geom_radical.restype = None
    L.orc_geom_radical.argtypes = [C.c_void_p] * 3 + [C.c_uint64, C.c_void_p, C.c_void_p]
    L.orc_geom_radical(perms.ctypes.data, bi.ctypes.data, aa.ctypes.data, nr, ro.ctypes.data, ri.ctypes.data)
    out["rad_out"], out["radi_out"] = ro, ri
    nh = len(d["hal_spp"])
    ho, hm = np.zeros((nh, 4, 34), np.float32), np.zeros((nh, 12), np.uint64)
    hs = [np.ascontiguousarray(d["hal_spp"], np.int64), np.ascontiguousarray(d["hal_bounds"], np.int32), np.ascontiguousarray(d["hal_pixel"], np.int32),
          np.ascontiguousarray(d["hal_center"], np.uint8), np.ascontiguousarray(d["hal_arrays"], np.int32)]
    L.orc_geom_halton.restype = None
    L.orc_geom_halton.argtypes = [C.c_void_p, C.c_uint64] + [C.c_void_p] * 5 + [C.c_uint64, C.c_void_p, C.c_void_p]
    L.orc_geom_halton(perms.ctypes.data, n_p, *[x.ctypes.data for x in hs], nh, ho.ctypes.data, hm.ctypes.data)
    out["hal_out"], out["halm_out"] = ho, hm
    npx = len(d["pix_kind"])
    po, pm = np.zeros((npx, 4, 34), np.float32), np.zeros((npx, 2), np.uint64)
    rows = np.ascontiguousarray(scenes.maxmin_tables(), np.uint32)
    ps = [np.ascontiguousarray(d["pix_kind"], np.int32), np.ascontiguousarray(d["pix_par"], np.int64), np.ascontiguousarray(d["pix_seed"], np.uint64),
          np.ascontiguousarray(d["pix_pixel"], np.int32), np.ascontiguousarray(d["pix_arrays"], np.int32)]
    L.orc_geom_pixel.restype = None
    L.orc_geom_pixel.argtypes = [C.c_void_p] * 6 + [C.c_uint64, C.c_void_p, C.c_void_p]
    L.orc_geom_pixel(rows.ctypes.data, *[x.ctypes.data for x in ps], npx, po.ctypes.data, pm.ctypes.data)
    out["pix_out"], out["pixm_out"] = po, pm
    out["trp_out"] = out["tri_out"]      # Triangle::intersect_p repeats intersect's watertight test (triangle.rs:450-591); the oracle shares one function
    ou, of = np.zeros((n, 6), np.uint32), np.zeros((n, 2), np.float32)
    L.orc_geom_rng.restype = None
    L.orc_geom_rng.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    seq, bound = np.ascontiguousarray(d["rng_seq"], np.uint64), np.ascontiguousarray(d["rng_bound"], np.uint32)
    L.orc_geom_rng(seq.ctypes.data, bound.ctypes.data, n, ou.ctypes.data, of.ctypes.data)
    out["rng_u_out"], out["rng_f_out"] = ou, of
    return out
