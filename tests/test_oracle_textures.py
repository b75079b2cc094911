"""CPU: known-answer tests that pin the oracle's texture restatement and the host-side image
pyramid builder against closed forms of the reference's formulas (mipmap.rs, texture.rs, interaction.rs:388-479,
material.rs:116-219, perspective.rs:190-280).  The reference has no tests or fixtures for these either."""
import ctypes as C
import math

import numpy as np
import pytest

from rs_pbrt_amd import abi, scenes

F32 = np.float32


def _scene(sb, oracle):
    m = sb.add_material(scenes.matte((0.5, 0.5, 0.5)))
    sb.add_quad([(-1, 0, -1), (1, 0, -1), (1, 0, 1), (-1, 0, 1)], m)
    return sb.finish(oracle.bvh_build)


def test_build_image_pow2_passthrough_and_pyramid():
    rng = np.random.default_rng(1)
    img = rng.random((8, 16, 3)).astype(F32)
    im = scenes.build_image(img)
    assert (im["width"], im["height"], im["n_levels"], im["channels"]) == (16, 8, 5, 3)
    lv0 = im["texels"][: 16 * 8 * 3].reshape(8, 16, 3)
    assert np.array_equal(lv0, img[::-1])  # rows flipped: t = 0 is the bottom row of the file (imagemap.rs:63-71)
    # every level is the box filter of the one below (mipmap.rs:166-183); the last one is the mean
    sizes = [(8 >> l or 1, 16 >> l or 1) for l in range(5)]
    off = np.cumsum([0] + [h * w * 3 for h, w in sizes])
    assert off[-1] == len(im["texels"])
    top = im["texels"][off[4]:off[5]]
    assert np.allclose(top, img.mean(axis=(0, 1)), rtol=1e-5)
    lv1 = im["texels"][off[1]:off[2]].reshape(4, 8, 3)
    want = ((lv0[0::2, 0::2] + lv0[0::2, 1::2]).astype(F32) + lv0[1::2, 0::2]).astype(F32) + lv0[1::2, 1::2]
    assert np.array_equal(lv1, (want.astype(F32) * F32(0.25)).astype(F32))


def test_build_image_resamples_to_pow2_like_mipmap_new():
    img = np.full((5, 6, 3), 0.25, F32)
    im = scenes.build_image(img)  # 6x5 -> 8x8 with normalised 4-tap Lanczos weights (mipmap.rs:64-148, 298-322)
    assert (im["width"], im["height"]) == (8, 8)
    assert np.allclose(im["texels"], 0.25, atol=1e-6)
    first, wt = scenes._resample_weights(6, 8)
    assert np.allclose(wt.sum(1), 1.0, atol=1e-6) and wt.shape == (8, 4)
    # first_texel = floor(center - 2 + 0.5), center = (i + 0.5) * 6 / 8
    assert list(first) == [math.floor((i + 0.5) * 6 / 8 - 2 + 0.5) for i in range(8)]
    # black wrap drops the out-of-range taps (no renormalisation): border texels darken; clamp keeps the constant
    assert scenes.build_image(img, wrap=abi.WRAP_BLACK)["texels"][: 8 * 8 * 3].min() < 0.24
    assert np.allclose(scenes.build_image(img, wrap=abi.WRAP_CLAMP)["texels"], 0.25, atol=1e-6)
    # float textures keep y() (convert_to_float), gamma / scale applied first (imagemap.rs:74-86)
    g = scenes.build_image(np.full((4, 4, 3), 0.5, F32), scale=2.0, gamma=True, channels=1)
    lin = ((0.5 + 0.055) / 1.055) ** 2.4
    assert g["channels"] == 1 and np.allclose(g["texels"], 2.0 * lin * (0.212671 + 0.715160 + 0.072169), rtol=1e-5)


def test_constant_scale_and_bilinear_lookup(oracle):
    sb = scenes.SceneBuilder()
    img = np.zeros((2, 2, 3), F32)
    img[0, 0] = (1, 0, 0); img[0, 1] = (0, 1, 0); img[1, 0] = (0, 0, 1); img[1, 1] = (1, 1, 1)  # file rows: top first
    t_img = sb.image_texture(img, wrap="clamp")
    t_c = sb.constant_texture((0.5, 0.25, 2.0))
    t_s = sb.scale_texture(t_img, t_c)
    t_uv = sb.image_texture(img, su=2.0, sv=1.0, du=0.25, dv=0.0, wrap="repeat")
    sc = _scene(sb, oracle)
    assert np.array_equal(oracle.tex_eval(sc, t_c), np.array([0.5, 0.25, 2.0], F32))
    # no differentials -> MipMap::triangle on level 0 (both filters): texel centres reproduce the texels (t = 0 is the bottom row)
    assert np.array_equal(oracle.tex_eval(sc, t_img, uv=(0.25, 0.25)), np.array([0, 0, 1], F32))
    assert np.array_equal(oracle.tex_eval(sc, t_img, uv=(0.75, 0.75)), np.array([0, 1, 0], F32))
    assert np.allclose(oracle.tex_eval(sc, t_img, uv=(0.5, 0.5)), np.array([0.5, 0.5, 0.5], F32))
    assert np.allclose(oracle.tex_eval(sc, t_img, uv=(0.5, 0.25)), np.array([0.5, 0.5, 1.0], F32))  # between blue and white
    assert np.array_equal(oracle.tex_eval(sc, t_img, uv=(-3.0, 0.25)), np.array([0, 0, 1], F32))   # clamp
    assert np.allclose(oracle.tex_eval(sc, t_s, uv=(0.75, 0.75)), np.array([0, 0.25, 0], F32))     # ScaleTexture: product
    # UVMapping2D: st = (su * u + du, sv * v + dv) (texture.rs:113-117); repeat wraps
    assert np.array_equal(oracle.tex_eval(sc, t_uv, uv=(0.5, 0.25)), oracle.tex_eval(sc, t_img, uv=(0.25, 0.25)))  # s = 1.25 -> 0.25


def test_trilinear_level_selection_and_ewa(oracle):
    sb = scenes.SceneBuilder()
    img = np.zeros((4, 4, 3), F32); img[:, ::2] = 1.0  # vertical stripes: level 0 alternates, level >= 1 is 0.5
    t_tri = sb.image_texture(img, trilinear=True)
    t_ewa = sb.image_texture(img, trilinear=False)
    const = sb.image_texture(np.full((16, 16, 3), 0.3, F32), trilinear=False)
    sc = _scene(sb, oracle)
    uv = (0.125, 0.125)  # centre of texel (0, 0): stripe value 1
    assert oracle.tex_eval(sc, t_tri, uv=uv)[0] == 1.0
    # width = max |d(st)| ; level = n_levels - 1 + log2(width) (mipmap.rs:236): width 1/4 -> level 0, 1/2 -> 1, in between lerps
    assert oracle.tex_eval(sc, t_tri, uv=uv, duv=(0.25, 0, 0, 0))[0] == 1.0
    assert oracle.tex_eval(sc, t_tri, uv=uv, duv=(0.5, 0, 0, 0))[0] == 0.5
    mid = oracle.tex_eval(sc, t_tri, uv=uv, duv=(2 ** -1.5, 0, 0, 0))[0]
    assert abs(mid - 0.75) < 1e-6
    assert oracle.tex_eval(sc, t_tri, uv=uv, duv=(4.0, 0, 0, 0))[0] == 0.5  # beyond the top level: its single texel
    # EWA: a constant image stays constant for any footprint (weights normalised, mipmap.rs:396); zero minor axis -> bilinear
    for duv in ((0.1, 0.0, 0.0, 0.05), (0.3, 0.1, -0.05, 0.2), (1e-3, 0, 0, 1e-4)):
        assert np.allclose(oracle.tex_eval(sc, const, uv=(0.37, 0.61), duv=duv), 0.3, atol=1e-6)
    assert oracle.tex_eval(sc, t_ewa, uv=uv, duv=(0.25, 0, 0, 0))[0] == 1.0
    # an isotropic footprint of two texels on the stripes averages towards 0.5
    v = oracle.tex_eval(sc, t_ewa, uv=(0.5, 0.5), duv=(0.5, 0, 0, 0.5))[0]
    assert 0.4 < v < 0.6


def test_planar_mapping(oracle):
    sb = scenes.SceneBuilder()
    img = np.zeros((2, 2, 3), F32); img[1, 0] = (0, 0, 1); img[0, 1] = (0, 1, 0)
    t = sb.image_texture(img, mapping="planar", v1=(0.5, 0, 0), v2=(0, 0, 0.5), du=0.25, dv=0.25, wrap="clamp")
    sc = _scene(sb, oracle)
    # st = (ds + p . vs, dt + p . vt) (texture.rs:252-256): p = (0, 7, 0) -> (0.25, 0.25) = the bottom-left texel
    assert np.array_equal(oracle.tex_eval(sc, t, p=(0.0, 7.0, 0.0)), np.array([0, 0, 1], F32))
    assert np.array_equal(oracle.tex_eval(sc, t, p=(1.0, -2.0, 1.0)), np.array([0, 1, 0], F32))


def test_camera_differentials_and_compute_differentials(oracle):
    L = oracle.lib()
    rd = scenes.make_render_desc(64, 64, 4, ((0, 0, -5), (0, 0, 0), (0, 1, 0)), 40.0)
    cs = np.array([32.0, 32.0, 0.0, 0.5, 0.5], F32)
    out = np.zeros(18, F32)
    L.orc_camera_ray_diff(C.addressof(rd), cs.ctypes.data, out.ctypes.data)
    o, d, rxo, rxd, ryo, ryd = out.reshape(6, 3)
    # pinhole: offset origins coincide with the camera position up to the ray-origin error offset; directions differ from d
    # by one pixel's angle scaled by 1 / sqrt(spp) (integrator.rs:140-144)
    assert np.allclose(rxo, o, atol=1e-5) and np.allclose(ryo, o, atol=1e-5)
    px_angle = 2 * math.tan(math.radians(20.0)) / 64
    assert abs(np.linalg.norm(rxd - d) - px_angle / 2) < 2e-4 and abs(np.linalg.norm(ryd - d) - px_angle / 2) < 2e-4
    # a plane z = 0 facing the camera with dpdu = +x, dpdv = +y: dp/dx = t * d(direction), du/dx = dpdx.x (interaction.rs:388-479)
    hit = np.concatenate([[0, 0, 0], [0, 0, -1], [1, 0, 0], [0, 1, 0], out]).astype(F32)
    dd = np.zeros(10, F32)
    L.orc_compute_differentials(hit.ctypes.data, dd.ctypes.data)
    dudx, dvdx, dudy, dvdy = dd[:4]
    assert abs(abs(dudx) - 5.0 * px_angle / 2) < 2e-3 and abs(dvdx) < 1e-4
    assert abs(abs(dvdy) - 5.0 * px_angle / 2) < 2e-3 and abs(dudy) < 1e-4
    assert np.allclose(dd[4:7], [dudx, dvdx, 0], atol=1e-6) and np.allclose(dd[7:10], [dudy, dvdy, 0], atol=1e-6)
    # grazing plane (normal perpendicular to the ray): t is inf / nan -> all differentials zero (interaction.rs:404-411)
    hit[3:6] = (1, 0, 0)
    hit[12 + 3:12 + 6] = (0, 0, 1); hit[12 + 9:12 + 12] = (0, 0, 1); hit[12 + 15:12 + 18] = (0, 0, 1)
    L.orc_compute_differentials(hit.ctypes.data, dd.ctypes.data)
    assert not dd.any()


def test_bump_mapping(oracle):
    sb = scenes.SceneBuilder()
    flat = sb.constant_texture(0.3)
    # height = 0.5 * u: a ramp along u.  16 texels wide, sampled with bilinear interpolation away from the borders
    ramp_img = np.repeat((np.arange(16, dtype=F32) + F32(0.5))[None, :, None] / F32(16), 16, 0).repeat(3, 2) * F32(0.5)
    ramp = sb.image_texture(ramp_img.astype(F32), channels=1, wrap="clamp", trilinear=True)
    sc = _scene(sb, oracle)
    n, dpdu = oracle.bump(sc, flat, uv=(0.5, 0.5))
    assert np.array_equal(n, np.array([0, 0, 1], F32)) and np.array_equal(dpdu, np.array([1, 0, 0], F32))  # no gradient: untouched
    # ramp: dpdu' = dpdu + n * d(height)/du = (1, 0, 0.5) (material.rs:205-207), normal = normalize(cross(dpdu', dpdv)) faced to n
    n, dpdu = oracle.bump(sc, ramp, uv=(0.5, 0.5))
    assert np.allclose(dpdu, [1, 0, 0.5], atol=2e-3)
    want = np.cross([1, 0, 0.5], [0, 1, 0]); want /= np.linalg.norm(want)
    assert np.allclose(n, want, atol=2e-3) and n[2] > 0
    # du = 0.5 * (|dudx| + |dudy|) when differentials exist (material.rs:174), else 0.0005: same slope either way on a ramp
    n2, dpdu2 = oracle.bump(sc, ramp, uv=(0.5, 0.5), duv=(0.01, 0, 0, 0.01))
    assert np.allclose(dpdu2, [1, 0, 0.5], atol=2e-3)


def test_textured_lobe_is_dropped_when_black(oracle):
    """matte.rs:70 `if !r.is_black()`: over the black half of the texture the material has no lobe, so a path ends
    there without next-event estimation; over the white half it is an ordinary matte"""
    sb = scenes.SceneBuilder()
    img = np.zeros((2, 2, 3), F32); img[:, 1] = 1.0  # left half black, right half white
    kd = sb.image_texture(img, wrap="clamp")
    m = sb.add_material(scenes.matte(kd))
    white = sb.add_material(scenes.matte((0.7, 0.7, 0.7)))
    sb.add_quad([(-1, 0, -1), (1, 0, -1), (1, 0, 1), (-1, 0, 1)], m, UV=[[0, 0], [1, 0], [1, 1], [0, 1]])
    sb.add_quad([(-0.5, 2, -0.5), (0.5, 2, -0.5), (0.5, 2, 0.5), (-0.5, 2, 0.5)], white, emit=(10, 10, 10), two_sided=True)
    sc = sb.finish(oracle.bvh_build)
    rd = scenes.make_render_desc(32, 32, 16, ((0, 1.5, 0), (0, 0, 0), (0, 0, 1)), 60.0, max_depth=2)  # below the light, looking down
    r = oracle.render(sc, rd, threads=4, want_li=True)
    rgb = scenes.film_to_rgb(r["film"]).reshape(32, 32, 3)
    assert r["counters"]["nan_samples"] == 0
    # outer image columns: u < 0.25 (pure black texel under clamp) vs u > 0.75 (pure white)
    a, b = rgb[:, :5].mean(), rgb[:, -5:].mean()
    assert min(a, b) == 0.0 and max(a, b) > 0.05


def test_procedural_textures_closed_forms(oracle):
    """checkerboard (checkerboard.rs: floor-parity, no anti-aliasing), mix (mix.rs), Perlin noise properties
    (texture.rs:294-364: zero on the integer lattice, within [-1, 1]), fbm / turbulence octave logic (366-424),
    dots (dots.rs), marble range, spherical / cylindrical mappings (texture.rs:123-220)"""
    sb = scenes.SceneBuilder()
    a, b = sb.constant_texture((0.8, 0.1, 0.1)), sb.constant_texture((0.1, 0.1, 0.8))
    ck = sb.checkerboard_texture(a, b, su=4.0, sv=4.0)
    mx = sb.mix_texture(a, b, 0.25)
    fb, wr, wi, mb = sb.fbm_texture(), sb.wrinkled_texture(octaves=4), sb.windy_texture(), sb.marble_texture(scale=2.0)
    fb1 = sb.fbm_texture(octaves=1)
    dt = sb.dots_texture(a, b, su=6, sv=6)
    img = np.zeros((2, 4, 3), F32); img[:, :, 0] = np.arange(4)[None, :] / 4.0  # red = column index / 4
    sph = sb.image_texture(img, mapping="spherical", wrap="clamp", trilinear=True)
    cyl = sb.image_texture(img, mapping="cylindrical", wrap="clamp", trilinear=True)
    sc = _scene(sb, oracle)
    red, blue = np.array([0.8, 0.1, 0.1], F32), np.array([0.1, 0.1, 0.8], F32)
    for u, v, want in ((0.1, 0.1, red), (0.3, 0.1, blue), (0.3, 0.3, red), (0.6, 0.3, blue), (0.99, 0.99, red)):
        assert np.array_equal(oracle.tex_eval(sc, ck, uv=(u, v)), want)
    assert np.allclose(oracle.tex_eval(sc, mx), 0.75 * red + 0.25 * blue, atol=1e-7)
    # noise: zero on the lattice (all gradients dotted with zero offsets), so fbm with any octave count is zero there
    assert oracle.tex_eval(sc, fb1, p=(3.0, -2.0, 7.0), dpdx=(0.3, 0, 0), dpdy=(0, 0.3, 0))[0] == 0.0
    rng = np.random.default_rng(3)
    vals = np.array([oracle.tex_eval(sc, fb1, p=tuple(rng.uniform(-5, 5, 3)), dpdx=(0.3, 0, 0), dpdy=(0, 0.3, 0))[0] for _ in range(300)])
    assert np.abs(vals).max() <= 1.0 and vals.std() > 0.05 and abs(vals.mean()) < 0.1
    # octaves: n = clamp(-1 - log2(|dp|^2) / 2, 0, max); |dp| = 1/2 gives n = 0: only the partial octave with smooth_step(0) = 0
    assert oracle.tex_eval(sc, fb, p=(0.3, 0.4, 0.5), dpdx=(0.5, 0, 0), dpdy=(0, 0.5, 0))[0] == 0.0
    # turbulence adds 0.2 * omega^i for the octaves beyond n (texture.rs:418-422): with n = 0 that is 0.2 * (1 + .5 + .25 + .125) + 0.2
    w0 = oracle.tex_eval(sc, wr, p=(0.3, 0.4, 0.5), dpdx=(0.5, 0, 0), dpdy=(0, 0.5, 0))[0]
    assert abs(w0 - (0.2 + 0.2 * (1 + 0.5 + 0.25 + 0.125))) < 1e-6
    # no differentials: log2(0) = -inf -> all octaves; windy = |fbm(p / 10, 3 octaves)| * fbm(p, 6 octaves)
    p = (0.37, 1.71, -2.2)
    assert abs(oracle.tex_eval(sc, wi, p=p)[0]) < 1.0
    m = oracle.tex_eval(sc, mb, p=p)
    assert (m > 0.25).all() and (m < 1.0).all()  # Bezier over control points in [0.2, 0.6], times 1.5
    inside = [oracle.tex_eval(sc, dt, uv=(u, 0.33))[0] for u in np.linspace(0, 1, 200)]
    assert 0.02 < np.mean(np.array(inside) < 0.5) < 0.9  # some samples fall inside dots (blue = inside -> red channel 0.1)
    # spherical: st = (theta / pi, phi / 2 pi); p on +x: theta = pi / 2, phi = 0 -> s = 0.5 -> between columns 1 and 2
    v = oracle.tex_eval(sc, sph, p=(2.0, 0.0, 0.0))[0]
    assert abs(v - 0.375) < 1e-6
    # cylindrical: s = pi + atan2(y, x) / 2 pi (texture.rs:188: as written), far beyond 1 -> clamp to the last column
    assert oracle.tex_eval(sc, cyl, p=(2.0, 0.0, 0.3))[0] == 0.75


def test_texture_graph_depth_and_cycles_rejected():
    """the device evaluates graphs up to three levels; deeper ones and cycles must be refused by the host (checked on the
    GPU in test_texture_validation); here: the builder produces the depths we think"""
    sb = scenes.SceneBuilder()
    a, b = sb.constant_texture(0.2), sb.constant_texture(0.7)
    l2 = sb.scale_texture(a, b)
    l3 = sb.mix_texture(l2, a, 0.5)
    assert [int(t["kind"]) for t in sb.textures][-1] == abi.TEX_MIX and sb.textures[l3.index]["tex1"] == l2.index
