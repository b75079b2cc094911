#!/usr/bin/env python3
"""How rs_pbrt_amd/scenes.py `cornell_box_docs` was recovered from the reference's two documentation renders
(tests/golden/reference_cornell_docs.npz = docs/source/cornell_box_{8,256}_pixelsamples.png; the scene file is not in the reference tree).
Every step is a search over scene descriptions whose objective is agreement of the ORACLE's render with those images; nothing here touches
the product.  Test infrastructure (it drives oracle/), minutes of CPU per step:

    python tools/recover_cornell_docs.py [fov] [fit] [handedness] [frames] [light] [corners]      (no argument: all of them)

Steps and what they print (the values frozen in scenes.py are the ones a run of this script ends on):
  fov         sub-pixel frame edges of the 256-spp image -> field of view
  fit         least squares (L, wall / red / green / block albedos) on the 2 x 2 box-filtered 256-spp image
  handedness  8-spp noise correlation with the camera mirrored vs. the world mirrored
  frames      per visible triangle: diagonal x rotation x winding by noise correlation over its footprint
  light       the 36 downward-facing triangulations of the emitter by the log-ratio spread on the directly lit faces
  corners     half-unit scans of the block corners around their silhouette edges (byte differences at 8 spp)
"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from rs_pbrt_amd import abi, scenes  # noqa: E402
from rs_pbrt_amd.scenes import F32, SceneBuilder, matte  # noqa: E402
from oracle import pyoracle as oracle  # noqa: E402

THREADS = os.cpu_count() or 8
G = np.load(os.path.join(ROOT, "tests", "golden", "reference_cornell_docs.npz"))
U8 = {8: G["spp8"].astype(np.int32), 256: G["spp256"].astype(np.int32)}


def to_linear(u8):
    a = np.asarray(u8, np.float32) / 255.0
    return np.where(a <= 0.04045, a / 12.92, ((a + 0.055) / 1.055) ** 2.4)   # inverse of Film::write_image's gamma_correct (film.rs:465-520)


def to_u8(rgb):
    a = np.asarray(rgb, np.float32)
    a = np.where(a <= 0.0031308, 12.92 * a, 1.055 * np.power(np.maximum(a, 0), np.float32(1 / 2.4)) - 0.055)
    return np.clip(255.0 * a + 0.5, 0, 255).astype(np.int32)


R8, R256 = to_linear(U8[8]), to_linear(U8[256])
NAMES = ["floor", "ceil", "back", "green", "red", "s_top", "s_1", "s_2", "s_3", "s_4", "t_top", "t_1", "t_2", "t_3", "t_4", "light"]


def public_quads(short=None, tall=None):
    """the public Cornell data, quad by quad (floor, ceiling, back, green, red, short block top + 4 sides, tall block top + 4 sides)"""
    room = [q for q, _, _ in scenes._DOCS_ROOM]
    out = list(room)
    for (a, b, c, d), h, tall_order in ((short or [(130, 65), (82, 225), (240, 272), (290, 114)], 165.0, False),
                                         (tall or [(423, 247), (265, 296), (314, 456), (472, 406)], 330.0, True)):
        side = lambda u, v: [(u[0], 0, u[1]), (u[0], h, u[1]), (v[0], h, v[1]), (v[0], 0, v[1])]  # noqa: E731
        order = ((a, d), (d, c), (c, b), (b, a)) if tall_order else ((d, c), (a, d), (b, a), (c, b))
        out += [[(p[0], h, p[1]) for p in (a, b, c, d)]] + [side(u, v) for u, v in order]
    return out


def build(tris, light_tris, mirror_world=True, L=(100.0,) * 3, white=(0.4,) * 3, red=0.5, green=0.5, block=0.5, quads=None, light_y=548.0):
    """tris[i]: the two index triples of quad i"""
    sb = SceneBuilder()
    m = [sb.add_material(matte(tuple(white)))] * 3 + [sb.add_material(matte((0, green, 0))), sb.add_material(matte((red, 0, 0)))] + [sb.add_material(matte((block,) * 3))] * 10
    sx = -1.0 if mirror_world else 1.0
    for q, t, mat in zip(quads or public_quads(), tris, m):
        sb.add_mesh(np.array([(sx * p[0], p[1], p[2]) for p in q], F32), t, mat)
    sb.add_mesh(np.array([(sx * p[0], light_y, p[2]) for p in scenes._DOCS_LIGHT], F32), light_tris, m[0], emit=tuple(L))
    return sb.finish(oracle.bvh_build)


def render(sc, spp, res=500, mirror_camera=False, fov=scenes.CORNELL_DOCS_FOV, **kw):
    rd = scenes.cornell_docs_render_desc(spp, res, mirror_camera=mirror_camera, **kw)
    if fov != scenes.CORNELL_DOCS_FOV:
        rd2 = scenes.make_render_desc(res, res, spp, scenes.CORNELL_DOCS_LOOK_AT, fov, **kw)
        rd.raster_to_camera[:] = rd2.raster_to_camera[:]
    return scenes.film_to_rgb(oracle.render(sc, rd, threads=THREADS)["film"]).reshape(res, res, 3)


def tri_options(diag, rot_a, wind_a, rot_b, wind_b):
    base = [[0, 1, 2], [0, 2, 3]] if diag == 0 else [[1, 2, 3], [1, 3, 0]]
    out = []
    for t, rot, wind in zip(base, (rot_a, rot_b), (wind_a, wind_b)):
        t = t[rot:] + t[:rot]
        out.append(t[::-1] if wind else t)
    return out


FROZEN_TRIS = [scenes._docs_fan(k) for k in [k for _, _, k in scenes._DOCS_ROOM] + list(scenes._DOCS_SHORT[2]) + list(scenes._DOCS_TALL[2])]
FROZEN_LIGHT = scenes._docs_fan(3, rev=True)
FROZEN_QUADS = public_quads(scenes._DOCS_SHORT[0], scenes._DOCS_TALL[0])


def byte_stats(img, window=None):
    d = np.abs(to_u8(img) - U8[8])
    if window is not None:
        r0, r1, c0, c1 = window
        return float(d[r0:r1, c0:c1].mean())
    dm = d.max(-1)
    return dict(exact=round(float((dm == 0).mean()), 4), within1=round(float((dm <= 1).mean()), 4), within4=round(float((dm <= 4).mean()), 4), mean_abs=round(float(d.mean()), 3))


def footprints(diag, quads=None):
    """triangle 2 * quad + (0 | 1) seen through every pixel centre (-1: nothing), by brute force in numpy"""
    t = np.tan(np.radians(scenes.CORNELL_DOCS_FOV) / 2)
    px = (np.arange(500) + 0.5) / 500 * 2 - 1
    X, Y = np.meshgrid(px, -px)
    d = np.stack([X * t, Y * t, np.ones_like(X)], -1).reshape(-1, 3)
    o = np.array(scenes.CORNELL_DOCS_LOOK_AT[0], float)
    best_t = np.full(len(d), np.inf); best = np.full(len(d), -1)
    for qi, q in enumerate((quads or public_quads()) + [scenes._DOCS_LIGHT]):
        Q = np.array([(-v[0], v[1], v[2]) for v in q], float)
        for ti, tr in enumerate([[0, 1, 2], [0, 2, 3]] if diag == 0 else [[1, 2, 3], [1, 3, 0]]):
            p0, e1, e2 = Q[tr[0]], Q[tr[1]] - Q[tr[0]], Q[tr[2]] - Q[tr[0]]
            pv = np.cross(d, e2); inv = 1.0 / (pv @ e1)
            tv = o - p0; u = (pv @ tv) * inv
            qv = np.cross(tv, e1); v = (d @ qv) * inv; tt = (qv @ e2) * inv
            hit = (u >= 0) & (v >= 0) & (u + v <= 1) & (tt > 0) & (tt < best_t)
            best_t[hit] = tt[hit]; best[hit] = 2 * qi + ti
    return best.reshape(500, 500)


def noise_corr(img, mask, channel=None):
    a = (R8 - R256)[mask]; b = (np.minimum(img, 1) - R256)[mask]
    c = int(np.argmax(np.abs(a).mean(0))) if channel is None else channel
    return float(np.corrcoef(a[:, c], b[:, c])[0, 1])


# ---- steps ----
def step_fov():
    print("== fov from the frame: the room's opening is x in [0, 556] at z = 0, the camera 800 in front of it at x = 278")
    rows = np.arange(60, 440, 20)
    est = []
    for y in rows:   # right edge of the picture = x = 0 (the green wall's front edge is vertical)
        row = R256[y, :, 1]
        c = np.max(np.nonzero(row > 0.02)[0]); inside = row[c - 3:c - 1].mean()
        edge = c + row[c] / inside if row[c] < 0.8 * inside else c + 1.0     # column where the wall ends (coverage of the last pixel)
        est.append(2 * np.degrees(np.arctan(278 * 250 / (800 * (edge - 250)))))
    print("   per row:", np.round(est, 3)); print("   median %.3f degrees (scenes.CORNELL_DOCS_FOV = %.5f)" % (np.median(est), scenes.CORNELL_DOCS_FOV))


def step_fit():
    from scipy.optimize import least_squares
    print("== least squares on the 256-spp image (2 x 2 box filtered, unsaturated pixels): L, walls, red, green, blocks")
    ref = R256.reshape(250, 2, 250, 2, 3).mean(axis=(1, 3)); mask = R256.reshape(250, 2, 250, 2, 3).max(axis=(1, 3)) < 0.9

    def resid(p):
        sc = build(FROZEN_TRIS, FROZEN_LIGHT, L=(p[0],) * 3, white=(p[1],) * 3, red=p[2], green=p[3], block=p[4], quads=FROZEN_QUADS)
        return ((np.minimum(render(sc, 64, res=250), 1.0) - ref) * mask).reshape(-1)
    r = least_squares(resid, np.array([60.0, 0.6, 0.6, 0.6, 0.6]), diff_step=0.02, bounds=([1] + [0.05] * 4, [500] + [0.999] * 4), max_nfev=14, x_scale=[50, 1, 1, 1, 1])
    print("   L %.1f  walls %.3f  red %.3f  green %.3f  blocks %.3f   rmse %.4f" % (*r.x, np.sqrt((r.fun ** 2).sum() / mask.sum())))
    print("   rmse at (100, 0.4, 0.5, 0.5, 0.5): %.4f" % np.sqrt((resid([100, 0.4, 0.5, 0.5, 0.5]) ** 2).sum() / mask.sum()))


REGIONS = dict(floor=(420, 480, 100, 400), ceil=(20, 90, 120, 380), back=(120, 200, 150, 350), red=(150, 400, 20, 90), green=(150, 400, 410, 480))


def region_scores(img):
    out = {}
    for n, (y0, y1, x0, x1) in REGIONS.items():
        m = np.zeros((500, 500), bool); m[y0:y1, x0:x1] = True; m &= R8.max(-1) < 0.95
        out[n] = round(noise_corr(img, m, 0 if n == "red" else 1), 3)
    return out


def step_handedness():
    print("== 8-spp noise correlation (reference noise = 8 spp - 256 spp, ours = our 8 spp - their 256 spp), same fans either way")
    print("   world mirrored, camera plain :", region_scores(render(build(FROZEN_TRIS, FROZEN_LIGHT, quads=FROZEN_QUADS), 8)))
    sc = build(FROZEN_TRIS, scenes._docs_fan(3, rev=False), mirror_world=False, quads=FROZEN_QUADS)
    print("   world plain, camera mirrored :", region_scores(render(sc, 8, mirror_camera=True)))


def step_frames():
    print("== per visible triangle: 2 diagonals x 3 rotations x 2 windings, noise correlation over the triangle's footprint (all quads share the option)")
    res = {}
    for diag in (0, 1):
        fp = footprints(diag, FROZEN_QUADS)
        for rot in range(3):
            for wind in (0, 1):
                img = render(build([tri_options(diag, rot, wind, rot, wind)] * 15, FROZEN_LIGHT, quads=FROZEN_QUADS), 8)
                for t in range(30):
                    m = (fp == t) & (R8.max(-1) < 0.95)
                    if m.sum() >= 300:
                        res[(t, diag, rot, wind)] = (noise_corr(img, m), int(m.sum()))
    for q in range(15):
        cand = []
        for diag in (0, 1):
            best = [max(((res.get((2 * q + s, diag, r, w), (-9, 0))[0], (r, w)) for r in range(3) for w in (0, 1))) for s in (0, 1)]
            n = [res.get((2 * q + s, diag, 0, 0), (0, 0))[1] for s in (0, 1)]
            if sum(n):
                cand.append((sum(max(b[0], 0) * k for b, k in zip(best, n)) / sum(n), diag, best))
        if cand:
            sc_, diag, best = max(cand)
            print("   %-6s diagonal %d  triangles %s  (correlations %.2f, %.2f)   frozen fan: %s" % (NAMES[q], diag, [tri_options(diag, best[0][1][0], best[0][1][1], best[1][1][0], best[1][1][1])],
                                                                                                   best[0][0], best[1][0], FROZEN_TRIS[q]))


def lit_spread(img):
    fp = footprints(0, FROZEN_QUADS) // 2; out = []
    O = np.minimum(img, 1)
    for q in (0, 2, 3):
        m = (fp == q) & (R8[..., 1] < 0.9) & (R8[..., 1] > 0.02) & (O[..., 1] > 0.02)
        lr = np.log(O[..., 1][m] / R8[..., 1][m]); out.append(round(float(np.subtract(*np.percentile(lr, [75, 25]))), 4))
    return out


def step_light():
    print("== the emitter: 36 downward-facing triangulations; inter-quartile range of log(ours / reference) per pixel on floor, back wall, green wall")
    rows = []
    for diag in (0, 1):
        cyc = [0, 3, 2, 1]
        base = [[cyc[0], cyc[1], cyc[2]], [cyc[0], cyc[2], cyc[3]]] if diag == 0 else [[cyc[1], cyc[2], cyc[3]], [cyc[1], cyc[3], cyc[0]]]
        for ra in range(3):
            for rb in range(3):
                for order in (0, 1):
                    ta, tb = base[0][ra:] + base[0][:ra], base[1][rb:] + base[1][:rb]
                    tris = [ta, tb] if order == 0 else [tb, ta]
                    img = render(build(FROZEN_TRIS, tris, quads=FROZEN_QUADS), 8)
                    rows.append((lit_spread(img), byte_stats(img)["mean_abs"], tris))
    for r in sorted(rows, key=lambda r: sum(r[0]))[:5]:
        print("   ", r)
    print("   frozen:", FROZEN_LIGHT)


def step_corners():
    print("== block corners: scans of the mean byte difference in a window around each corner's silhouette edge, from the public data")
    th = np.tan(np.radians(scenes.CORNELL_DOCS_FOV) / 2)
    short, tall = [(130, 65), (82, 225), (240, 272), (290, 114)], [(423, 247), (265, 296), (314, 456), (472, 406)]
    for blk, h, name in ((short, 165.0, "short"), (tall, 330.0, "tall")):
        for i in range(4):
            x, z = blk[i]
            c = 250 - 250 * (x - 278) / ((z + 800) * th); r0 = 250 - 250 * (h - 273) / ((z + 800) * th); r1 = 250 - 250 * (0 - 273) / ((z + 800) * th)
            w = (max(int(r0) - 3, 0), min(int(r1) + 3, 500), max(int(c) - 8, 0), min(int(c) + 9, 500))
            for j in (0, 1):
                scan = []
                for dlt in (-3, -2, -1, -0.5, 0, 0.5, 1, 2, 3):
                    b2 = [list(p) for p in blk]; b2[i][j] += dlt
                    q = public_quads(short=b2 if blk is short else short, tall=b2 if blk is tall else tall)
                    scan.append((byte_stats(render(build(FROZEN_TRIS, FROZEN_LIGHT, quads=q), 8), w), dlt))
                best = min(scan)
                if best[0] < dict((d, e) for e, d in scan)[0] - 0.02:
                    blk[i] = tuple(v + (best[1] if k == j else 0) for k, v in enumerate(blk[i]))
                print("   %s corner %d %s: best offset %+.1f (%.2f; at 0: %.2f)" % (name, i, "xz"[j], best[1], best[0], dict((d, e) for e, d in scan)[0]))
    print("   short", short, " tall", tall, "\n   frozen", scenes._DOCS_SHORT[0], scenes._DOCS_TALL[0])


def main():
    steps = [a for a in sys.argv[1:] if not a.startswith("-")] or ["fov", "fit", "handedness", "frames", "light", "corners"]
    oracle.build()
    for s in steps:
        globals()["step_" + s]()
    print("== the frozen scene against the 8-spp image:", byte_stats(render(scenes.cornell_box_docs(oracle.bvh_build), 8)))


if __name__ == "__main__":
    main()
