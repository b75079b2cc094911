#!/usr/bin/env python3
"""How rs_pbrt_amd/scenes.py `cornell_box_docs` was recovered from the reference's two documentation renders
(tests/golden/reference_cornell_docs.npz = docs/source/cornell_box_{8,256}_pixelsamples.png; the scene file is not in the reference tree).
Every step is a search over scene descriptions whose objective is agreement of the ORACLE's render with those images; nothing here touches
the product.  Test infrastructure (it drives oracle/), minutes of CPU per step:

    python tools/recover_cornell_docs.py [fov] [fit] [handedness] [frames] [light] [blocks] [sharpness]      (no argument: all of them)

Steps and what they print (the values frozen in scenes.py are the ones a run of this script ends on):
  fov         sub-pixel frame edges of the 256-spp image -> field of view
  fit         least squares (L, wall / red / green / block albedos) on the 2 x 2 box-filtered 256-spp image
  handedness  8-spp noise correlation with the camera mirrored vs. the world mirrored
  frames      per visible triangle: diagonal x rotation x winding by noise correlation over its footprint
  light       the 36 downward-facing triangulations of the emitter by the log-ratio spread on the directly lit faces
  blocks      the two blocks as squares (centre, side, angle): coordinate descent on the byte differences at 8 spp, steps 0.25 .. 0.02, from the public data
  sharpness   the frozen scene with one value moved at a time (fov, light height / size, radiance, albedos): every one of them is an optimum

How the search actually went (round 3): fov and fit first; then handedness with arbitrary vertex orders (only the mirrored WORLD correlated,
0.4 .. 0.7), frames and light in that world (correlations 0.8 .. 0.96, 16 % byte-equal pixels); then, with the fans known, the mirrored CAMERA
turned out to be the real thing (74 % byte-equal at once), and the blocks (90 %, then 94 % as exact squares).  The steps below run in the
final configuration, so each one shows its ingredient's optimum with everything else already in place.
"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from rs_pbrt_amd import scenes  # noqa: E402
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


PUBLIC_SHORT = [(130, 65), (82, 225), (240, 272), (290, 114)]
PUBLIC_TALL = [(423, 247), (265, 296), (314, 456), (472, 406)]


def public_quads(short=None, tall=None):
    """floor, ceiling, back, green, red, short block top + 4 sides, tall block top + 4 sides; blocks from their top corners a, b, c, d"""
    return ([q for q, _, _ in scenes._DOCS_ROOM] + scenes._docs_block_quads(short or PUBLIC_SHORT, 165.0, False)
            + scenes._docs_block_quads(tall or PUBLIC_TALL, 330.0, True))


def build(tris, light_tris, mirror_world=False, L=(100.0,) * 3, white=(0.4,) * 3, red=0.5, green=0.5, block=0.5, quads=None, light_y=547.8, light=None):
    """tris[i]: the two index triples of quad i"""
    sb = SceneBuilder()
    m = [sb.add_material(matte(tuple(white)))] * 3 + [sb.add_material(matte((0, green, 0))), sb.add_material(matte((red, 0, 0)))] + [sb.add_material(matte((block,) * 3))] * 10
    sx = -1.0 if mirror_world else 1.0
    for q, t, mat in zip(quads or public_quads(), tris, m):
        sb.add_mesh(np.array([(sx * p[0], p[1], p[2]) for p in q], F32), t, mat)
    sb.add_mesh(np.array([(sx * p[0], light_y, p[2]) for p in (light or scenes._DOCS_LIGHT)], F32), light_tris, m[0], emit=tuple(L))
    return sb.finish(oracle.bvh_build)


def render(sc, spp, res=500, mirror_camera=True, fov=scenes.CORNELL_DOCS_FOV, **kw):
    rd = scenes.cornell_docs_render_desc(spp, res, mirror_camera=mirror_camera, **kw)
    if fov != scenes.CORNELL_DOCS_FOV:
        rd2 = scenes.make_render_desc(res, res, spp, scenes.CORNELL_DOCS_LOOK_AT, fov, **kw)   # the fov is in raster_to_camera only
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
FROZEN_LIGHT = scenes._docs_fan(3)
FROZEN_QUADS = public_quads(scenes._docs_square(*scenes._DOCS_SHORT[0]), scenes._docs_square(*scenes._DOCS_TALL[0]))


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
    d[:, 0] *= -1.0   # `Scale -1 1 1` on the camera
    o = np.array(scenes.CORNELL_DOCS_LOOK_AT[0], float)
    best_t = np.full(len(d), np.inf); best = np.full(len(d), -1)
    for qi, q in enumerate((quads or public_quads()) + [scenes._DOCS_LIGHT]):
        Q = np.array(q, float)
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
    print("== the mirror: on the camera (`Scale -1 1 1`) or in the world (x negated, light cycle turned so that it still faces down) — the same picture")
    img = render(build(FROZEN_TRIS, FROZEN_LIGHT, quads=FROZEN_QUADS), 8)
    print("   world plain, camera mirrored :", region_scores(img), byte_stats(img))
    img = render(build(FROZEN_TRIS, scenes._docs_fan(3, rev=True), mirror_world=True, quads=FROZEN_QUADS), 8, mirror_camera=False)
    print("   world mirrored, camera plain :", region_scores(img), byte_stats(img))


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
    print("== the emitter: 36 downward-facing triangulations; inter-quartile range of log(ours / reference) per pixel on floor, back wall, green wall; byte stats")
    rows = []
    for diag in (0, 1):
        cyc = [0, 1, 2, 3]
        base = [[cyc[0], cyc[1], cyc[2]], [cyc[0], cyc[2], cyc[3]]] if diag == 0 else [[cyc[1], cyc[2], cyc[3]], [cyc[1], cyc[3], cyc[0]]]
        for ra in range(3):
            for rb in range(3):
                for order in (0, 1):
                    ta, tb = base[0][ra:] + base[0][:ra], base[1][rb:] + base[1][:rb]
                    tris = [ta, tb] if order == 0 else [tb, ta]
                    img = render(build(FROZEN_TRIS, tris, quads=FROZEN_QUADS), 8)
                    rows.append((lit_spread(img), byte_stats(img)["exact"], tris))
    for r in sorted(rows, key=lambda r: -r[1])[:4]:
        print("   ", r)
    print("   frozen:", FROZEN_LIGHT)


def step_blocks():
    print("== the blocks as squares (centre x, centre z, side, angle of a->d in degrees), coordinate descent on the mean byte difference, from the public data's fit")
    def sq_of(c4):   # the square closest to four corners a, b, c, d
        c4 = np.array(c4, float); ctr = c4.mean(0); e = ((c4[3] - c4[0]) + (c4[2] - c4[1])) / 2; f = ((c4[1] - c4[0]) + (c4[2] - c4[3])) / 2
        return [float(ctr[0]), float(ctr[1]), float((np.hypot(*e) + np.hypot(*f)) / 2), float(np.degrees(np.arctan2(e[1], e[0])))]
    P = dict(s=sq_of(PUBLIC_SHORT), t=sq_of(PUBLIC_TALL))
    run = lambda: byte_stats(render(build(FROZEN_TRIS, FROZEN_LIGHT, quads=public_quads(scenes._docs_square(*P["s"][:3], np.radians(P["s"][3])), scenes._docs_square(*P["t"][:3], np.radians(P["t"][3])))), 8))  # noqa: E731
    base = run(); print("   public data as squares", {k: np.round(v, 3).tolist() for k, v in P.items()}, base, flush=True)
    for step in (1.0, 0.5, 0.25, 0.1, 0.05, 0.02):
        for sweep in range(3):
            changed = False
            for key in ("s", "t"):
                for i in range(4):
                    for d in (-step, step):
                        dd = d * (0.2 if i == 3 else 1.0)
                        P[key][i] += dd; st = run()
                        if st["mean_abs"] < base["mean_abs"] - 5e-4:
                            base = st; changed = True
                        else:
                            P[key][i] -= dd
            if not changed:
                break
        print("   step %.2f" % step, {k: np.round(v, 3).tolist() for k, v in P.items()}, base, flush=True)
    print("   in radians: %.4f, %.4f;  frozen: %s %s" % (np.radians(P["s"][3]), np.radians(P["t"][3]), scenes._DOCS_SHORT[0], scenes._DOCS_TALL[0]))


def step_sharpness():
    print("== the frozen scene with one value moved: share of byte-equal pixels (frozen: %s)" % byte_stats(render(build(FROZEN_TRIS, FROZEN_LIGHT, quads=FROZEN_QUADS), 8))["exact"])
    ex = lambda **kw: byte_stats(render(build(FROZEN_TRIS, FROZEN_LIGHT, quads=FROZEN_QUADS, **{k: v for k, v in kw.items() if k != "fov"}), 8, **{k: v for k, v in kw.items() if k == "fov"}))["exact"]  # noqa: E731
    print("   fov        ", [(f, ex(fov=f)) for f in (39.14, 39.1445, 39.148, 39.15)])
    print("   light y    ", [(y, ex(light_y=y)) for y in (548.8, 548.0, 547.9, 547.85, 547.75, 547.7, 547.0)])
    L0 = scenes._DOCS_LIGHT
    print("   light size ", [(d, ex(light=[(x + (d if x > 300 else -d), y, z + (d if z > 300 else -d)) for x, y, z in L0])) for d in (-0.5, 0.5)])
    print("   light shift", [(d, ex(light=[(x + d[0], y, z + d[1]) for x, y, z in L0])) for d in ((-0.5, 0), (0.5, 0), (0, -0.5), (0, 0.5))])
    print("   radiance   ", [(v, ex(L=(v,) * 3)) for v in (99.0, 99.5, 100.5, 101.0)])
    print("   walls      ", [(v, ex(white=(v,) * 3)) for v in (0.395, 0.398, 0.402, 0.405)])
    print("   red, green, blocks at 0.495 / 0.505:", [ex(**{k: v}) for k in ("red", "green", "block") for v in (0.495, 0.505)])
    for q in range(15):
        alt = []
        for k in range(4):
            t = list(FROZEN_TRIS); t[q] = scenes._docs_fan(k)
            alt.append(byte_stats(render(build(t, FROZEN_LIGHT, quads=FROZEN_QUADS), 8))["exact"])
        print("   fan start of %-6s 0..3: %s" % (NAMES[q], alt))


def main():
    steps = [a for a in sys.argv[1:] if not a.startswith("-")] or ["fov", "fit", "handedness", "frames", "light", "blocks", "sharpness"]
    oracle.build()
    for s in steps:
        globals()["step_" + s]()
    print("== the frozen scene against the 8-spp image:", byte_stats(render(scenes.cornell_box_docs(oracle.bvh_build), 8)))


if __name__ == "__main__":
    main()
