#!/usr/bin/env python3
"""Verbose GPU-vs-oracle check used while developing (prints numbers instead of asserting)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle
from rs_pbrt_amd import scenes, lib, abi


def rand_rays(n, seed, lo, hi, inward=True):
    rng = np.random.default_rng(seed)
    rays = np.zeros(n, abi.RAY_DT)
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1)[:, None]
    rays["o"] = o; rays["d"] = d.astype(np.float32); rays["t_max"] = np.inf; rays["id"] = np.arange(n)
    return rays


def cmp_trace(name, sc, ds, rays):
    for any_hit in (False, True):
        t = time.time(); ref = pyoracle.trace(sc, rays, any_hit=any_hit); tc = time.time() - t
        t = time.time(); got = lib.trace(ds, rays, any_hit=any_hit); tg = time.time() - t
        same = ref.tobytes() == got.tobytes()
        nm = int((ref["prim"] != got["prim"]).sum())
        print(f"[trace] {name} any={any_hit} n={len(rays)} hits={(ref['prim'] != abi.MISS).sum()} bit-identical={same} prim-mismatch={nm} cpu {tc:.2f}s gpu {tg:.2f}s")
        if not same and not any_hit:
            bad = np.nonzero((ref["prim"] != got["prim"]) | (ref["t"] != got["t"]) | (ref["b0"] != got["b0"]))[0][:5]
            for i in bad: print("   ", i, ref[i], got[i])


def cmp_render(name, sc, ds, rd, threads=8):
    t = time.time(); ref = pyoracle.render(sc, rd, threads=threads, want_li=True); tc = time.time() - t
    t = time.time(); li, st = lib.render_samples(ds, rd); tg = time.time() - t
    film, st2 = lib.render(ds, rd)
    rli = ref["li"]
    same = (rli == li).all(axis=2)
    print(f"[render] {name} samples={rli.shape[0] * rli.shape[1]} bit-identical samples {same.mean() * 100:.4f}%  max|d|={np.abs(rli - li).max():.3e} mean|d|={np.abs(rli - li).mean():.3e}")
    a = scenes.film_to_rgb(ref["film"]); b = scenes.film_to_rgb(film)
    rmse = float(np.sqrt(np.mean((a.astype(np.float64) - b) ** 2)))
    print(f"         film RMSE {rmse:.3e}  mean {a.mean():.4f}/{b.mean():.4f}  weight-equal {(ref['film'][:, 3] == film[:, 3]).all()}  cpu({threads}t) {tc:.2f}s  gpu {st2['t_render_s']:.3f}s kernels {st2['t_kernels_s']:.3f}s trace {st2['t_trace_s']:.3f}s nan {st2['nan_samples']}")
    print("         oracle counters", ref["counters"])
    print("         gpu stats", {k: v for k, v in st2.items() if k not in ('t_render_s', 't_kernels_s', 't_trace_s')})
    return rmse


def main():
    lib.init(0)
    sc = scenes.cornell_box(lib.bvh_build)
    ds = lib.DeviceScene(sc)
    cmp_trace("cornell", sc, ds, rand_rays(200000, 1, 50, 500))
    rd = scenes.cornell_render_desc(res=64, spp=16)
    cmp_render("cornell 64x64x16", sc, ds, rd)
    for variant in ("mixed", "rough"):
        scv = scenes.cornell_box(lib.bvh_build, variant=variant)
        dsv = lib.DeviceScene(scv)
        cmp_render(f"cornell-{variant} 64x64x16", scv, dsv, rd)
    soup = scenes.triangle_soup(lib.bvh_build, n_tris=100000)
    dso = lib.DeviceScene(soup)
    cmp_trace("soup100k", soup, dso, rand_rays(200000, 2, -1.2, 1.2))
    rds = scenes.soup_render_desc(res=64, spp=8)
    cmp_render("soup100k 64x64x8", soup, dso, rds)
    rd = scenes.cornell_render_desc(res=400, spp=64)
    os.environ["RSPT_COUNTERS"] = "1"
    cmp_render("cornell 400x400x64 (C1)", sc, ds, rd)


if __name__ == "__main__":
    main()
