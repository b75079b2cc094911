#!/usr/bin/env python3
"""Micro-benchmark of the traversal stage through rspt_trace_device on the C2 soup:
coherent camera rays and incoherent interior rays, closest- and any-hit."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rs_pbrt_amd import scenes, lib, abi

ap = argparse.ArgumentParser()
ap.add_argument("--tris", type=int, default=1_000_000)
ap.add_argument("--rays", type=int, default=1 << 22)
ap.add_argument("--repeat", type=int, default=5)
ap.add_argument("--check", action="store_true", help="compare the two kernels' outputs")
args = ap.parse_args()

lib.init(0)
sc = scenes.triangle_soup(lib.bvh_build, n_tris=args.tris)
ds = lib.DeviceScene(sc)
n = args.rays
rng = np.random.default_rng(5)
sets = {}
r = np.zeros(n, abi.RAY_DT)
r["o"] = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
d = rng.normal(size=(n, 3)); r["d"] = (d / np.linalg.norm(d, axis=1)[:, None]).astype(np.float32); r["t_max"] = np.inf
sets["incoherent"] = r
r = np.zeros(n, abi.RAY_DT)
side = int(np.sqrt(n)); yy, xx = np.mgrid[0:side, 0:side]
px = ((xx.reshape(-1) + 0.5) / side - 0.5) * 1.4; py = ((yy.reshape(-1) + 0.5) / side - 0.5) * 1.4
dd = np.stack([px, py, np.ones_like(px) * 2.0], 1); dd /= np.linalg.norm(dd, axis=1)[:, None]
r = r[:side * side]; r["o"] = (0, 0, -4); r["d"] = dd.astype(np.float32); r["t_max"] = np.inf
sets["camera"] = r
to_light = sets["incoherent"].copy()
tgt = np.stack([rng.uniform(-.5, .5, n), np.full(n, 1.5), rng.uniform(-.5, .5, n)], 1).astype(np.float32)
to_light["d"] = tgt - to_light["o"]; to_light["t_max"] = 0.9999
sets["shadow"] = to_light
for name, rays in sets.items():
    rb = lib.DeviceBuffer(rays.nbytes); rb.upload(rays)
    hb = lib.DeviceBuffer(len(rays) * abi.HIT_DT.itemsize)
    for any_hit in ((True,) if name == "shadow" else (False, True)):
        os.environ["RSPT_COUNTERS"] = "1"
        lib.trace_device(ds, rb, len(rays), hb, any_hit=any_hit, repeat=1)
        nodes, tris, _ = lib.last_counters()
        ref = hb.download(abi.HIT_DT, len(rays)) if args.check else None
        os.environ["RSPT_COUNTERS"] = "0"
        ms = lib.trace_device(ds, rb, len(rays), hb, any_hit=any_hit, repeat=args.repeat)
        extra = ""
        if args.check:
            got = hb.download(abi.HIT_DT, len(rays))
            extra = " identical=%s" % (got.tobytes() == ref.tobytes())
        gb = (32.0 * nodes + 48.0 * tris + (72 if any_hit else 96) * len(rays)) / 1e9
        print(f"{name:11s} any={int(any_hit)} rays={len(rays)} {ms:8.3f} ms  {len(rays) / ms / 1e3:8.1f} Mrays/s  nodes/ray {nodes / len(rays):6.1f} tris/ray {tris / len(rays):5.2f}  alg {gb / ms * 1e3:7.1f} GB/s{extra}", flush=True)
    rb.free(); hb.free()
