#!/usr/bin/env python3
"""One scene, many settings of the library's run-time knobs (environment variables the library reads at every launch): the scene is built and
uploaded once, each setting renders <steps> timed frames after one warm-up.
usage (GPU box): python tools/sweep_env.py <soup1m|statue> "<VAR=a,b,c>" ["<VAR2=x,y>" ...] [--steps N] [--spp S]      (the cross product is run, the first
value of every variable first and again last as a drift check)"""
import itertools
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (first, so that librspt binds to the HIP runtime torch has loaded)
from rs_pbrt_amd import lib, scenes  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
opt = {a.split("=")[0][2:]: a.split("=")[1] for a in sys.argv[1:] if a.startswith("--")}
wl, steps = args[0], int(opt.get("steps", 2))
var = [(a.split("=")[0], a.split("=")[1].split(",")) for a in args[1:]]
lib.init(0)
if wl == "soup1m":
    sc = scenes.triangle_soup(lib.bvh_build, n_tris=1_000_000)
    rd = scenes.soup_render_desc(res=1024, spp=int(opt.get("spp", 256)), max_depth=8)
else:
    sc = scenes.statue_standin(lib.bvh_build)
    rd = scenes.statue_render_desc(spp=int(opt.get("spp", 256)))
film = torch.zeros(scenes.n_pixels(rd) * 4, dtype=torch.float32, device="cuda")
combos = list(itertools.product(*[v for _, v in var]))
combos.append(combos[0])
with lib.DeviceScene(sc) as ds:
    for c in combos:
        for (k, _), v in zip(var, c):
            os.environ[k] = v
        lib.render_device(ds, rd, film.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            st = lib.render_device(ds, rd, film.data_ptr())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print("%s %s: %.1f Msamples/s (trace closest %.3f any %.3f shade %.3f s)" % (wl, " ".join("%s=%s" % (k, v) for (k, _), v in zip(var, c)), st["samples"] / dt / 1e6,
                                                                                      st["t_trace_closest_s"], st["t_trace_any_s"], st["t_shade_s"]), flush=True)
