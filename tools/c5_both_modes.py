#!/usr/bin/env python3
"""The C5 (Landscape) stand-in in both instancing modes from ONE host-side scene build (the build is most of a bench.py --workload c5 run):
Msamples/s of the 1920x1080 frame at 64 spp, path depth 5, Sobol' — what profiles/rNN_bench_c5_*.json report, without their CPU legs.
usage (GPU box): python tools/c5_both_modes.py [steps] > gpurun_out/<tag>/c5_both_modes.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rs_pbrt_amd import abi, lib, scenes

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
lib.init(0)
t0 = time.time()
sc = scenes.landscape_standin(lib.bvh_build_gpu, instancing="fixed")
print("scene build %.1f s, library %s" % (time.time() - t0, lib.source_hash()), flush=True)
rd = scenes.landscape_render_desc(xres=1920, yres=1080, spp=64)
for mode, flag in (("fixed", abi.INSTANCING_FIXED), ("reference", abi.INSTANCING_REFERENCE)):
    sc.desc.instancing_mode = flag
    with lib.DeviceScene(sc) as ds:
        lib.render(ds, rd)   # warm-up
        best = 0.0
        for _ in range(steps):
            film, st = lib.render(ds, rd)
            best = max(best, st["samples"] / st["t_render_s"] / 1e6)
        print("c5 stand-in, instancing %-9s: %.1f Msamples/s (best of %d; closest-hit launches %.3f s, shadow-ray launches %.3f s, shade %.3f s, truncated paths %d)" % (
            mode, best, steps, st["t_trace_closest_s"], st["t_trace_any_s"], st["t_shade_s"], st["truncated_paths"]), flush=True)

# round 5: every tree a MOVING instance (two keys that differ by a rotation; top-level boxes from rspt_motion_bounds): through k_trace_w4<INST, ANIM>
# (RSPT_ANIM_W4=1, default) and through the reference-order loop with the interpolation that served such scenes until round 4 (RSPT_ANIM_W4=0)
if os.environ.get("C5_MOVING", "1") != "0":
    t0 = time.time()
    scm = scenes.landscape_standin(lib.bvh_build_gpu, instancing="fixed", moving=True)
    print("moving scene build %.1f s (%d instances, each with two keys)" % (time.time() - t0, len(scm.instances)), flush=True)
    for w4 in ("1", "0"):
        os.environ["RSPT_ANIM_W4"] = w4
        with lib.DeviceScene(scm) as ds:
            lib.render(ds, rd)
            best = 0.0
            for _ in range(steps):
                film, st = lib.render(ds, rd)
                best = max(best, st["samples"] / st["t_render_s"] / 1e6)
            print("c5 stand-in, every instance moving, %s: %.1f Msamples/s (best of %d; closest-hit launches %.3f s, shadow-ray launches %.3f s, shade %.3f s)" % (
                "k_trace_w4<INST, ANIM>       " if w4 == "1" else "reference-order k_trace<ANIM>", best, steps, st["t_trace_closest_s"], st["t_trace_any_s"], st["t_shade_s"]), flush=True)
    os.environ.pop("RSPT_ANIM_W4", None)
