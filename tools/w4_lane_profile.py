#!/usr/bin/env python3
"""Wave-level occupancy of k_trace_w4's phases (experiments/trace_w4_lane_profile.patch, built with
`AB_DEFS=-DRSPT_W4_PROFILE=1 tools/ab_build.sh prof experiments/trace_w4_lane_profile.patch`): how many of a wave's 64 lanes do useful work
in a node step, in a leaf phase, and how many sit idle per outer iteration — over the rays of real wavefront renders (all bounces).
usage (GPU box): RSPT_LIB=exp/librspt_prof.so python tools/w4_lane_profile.py [soup1m|statue] [spp]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rs_pbrt_amd import lib, scenes  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "soup1m"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
os.environ["RSPT_TRACE_STREAMS"] = "1"
lib.init(0)
L = lib.lib()
if wl == "soup1m":
    sc = scenes.triangle_soup(lib.bvh_build, n_tris=1_000_000)
    rd = scenes.soup_render_desc(res=1024, spp=spp, max_depth=8)
elif wl.startswith("c5"):   # c5 (instancing as the reference has it) / c5fixed
    sc = scenes.landscape_standin(lib.bvh_build_gpu, instancing="fixed" if wl == "c5fixed" else "reference")
    rd = scenes.landscape_render_desc(xres=1920, yres=1080, spp=spp)
else:
    sc = scenes.statue_standin(lib.bvh_build)
    rd = scenes.statue_render_desc(spp=spp)
out = (C.c_uint64 * 16)()
with lib.DeviceScene(sc) as ds:
    prof = L.rspt_debug_w4q_prof if os.environ.get("RSPT_TRACE_KERNEL") == "3" else L.rspt_debug_w4_prof   # (one copy of the counters per translation unit)
    lib.render(ds, rd)
    prof(out, 1)
    film, st = lib.render(ds, rd)
    prof(out, 1)
p = [int(v) for v in out]
iters, idle, steps, step_lanes, fetches, fetch_lanes, top_lanes, leafs, leaf_lanes, leaf_trips, leaf_tris = p[:11]
print("workload %s, %d spp: %.1f M samples, %.3f s in trace launches" % (wl, spp, st["samples"] / 1e6, st["t_trace_closest_s"] + st["t_trace_any_s"]))
print("outer iterations per wave-launch total %d; idle lanes at the top of an iteration %.1f / 64" % (iters, idle / max(iters, 1)))
print("node steps %d: lanes in the step %.1f / 64 (%.0f %%); record fetches: %.1f / 64 lanes of a step fetch, %.1f %% of them from the LDS-resident top"
      % (steps, step_lanes / max(steps, 1), 100 * step_lanes / max(64 * steps, 1), fetch_lanes / max(fetches, 1), 100 * top_lanes / max(fetch_lanes, 1)))
print("leaf phases %d (one per %.2f node steps): parked lanes %.1f / 64 (%.0f %%); triangles per lane %.2f, loop trips per phase %.2f (lane utilisation inside the loop %.0f %%)"
      % (leafs, steps / max(leafs, 1), leaf_lanes / max(leafs, 1), 100 * leaf_lanes / max(64 * leafs, 1), leaf_tris / max(leaf_lanes, 1), leaf_trips / max(leafs, 1),
         100 * leaf_tris / max(64 * leaf_trips, 1)))
print("raw:", p[:12], " Q kernel: leaves stopped by the exact box test %d (%.1f %% of the leaf arrivals)" % (p[11], 100.0 * p[11] / max(leaf_lanes, 1)))
os.environ["RSPT_COUNTERS"] = "1"
with lib.DeviceScene(sc) as ds:
    film, st = lib.render(ds, rd)
rays = st["rays_closest"] + st["rays_any"]
print("reference-order counters: %.2f rays per sample (%.2f closest, %.2f any), %.1f nodes and %.2f triangle tests per ray; record fetches per ray (profile) %.1f"
      % (rays / st["samples"], st["rays_closest"] / st["samples"], st["rays_any"] / st["samples"], st["nodes_visited"] / rays, st["tris_tested"] / rays, fetch_lanes / rays))
