#!/usr/bin/env python3
"""One eighth of the headline frame (shard 0 of 8), rendered a few times: wall time per call against the library's own kernel time (rspt_stats.t_kernels_s) — what a rank of an
8-GPU run pays beyond its share of the frame.  Under `rocprofv3 --kernel-trace` tools/per_dispatch.py lists the launches.  usage: python tools/shard_probe.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rs_pbrt_amd import lib, multigpu, scenes
import bench

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
lib.init(0)
sys.argv = sys.argv[:1]
args = bench.parse()
sc, mk_rd, spp, name = bench.build_workload(args, "soup1m", lib, scenes)
ds = lib.DeviceScene(sc)
film = torch.zeros(1024 * 1024, 4, device="cuda")
full = mk_rd(spp, (0, 1, 64))
lib.render_device(ds, full, film.data_ptr())          # (allocations, the shadow-ray kernel measurement)
for shard, label in (((0, 1, 64), "full frame"), (multigpu.shard_for_rank(0, 8), "shard 0 of 8")):
    rd = mk_rd(spp, shard)
    lib.render_device(ds, rd, film.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sts = [lib.render_device(ds, rd, film.data_ptr()) for _ in range(reps)]
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e3
    st = sts[-1]
    print("%s: wall %.2f ms per call; library: t_render %.2f ms, kernels %.2f ms (trace %.2f, shade %.2f), %d samples" % (
        label, wall, st["t_render_s"] * 1e3, st["t_kernels_s"] * 1e3, st["t_trace_s"] * 1e3, st["t_shade_s"] * 1e3, st["samples"]))
