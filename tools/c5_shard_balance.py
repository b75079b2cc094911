#!/usr/bin/env python3
"""How even is the static deal of Morton tile chunks over 8 ranks?  Renders every shard (r, 8, chunk) of a frame on ONE GPU in turn and
prints max / mean shard time and the 8-GPU speed-up it predicts (before the film reduce).  usage: python tools/c5_shard_balance.py [c5|c2]"""
import sys
import time

sys.path.insert(0, ".")
import torch  # noqa: E402

from rs_pbrt_amd import lib, scenes  # noqa: E402

lib.init(0)
which = sys.argv[1] if len(sys.argv) > 1 else "c5"
cases = [("c5 fixed", lambda: scenes.landscape_standin(lib.bvh_build_gpu, instancing="fixed"), lambda sh: scenes.landscape_render_desc(spp=64, shard=sh), 1920 * 1080)] if which == "c5" else \
        [("c2", lambda: scenes.triangle_soup(lib.bvh_build_gpu), lambda sh: scenes.soup_render_desc(shard=sh), 1024 * 1024)]
for name, mk_scene, mk_rd, npix in cases:
    ds = lib.DeviceScene(mk_scene())
    film = torch.zeros(npix * 4, dtype=torch.float32, device="cuda")

    def t(sh):
        rd = mk_rd(sh)
        lib.render_device(ds, rd, film.data_ptr())
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lib.render_device(ds, rd, film.data_ptr())
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3
    full = t((0, 1, 64))
    for chunk in (64, 16, 4, 1):
        ms = [t((r, 8, chunk)) for r in range(8)]
        print("%s chunk %2d: full %.1f ms; shards %s; max/mean %.3f; predicted 8-GPU speed-up %.2f" % (name, chunk, full, " ".join("%.1f" % x for x in ms), max(ms) / (sum(ms) / 8), full / max(ms)))
    ds.close()
