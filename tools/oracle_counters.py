#!/usr/bin/env python3
"""per-sample traversal / ray / bounce counts of the BASELINE configs, counted by
the CPU oracle on the identical flattened BVH with identical Sobol' samples, and the algorithmic
bytes per path sample they imply:
    B_alg = sum_rays (32 N_node + 48 N_tri) + 96 R_closest + 72 R_any + 96 N_bounce + 32
C2 / C3 are counted at a reduced spp (per-sample means converge; stated in the output).
usage: python tools/oracle_counters.py > profiles/oracle_counters.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle  # noqa: E402
from rs_pbrt_amd import scenes  # noqa: E402


def count(name, sc, rd, note):
    r = pyoracle.render(sc, rd, threads=os.cpu_count() or 1)
    c = r["counters"]
    n = c["samples"]
    c["mis_rays_included_in_rays_closest"] = c.pop("mis_rays")
    b_alg = (32.0 * c["nodes_visited"] + 48.0 * c["tris_tested"] + 96.0 * c["rays_closest"] + 72.0 * c["rays_any"] + 96.0 * c["bounces"] + 32.0 * n) / n
    return {"config": name, "counted_at": note, "triangles": sc.n_tris, "bvh_nodes": len(sc.nodes), "counters": c,
            "per_sample": {k: c[k] / n for k in ("nodes_visited", "tris_tested", "rays_closest", "rays_any", "bounces")},
            "alg_bytes_per_sample": b_alg, "oracle_seconds": r["seconds"], "oracle_threads": os.cpu_count()}


def main():
    out = []
    out.append(count("C1 Cornell Box 400x400, 64 spp, depth 5", scenes.cornell_box(pyoracle.bvh_build), scenes.cornell_render_desc(400, 64), "full config"))
    out.append(count("C2 1M-triangle soup 1024x1024, 256 spp, depth 8", scenes.triangle_soup(pyoracle.bvh_build),
                     scenes.soup_render_desc(1024, 2, max_depth=8), "2 spp (1/128 of the samples)"))
    out.append(count("C3 statue stand-in 4.3M triangles 1920x1080, 1024 spp, depth 5", scenes.statue_standin(pyoracle.bvh_build),
                     scenes.statue_render_desc(spp=1), "1 spp (1/1024 of the samples)"))
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
