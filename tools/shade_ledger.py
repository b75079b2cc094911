#!/usr/bin/env python3
"""profiles/rNN_shade_ledger.md: where the bytes of the shade stage go, per wavefront iteration — the byte count the SOURCE implies per path next to what the PMC passes
measured, slot-for-life schedule against the MOVE schedule.
usage: shade_ledger.py <dir with dispatch_<w>_{FETCH,WRITE}_SIZE.txt of the slot schedule> <dir ... of the MOVE schedule> <dir with queues_<w>.txt> > profiles/r06_shade_ledger.md"""
import re
import sys


def load(fn):
    rows = []
    for l in open(fn):
        m = re.match(r"\s*\d+\s+(\S.*?)\s+grid\s+(\d+)\s+([\d.]+) us(.*)", l)
        if m:
            c = dict((k, float(v)) for k, v in re.findall(r"(\w+) ([\d.e+]+)", m.group(4)))
            rows.append((m.group(1).strip(), float(m.group(3)), c))
    return rows


def first_batch(rows):
    n, out = 0, []
    for r in rows:
        if r[0].startswith("k_raygen"):
            n += 1
        if n == 1 and "bvhdev" not in r[0]:
            out.append(r)
        if n == 2:
            break
    return out


def per_iter(d, w):
    f, wr = first_batch(load("%s/dispatch_%s_FETCH_SIZE.txt" % (d, w))), first_batch(load("%s/dispatch_%s_WRITE_SIZE.txt" % (d, w)))
    its, cur = [], None
    for a, b in zip(f, wr):
        assert a[0] == b[0]
        name = a[0]
        rd, wt = a[2].get("FETCH_SIZE", 0) * 1024 * 2, b[2].get("WRITE_SIZE", 0) * 1024      # FETCH_SIZE x 2: profiles/r06_pmc_calibration.md
        if name.startswith("k_shade"):
            cur = cur or {"bin": [0, 0, 0.0]}
            cur["shade"] = [rd, wt, a[1]]
            its.append(cur); cur = None
        elif name.startswith("k_bin_count") or name.startswith("k_bin_scatter"):
            cur = cur or {"bin": [0, 0, 0.0]}
            cur["bin"][0] += rd; cur["bin"][1] += wt; cur["bin"][2] += a[1]
        elif name.startswith("k_raygen"):
            ray = [rd, wt, a[1]]
    return ray, its


def queues(d, w):
    q = []
    for l in open("%s/queues_%s.txt" % (d, w)):
        m = re.search(r"it (\d+): active (\d+) \(\+ (\d+) that.*closest (\d+) any (\d+) of (\d+)", l)
        if m:
            q.append(tuple(int(x) for x in m.groups()))
    return q


SOURCE = """## 1. What the source moves per path and iteration (`kernels.h shade_path`, `k_bin_*`)

| array | bytes | read by | written by | notes |
|---|---|---|---|---|
| queue entry (`q_active` / `q_sorted`) | 4 | shade, bin_count, bin_scatter | shade (next queue), bin_scatter | dense |
| `state` | 4 | bin_count, shade | shade | |
| `hit_cont` (prim, b0, b1, b2) | 16 | bin_count (`.x`), shade | the closest-hit trace launch | |
| `L_eta` (L, eta scale) | 16 | shade (not in the first launch of a batch: `fresh`) | shade | |
| `beta` | 16 | shade, hits only (not in the first launch) | shade, continuing paths | |
| `ray_cont` | 32 | shade, hits only (12 B of direction used); the trace launch | shade, continuing paths | |
| `sobol_index` | 8 | shade, hits below the depth limit | raygen (MOVE: re-written at the new position, + 8) | |
| pending estimate `nee_c1` | 16 | shade, paths with ST_PENDING | shade, paths that made an estimate | compact form (round 4); the general form adds `nee_c2`, `nee_beta`, `hit_mis`, `ray_mis` (80 B) for the few estimates with a BSDF-sampled term |
| `occluded` | 4 | shade, paths with ST_PENDING | the shadow-ray trace launch | |
| `ray_sh` | 32 | the shadow-ray trace launch | shade, paths that made an estimate | |
| triangle record `tris[prim]` | 48 | shade (hits), bin_count (material word) | — (scene) | a gather: 1.33 lines per record when cold |
| `tri_nuv[prim]` (normals, uvs) | 80 | shade, hits on smooth-shaded meshes (C3) | — (scene) | a gather |
| class key | 1 | bin_scatter | bin_count | only scenes with several lobe-list shapes (C3) |
| MOVE only: `orig` | 4 | shade (from the second MOVE launch) | shade | the path's original slot, for `p_film` / the final radiance |
| MOVE only: `L_final[orig]` | 16 | k_film | shade, once per path, when it ends | replaces the last `L_eta` write |

A path that hits a surface, makes an estimate and continues therefore reads 4 + 4 + 16 + 16 + 16 + 32 + 8 + 16 + 4 = **116 B of path state** (+ 48 .. 128 B of scene records) and writes
32 + 16 + 32 + 16 + 16 + 4 + 12 = **128 B**; an escaping path reads 4 + 4 + 16 + 16 (+ 20 pending) and writes 20.  SURVEY 8(d)'s budget for the same bounce is 180 B (48 per
closest-hit ray + 36 per shadow ray + 96 of state): the stage's own minimum is 1.4 x that budget before any waste — `traffic_over_algorithmic` cannot reach 1.0 with this state layout,
and 1.6 is within 15 % of the floor.
"""


def main():
    slot_dir, move_dir, q_dir = sys.argv[1:4]
    print("# The shade stage's bytes, per wavefront iteration (round 6)\n")
    print("Sources: rocprofv3 `--kernel-trace --pmc FETCH_SIZE` / `WRITE_SIZE` (separate passes), `bench.py --workload <w> --steps 1 --warmup 0 --no-count`, first batch of the frame, one")
    print("MI355X; FETCH_SIZE x 2 (every L2 miss is a 128-byte line tallied at 64: `profiles/r06_pmc_calibration.md`); queue lengths from `RSPT_QUEUE_LOG=1`.  Kernels are")
    print("serialised under `--pmc`, so the microseconds are standalone durations.\n")
    print(SOURCE)
    print("## 2. Measured, per iteration: slots for life (round 5) against MOVE from iteration 1 (round 6)\n")
    for w, title in (("statue", "C3 stand-in (4.3 M triangles, 1920 x 1080, batch of 2^29 paths, two lobe-list shapes: bins on)"), ("soup1m", "C2 (1 M-triangle soup, 1024^2 x 256: one batch, one material: bins off)")):
        ray_a, a = per_iter(slot_dir, w)
        ray_b, b = per_iter(move_dir, w)
        q = queues(q_dir, w)
        print("### %s\n" % title)
        print("| it | paths shaded (front + estimate-only) | of the batch | slot: shade read / written GB | B per path | us | bins GB / us | MOVE: shade read / written GB | B per path | us | bins GB / us |")
        print("|---|---|---|---|---|---|---|---|---|---|---|")
        tot = [0.0] * 6
        for i in range(min(len(a), len(b), len(q))):
            n = q[i][1] + q[i][2]
            if n == 0:
                continue
            sa, sb = a[i]["shade"], b[i]["shade"]
            ba, bb = a[i]["bin"], b[i]["bin"]
            print("| %d | %.1f M + %.1f M | %.3f | %.2f / %.2f | %.0f | %.0f | %.2f / %.0f | %.2f / %.2f | %.0f | %.0f | %.2f / %.0f |" % (
                i, q[i][1] / 1e6, q[i][2] / 1e6, n / q[i][5], sa[0] / 1e9, sa[1] / 1e9, (sa[0] + sa[1]) / n, sa[2], (ba[0] + ba[1]) / 1e9, ba[2],
                sb[0] / 1e9, sb[1] / 1e9, (sb[0] + sb[1]) / n, sb[2], (bb[0] + bb[1]) / 1e9, bb[2]))
            tot[0] += sa[0] + sa[1] + ba[0] + ba[1]; tot[1] += sa[2] + ba[2]; tot[2] += sb[0] + sb[1] + bb[0] + bb[1]; tot[3] += sb[2] + bb[2]
            if i >= 1:
                tot[4] += sa[0] + sa[1] + ba[0] + ba[1]; tot[5] += sb[0] + sb[1] + bb[0] + bb[1]
        print("\nBatch totals (shade + bins): slots %.1f GB in %.1f ms; MOVE %.1f GB in %.1f ms (%+.1f %% bytes, %+.1f %% time).  Iterations >= 1 only: %.1f -> %.1f GB (%+.1f %%).  k_raygen writes %.1f GB.\n" % (
            tot[0] / 1e9, tot[1] / 1e3, tot[2] / 1e9, tot[3] / 1e3, (tot[2] / tot[0] - 1) * 100, (tot[3] / tot[1] - 1) * 100, tot[4] / 1e9, tot[5] / 1e9, (tot[5] / tot[4] - 1) * 100, ray_b[1] / 1e9))


if __name__ == "__main__":
    main()
