#!/usr/bin/env python3
"""Headline benchmark: Mpath-samples/s of the gfx950 wavefront path integrator on BASELINE.json
configs[1] — synthetic 1 M random-triangle soup, PathIntegrator depth 8, Sobol' 256 spp,
1024x1024 (resolution fixed by SURVEY.md §8d).

A "step" is one full render of that frame (1024*1024*256 = 268 M camera samples through
raygen -> trace -> shade -> film), with the scene already resident in HBM.  With N ranks the
Morton-ordered 16x16 tiles of the frame are dealt to the ranks (strong scaling: the frame is
fixed) and the per-rank film buffers are summed onto rank 0 with one RCCL reduce inside the
timed region.  Prints ONE JSON line on rank 0.

    python bench.py                       # N=1, 1 warm-up + 2 timed steps
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 ... bench.py --gpus 8
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--integrator", default="path", choices=["path", "ao"], help="ao: AOIntegrator (64 cosine-sampled shadow rays per camera sample)")
    ap.add_argument("--workload", default="soup1m", choices=["soup1m", "cornell", "statue", "statue_tex", "c4"])
    ap.add_argument("--tris", type=int, default=1_000_000)
    ap.add_argument("--res", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-spp", type=int, default=16)
    return ap.parse_args()


def build_workload(args, lib, scenes, shard):
    if args.workload == "soup1m":
        res, spp = args.res or 1024, args.spp or 256
        sc = scenes.triangle_soup(lib.bvh_build_gpu, n_tris=args.tris)
        mk = lambda s, sh: scenes.soup_render_desc(res=res, spp=s, max_depth=8, shard=sh, integrator=args.integrator)  # noqa: E731
        name = "synthetic %d-triangle soup, path depth 8, sobol %d spp, %dx%d" % (args.tris, spp, res, res)
    elif args.workload == "cornell":
        res, spp = args.res or 400, args.spp or 64
        sc = scenes.cornell_box(lib.bvh_build_gpu)
        mk = lambda s, sh: scenes.cornell_render_desc(res=res, spp=s, shard=sh, integrator=args.integrator)  # noqa: E731
        name = "Cornell Box, path depth 5, sobol %d spp, %dx%d" % (spp, res, res)
    else:
        xres, spp = args.res or 1920, args.spp or 1024
        yres = xres * 9 // 16
        tex = args.workload in ("statue_tex", "c4")  # image-textured Kd + bump map + textured ground
        sc = scenes.statue_standin(lib.bvh_build_gpu, textured=tex, many_lights=64 if args.workload == "c4" else 0)  # c4: SURVEY 8(d) C4 stand-in
        mk = lambda s, sh: scenes.statue_render_desc(xres=xres, yres=yres, spp=s, shard=sh, integrator=args.integrator)  # noqa: E731
        name = "statue stand-in (4.3 M triangles%s%s), path depth 5, sobol %d spp, %dx%d" % (", image-textured + bump-mapped" if tex else "",
                                                                                                   ", 64 small area lights (C4 stand-in)" if args.workload == "c4" else "", spp, xres, yres)
    if args.integrator == "ao":
        name += " [AOIntegrator, 64 shadow rays per camera sample]"
    return sc, mk, spp, name


def measured_traffic(args):
    """HBM-side bytes of the trace launches of one step from the committed rocprofv3 PMC passes
    (profiles/r01_pmc_traffic.json: FETCH_SIZE / WRITE_SIZE with the gfx950 correction of
    MI355X_MICROARCH.md).  PMC counters cannot be read from inside this process, so the number is
    only reported for the exact workload it was measured on; otherwise null."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
    except (OSError, ValueError):
        return None
    default_cfg = args.workload == "soup1m" and args.tris == 1_000_000 and not args.res and not args.spp and args.gpus == 1
    return t["trace_traffic_bytes_per_step"] if (default_cfg and t.get("workload") == args.workload) else None


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch  # first, so librspt binds to the HIP runtime torch already loaded
    import torch.distributed as dist
    from rs_pbrt_amd import lib, multigpu, scenes
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    lib.init(local_rank)

    shard = multigpu.shard_for_rank(rank, world)  # contiguous Morton chunks of 64 tiles, round-robin over ranks
    t0 = time.time()
    sc, mk_rd, spp, wl_name = build_workload(args, lib, scenes, shard)
    rd = mk_rd(spp, shard)
    t_scene = time.time() - t0
    t0 = time.time()
    ds = lib.DeviceScene(sc)
    t_upload = time.time() - t0
    npix = scenes.n_pixels(rd)
    film = torch.zeros(npix * 4, dtype=torch.float32, device="cuda")

    def step():
        st = lib.render_device(ds, rd, film.data_ptr())
        if world > 1:
            multigpu.reduce_film(film)  # X1: RCCL reduce(sum) over xGMI (tile borders overlap: sum, not gather)
        return st

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # counting pass (deterministic: identical counts in the timed passes) for the algorithmic-bytes roofline
    counts = None
    for w in range(args.warmup):
        if w == 0:
            os.environ["RSPT_COUNTERS"] = "1"
            counts = step()
            os.environ["RSPT_COUNTERS"] = "0"
        else:
            step()
    if counts is None:
        os.environ["RSPT_COUNTERS"] = "1"
        counts = step()
        os.environ["RSPT_COUNTERS"] = "0"
    fence()
    t0 = time.perf_counter()
    stats = [step() for _ in range(args.steps)]
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        samples_t = torch.tensor([float(stats[0]["samples"])], dtype=torch.float64, device="cuda")
        dist.all_reduce(samples_t, op=dist.ReduceOp.SUM)
        samples_per_step = float(samples_t.item())
    else:
        samples_per_step = float(stats[0]["samples"])

    if rank == 0:
        # roofline of the dominant kernel k_trace (closest + any launches of one step, this rank):
        # SURVEY.md §8(d) bytes: 32 B per BVH node fetched + 48 B per triangle tested + ray/hit queue
        # records (96 B per closest-hit ray, 72 B per any-hit ray)
        trace_bytes = 32.0 * counts["nodes_visited"] + 48.0 * counts["tris_tested"] + 96.0 * counts["rays_closest"] + 72.0 * counts["rays_any"]
        t_trace = sum(s["t_trace_s"] for s in stats) / len(stats)
        t_kernels = sum(s["t_kernels_s"] for s in stats) / len(stats)
        achieved = trace_bytes / t_trace / 1e9
        out = {
            "metric": "Mpath-samples/sec (whole node)", "value": samples_per_step * args.steps / elapsed / 1e6, "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl_name, "samples_per_step": samples_per_step, "tiles": "16x16 Morton, chunks of 64 dealt round-robin",
                       "film_reduce": "RCCL reduce(sum) to rank 0" if world > 1 else "none (1 GPU)"},
            "roofline": {"bound": "hbm", "kernel": "k_trace (BVH traversal + triangle test, closest + any launches)",
                         "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": measured_traffic(args),
                         "alg_bytes_per_step_rank0": trace_bytes, "trace_s_per_step": t_trace, "kernels_s_per_step": t_kernels,
                         "trace_launches_per_step": stats[0]["trace_launches"],
                         "whole_path_alg_bytes_per_sample": counts["alg_bytes"] / max(counts["samples"], 1),
                         "rays_per_sample": (counts["rays_closest"] + counts["rays_any"]) / max(counts["samples"], 1),
                         "nodes_per_ray": counts["nodes_visited"] / max(counts["rays_closest"] + counts["rays_any"], 1),
                         "mrays_per_s": {"closest": counts["rays_closest"] / t_trace / 1e6, "any": counts["rays_any"] / t_trace / 1e6,
                                         "note": "rays of one step / wall time of its (overlapped) trace launches"}},
            "setup_s": {"scene_and_bvh_build": t_scene, "upload": t_upload, "bvh_builder": "rspt_bvh_build_gpu (device, bit-identical to BVHAccel::new)"},
        }
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is reported on rank 0 at N = 1 only
            from oracle import pyoracle  # CPU baseline leg only
            ncores = os.cpu_count() or 1
            rd_cpu = mk_rd(args.cpu_spp, (0, 1, 64))
            r = pyoracle.render(sc, rd_cpu, threads=ncores)
            out["cpu_baseline"] = {"value": r["counters"]["samples"] / r["seconds"] / 1e6, "unit": "Msamples/s", "cores": ncores, "kind": "port",
                                   "sample": "same scene and frame at %d spp (%d samples), C++ oracle restatement of rs_pbrt's tile loop, %d threads"
                                             % (args.cpu_spp, r["counters"]["samples"], ncores)}
        print(json.dumps(out), flush=True)
    ds.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
