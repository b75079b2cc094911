#!/usr/bin/env python3
"""Ray-coherence A/B for the traversal kernel: the same set of bounce rays through rspt_trace_device in different
QUEUE ORDERS — the order the wavefront produces (path slots = pixel-major), fully sorted by (origin cell Morton code, direction
octant), by (octant, Morton), and coarsely binned (what a binning fused into the shade stage's queue append could deliver).
A full sort is an upper bound for what any re-ordering pass can buy; the pass itself is priced separately (a one-pass K7b-style
scatter moves ~18 G entries/s).  Rays: camera rays of the workload's own camera are traced, every hit spawns one cosine-distributed
bounce ray (diffuse surface), in pixel-major order — the closest-hit queue of wavefront iteration 1.
usage (GPU box): python tools/ray_order_ab.py [soup1m|statue] [--res N]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rs_pbrt_amd import scenes, lib, abi

ap = argparse.ArgumentParser()
ap.add_argument("workload", nargs="?", default="soup1m")
ap.add_argument("--res", type=int, default=2048)
ap.add_argument("--repeat", type=int, default=3)
args = ap.parse_args()
lib.init(0)
if args.workload == "soup1m":
    sc = scenes.triangle_soup(lib.bvh_build_gpu)
    eye, look, fov, aspect = np.array([0, 0, -4.0]), np.array([0, 0, 0.0]), 40.0, 1.0
else:
    sc = scenes.statue_standin(lib.bvh_build_gpu)
    (eye, look, _up) = [np.array(v, float) for v in scenes.STATUE_LOOK_AT]
    fov, aspect = scenes.STATUE_FOV, 16 / 9
ds = lib.DeviceScene(sc)
rng = np.random.default_rng(11)
w, h = args.res, int(args.res / aspect)
fwd = (look - eye) / np.linalg.norm(look - eye); right = np.cross([0, 1.0, 0], fwd); right /= np.linalg.norm(right); up = np.cross(fwd, right)
t = np.tan(np.radians(fov) / 2)
yy, xx = np.mgrid[0:h, 0:w]
sx = ((xx.reshape(-1) + rng.uniform(0, 1, w * h)) / w * 2 - 1) * t * (aspect if aspect > 1 else 1)
sy = (1 - (yy.reshape(-1) + rng.uniform(0, 1, w * h)) / h * 2) * t * (1 if aspect > 1 else 1 / aspect)
d = fwd[None] + sx[:, None] * right[None] + sy[:, None] * up[None]
cam = np.zeros(w * h, abi.RAY_DT)
cam["o"] = eye.astype(np.float32); cam["d"] = (d / np.linalg.norm(d, axis=1)[:, None]).astype(np.float32); cam["t_max"] = np.inf
hits = lib.trace(ds, cam)
ok = hits["prim"] != abi.MISS
P = sc.P[sc.prims["v"][hits["prim"][ok]]]                      # (n, 3, 3)
b = np.stack([hits["b0"][ok], hits["b1"][ok], hits["b2"][ok]], 1)[:, :, None]
p = (P * b).sum(1)
n = np.cross(P[:, 1] - P[:, 0], P[:, 2] - P[:, 0]); n /= np.linalg.norm(n, axis=1)[:, None]
n[(n * cam["d"][ok]).sum(1) > 0] *= -1                        # face the camera ray
u1, u2 = rng.uniform(0, 1, len(p)), rng.uniform(0, 1, len(p))
r_, phi = np.sqrt(u1), 2 * np.pi * u2
a = np.where(np.abs(n[:, :1]) > 0.9, [[0, 1.0, 0]], [[1.0, 0, 0]]); tx = np.cross(n, a); tx /= np.linalg.norm(tx, axis=1)[:, None]; ty = np.cross(n, tx)
dirs = tx * (r_ * np.cos(phi))[:, None] + ty * (r_ * np.sin(phi))[:, None] + n * np.sqrt(1 - u1)[:, None]
rays = np.zeros(len(p), abi.RAY_DT)
rays["o"] = (p + n * 1e-4).astype(np.float32); rays["d"] = dirs.astype(np.float32); rays["t_max"] = np.inf
print("%s: %d camera rays, %d hit -> bounce rays in pixel-major order" % (args.workload, w * h, len(rays)), flush=True)


def morton(o, bits):
    lo, hi = o.min(0), o.max(0)
    q = np.minimum(((o - lo) / (hi - lo + 1e-20) * (1 << bits)).astype(np.uint64), (1 << bits) - 1)
    code = np.zeros(len(o), np.uint64)
    for k in range(bits):
        for ax in range(3):
            code |= ((q[:, ax] >> np.uint64(k)) & np.uint64(1)) << np.uint64(3 * k + ax)
    return code


octant = ((rays["d"][:, 0] < 0).astype(np.uint64) | ((rays["d"][:, 1] < 0).astype(np.uint64) << np.uint64(1)) | ((rays["d"][:, 2] < 0).astype(np.uint64) << np.uint64(2)))
m10, m3, m5 = morton(rays["o"], 10), morton(rays["o"], 3), morton(rays["o"], 5)
orders = {
    "pixel-major (as produced)": np.arange(len(rays)),
    "random": rng.permutation(len(rays)),
    "sorted (morton30, octant)": np.argsort((m10 << np.uint64(3)) | octant, kind="stable"),
    "sorted (octant, morton30)": np.argsort((octant << np.uint64(30)) | m10, kind="stable"),
    "binned 4096 (octant, morton9), order kept inside a bin": np.argsort((octant << np.uint64(9)) | m3, kind="stable"),
    "binned 2^18 (octant, morton15)": np.argsort((octant << np.uint64(15)) | m5, kind="stable"),
    "binned 8 (octant only)": np.argsort(octant, kind="stable"),
}
hb = lib.DeviceBuffer(len(rays) * abi.HIT_DT.itemsize)
base = None
for name, perm in orders.items():
    rr = np.ascontiguousarray(rays[perm])
    rb = lib.DeviceBuffer(rr.nbytes); rb.upload(rr)
    for any_hit in (False, True):
        lib.trace_device(ds, rb, len(rr), hb, any_hit=any_hit, repeat=1)
        ms = lib.trace_device(ds, rb, len(rr), hb, any_hit=any_hit, repeat=args.repeat)
        if base is None:
            base = {}
        base.setdefault(any_hit, ms)
        print("%-56s any=%d %8.3f ms %8.1f Mrays/s  x%.2f vs pixel-major" % (name, any_hit, ms, len(rr) / ms / 1e3, base[any_hit] / ms), flush=True)
    rb.free()
