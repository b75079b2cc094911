"""-m gpu: the traversal stage (k_trace) through rspt_trace against the oracle's restatement of
BVHAccel::intersect / intersect_p + Triangle::intersect.  No transcendentals on this stage, so the
bar is bit-exact (prim, t, b0, b1, b2)."""
import numpy as np
import pytest

from rs_pbrt_amd import abi, scenes
from tests.util import random_rays, small_soup

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cornell(gpu):
    sc = scenes.cornell_box(gpu.bvh_build)
    ds = gpu.DeviceScene(sc)
    yield sc, ds
    ds.close()


@pytest.fixture(scope="module")
def soup(gpu):
    sc = small_soup(gpu.bvh_build)
    ds = gpu.DeviceScene(sc)
    yield sc, ds
    ds.close()


def test_closest_hit_bit_exact_cornell(gpu, oracle, cornell):
    sc, ds = cornell
    rays = random_rays(100000, 1, 20, 530)
    got, ref = gpu.trace(ds, rays), oracle.trace(sc, rays)
    assert (ref["prim"] != abi.MISS).sum() > 50000
    assert got.tobytes() == ref.tobytes()


def test_any_hit_bit_exact_cornell(gpu, oracle, cornell):
    sc, ds = cornell
    rays = random_rays(100000, 2, 20, 530, t_max=300.0)
    got, ref = gpu.trace(ds, rays, any_hit=True), oracle.trace(sc, rays, any_hit=True)
    assert 1000 < (ref["prim"] == 0).sum() < 99000
    assert got.tobytes() == ref.tobytes()


def test_closest_and_any_bit_exact_soup(gpu, oracle, soup):
    sc, ds = soup
    rays = random_rays(200000, 3, -1.3, 1.3)
    for any_hit in (False, True):
        got, ref = gpu.trace(ds, rays, any_hit=any_hit), oracle.trace(sc, rays, any_hit=any_hit)
        assert got.tobytes() == ref.tobytes()


def test_alternative_kernels_bit_exact(gpu, oracle, soup, monkeypatch):
    """RSPT_TRACE_KERNEL=0 (reference-order loop, also the counting and fix-up kernel) and =1 (two-box records)
    stay selectable and bit-identical to the default four-box kernel"""
    sc, ds = soup
    rays = random_rays(100000, 31, -1.3, 1.3)
    for kernel in ("0", "1", "2"):
        monkeypatch.setenv("RSPT_TRACE_KERNEL", kernel)
        for any_hit in (False, True):
            assert gpu.trace(ds, rays, any_hit=any_hit).tobytes() == oracle.trace(sc, rays, any_hit=any_hit).tobytes()


def test_round5_kernel_shapes_and_xcd_dealing_bit_exact(gpu, oracle, soup, monkeypatch):
    """the selectable forms of k_trace_w4 that round 5 measured and left off by default stay bit-identical: XCD-affine ray dealing (RSPT_XCD_DEAL=1: which wave
    traces which ray changes, no ray's result does), one 1024-thread / two 512-thread workgroups per CU with 512 / 256 root-side records in LDS (RSPT_W4_SHAPE=1 / 2)"""
    sc, ds = soup
    rays = random_rays(200000, 77, -1.3, 1.3)
    ref = {a: oracle.trace(sc, rays, any_hit=a).tobytes() for a in (False, True)}
    for env in (dict(RSPT_XCD_DEAL="1"), dict(RSPT_W4_SHAPE="1"), dict(RSPT_W4_SHAPE="2"), dict(RSPT_W4_SHAPE="1", RSPT_XCD_DEAL="1")):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for any_hit in (False, True):
            assert gpu.trace(ds, rays, any_hit=any_hit).tobytes() == ref[any_hit], env
        for k in env:
            monkeypatch.delenv(k)


def test_matches_brute_force(gpu, oracle, soup):
    """BVH result == O(N) scan over all triangles in list order (structural invariant)."""
    sc, ds = soup
    rays = random_rays(2000, 4, -1.3, 1.3)
    got, ref = gpu.trace(ds, rays), oracle.trace(sc, rays, brute=True)
    hit = ref["prim"] != abi.MISS
    assert np.array_equal(got["prim"] != abi.MISS, hit)
    assert np.array_equal(got["t"][hit], ref["t"][hit])


def test_axis_aligned_and_degenerate_rays(gpu, oracle, cornell):
    """zero direction components give infinite reciprocals; rays starting on surfaces; tiny t_max"""
    sc, ds = cornell
    rays = random_rays(6000, 5, 20, 530)
    rays["d"][:2000] = np.eye(3, dtype=np.float32)[np.arange(2000) % 3] * np.where(np.arange(2000) % 2, 1, -1)[:, None]
    rays["o"][2000:3000, 1] = 0.0            # on the floor plane
    rays["o"][3000:4000, 0] = 0.0            # on the green wall plane
    rays["t_max"][4000:5000] = 1e-3
    rays["d"][5000:, 2] = 0.0                # in-plane directions
    n = np.linalg.norm(rays["d"][5000:], axis=1)[:, None]
    rays["d"][5000:] /= n
    for any_hit in (False, True):
        assert gpu.trace(ds, rays, any_hit=any_hit).tobytes() == oracle.trace(sc, rays, any_hit=any_hit).tobytes()


def test_shared_edges_watertight(gpu, oracle, cornell):
    """rays aimed exactly at triangle edge midpoints and vertices (shared diagonals, shared quad edges,
    open boundary edges of the box): identical decisions, and the interior ones never leak through"""
    sc, ds = cornell
    P = sc.P[sc.prims["v"].reshape(-1)].reshape(-1, 3, 3)
    mids = np.concatenate([(P[:, 0] + P[:, 1]) / 2, (P[:, 1] + P[:, 2]) / 2, (P[:, 0] + P[:, 2]) / 2, P[:, 0]]).astype(np.float32)
    o = np.array([278, 273, -800], np.float32)
    rays = np.zeros(len(mids), abi.RAY_DT)
    rays["o"] = o
    d = mids - o
    rays["d"] = (d / np.linalg.norm(d, axis=1)[:, None]).astype(np.float32)
    rays["t_max"] = np.inf
    got, ref = gpu.trace(ds, rays), oracle.trace(sc, rays)
    assert got.tobytes() == ref.tobytes()
    assert (got["prim"] != abi.MISS).mean() > 0.9  # only open boundary edges of the box may miss


def test_empty_inputs_and_errors(gpu, cornell):
    sc, ds = cornell
    assert len(gpu.trace(ds, np.zeros(0, abi.RAY_DT))) == 0
    import ctypes as C
    L = gpu.lib()
    assert L.rspt_trace(None, None, 10, None, 0) == abi.E_INVALID
    assert b"null" in L.rspt_last_error()
    bad = scenes.cornell_box(gpu.bvh_build)
    bad.prims["v"][0, 0] = 10 ** 6  # out-of-range vertex index must be rejected on the host
    h = C.c_void_p()
    assert L.rspt_scene_create(C.addressof(bad.desc), C.addressof(h)) == abi.E_INVALID


def test_large_batch_sorted_properties(gpu, soup):
    """full-size property check (no oracle): closest t <= any other reported hit along the same ray,
    and any-hit == (closest-hit found something) for 4 M rays."""
    sc, ds = soup
    rays = random_rays(1 << 22, 6, -1.3, 1.3)
    c = gpu.trace(ds, rays)
    a = gpu.trace(ds, rays, any_hit=True)
    assert np.array_equal(c["prim"] != abi.MISS, a["prim"] == 0)
    hit = c["prim"] != abi.MISS
    assert (c["t"][hit] > 0).all()
    b = np.stack([c["b0"][hit], c["b1"][hit], c["b2"][hit]], 1)
    assert np.abs(b.sum(1) - 1).max() < 1e-5 and (b >= 0).all()
    # idempotence: shrinking t_max to slightly beyond the hit returns the same hit; to slightly
    # before it, a different (farther) one or none
    r2 = rays[hit][:100000].copy()
    r2["t_max"] = c["t"][hit][:100000] * np.float32(1.0001)
    c2 = gpu.trace(ds, r2)
    assert np.array_equal(c2["prim"], c["prim"][hit][:100000]) and np.array_equal(c2["t"], c["t"][hit][:100000])
    r2["t_max"] = c["t"][hit][:100000] * np.float32(0.9999)
    assert (gpu.trace(ds, r2)["prim"] == abi.MISS).all()


def test_deep_bvh_stack_overflow_fixup(gpu, oracle):
    """a chain-like BVH (exponentially spaced slabs) drives the traversal stack past the persistent
    kernel's 24-entry LDS column: those rays are redone by the 64-entry loop and must still be
    bit-identical to the reference order"""
    n = 30  # ratio 16 > 12 SAH buckets: every split isolates the farthest slab, so the tree is a chain
    xs = (16.0 ** np.arange(n)).astype(np.float32)
    P = np.zeros((n, 3, 3), np.float32)
    P[:, :, 0] = xs[:, None]
    P[:, 0, 1:] = (-4, -4); P[:, 1, 1:] = (4, -4); P[:, 2, 1:] = (0, 5)
    sb = scenes.SceneBuilder()
    m = sb.add_material(scenes.matte((0.5, 0.5, 0.5)))
    sb.add_mesh(P.reshape(-1, 3), np.arange(3 * n).reshape(-1, 3), m)
    sc = sb.finish(gpu.bvh_build, max_prims_in_node=1)
    depth = 0
    stack = [(0, 1)]
    while stack:
        i, d = stack.pop()
        depth = max(depth, d)
        if sc.nodes["n_prims"][i] == 0:
            stack += [(i + 1, d + 1), (int(sc.nodes["offset"][i]), d + 1)]
    assert 28 <= depth <= 64
    rng = np.random.default_rng(8)
    k = 20000
    rays = np.zeros(k, abi.RAY_DT)
    rays["o"] = np.stack([np.full(k, -1.0), rng.uniform(-1, 1, k), rng.uniform(-1, 1, k)], 1).astype(np.float32)
    rays["o"][: k // 2, 0] = xs[-1] * 1.5                      # half of them travel in -x
    tgt = np.stack([np.where(np.arange(k) < k // 2, -1.0, xs[-1] * 1.5), rng.uniform(-1, 1, k), rng.uniform(-1, 1, k)], 1)
    d = tgt - rays["o"]
    rays["d"] = (d / np.linalg.norm(d, axis=1)[:, None]).astype(np.float32)
    rays["t_max"] = np.inf
    import os
    ds = gpu.DeviceScene(sc)
    try:
        buf = gpu.DeviceBuffer(rays.nbytes); buf.upload(rays)
        out = gpu.DeviceBuffer(k * abi.HIT_DT.itemsize)
        # (kernel, spill rows): the four-box kernel with its global spill rows (no ray left over), the same
        # with the spill rows disabled (fix-up pass takes the deep rays), and the two-box kernel (LDS only)
        for kernel, spill_rows, expect_fixup in (("2", "84", False), ("2", "0", True), ("1", "84", True)):
            os.environ["RSPT_TRACE_KERNEL"], os.environ["RSPT_W4_SPILL_ROWS"] = kernel, spill_rows
            for any_hit in (False, True):
                gpu.trace_device(ds, buf, k, out, any_hit=any_hit)
                got = out.download(abi.HIT_DT, k)
                assert got.tobytes() == oracle.trace(sc, rays, any_hit=any_hit).tobytes()
                if not any_hit:
                    assert (gpu.last_counters()[2] > 0) == expect_fixup
        buf.free(); out.free()
    finally:
        os.environ.pop("RSPT_TRACE_KERNEL", None); os.environ.pop("RSPT_W4_SPILL_ROWS", None)
        ds.close()


def _centroid_clusters(cluster, seed):
    """`cluster` triangles around each of 9 centres whose bounding boxes share the centre bit-exactly (vertices
    c + d, c - d, c + e with e inside the box, all coordinates short dyadic fractions)"""
    rng = np.random.default_rng(seed)
    centres = [(0, 0, 0)] + [(x, y, z) for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)]
    tris = []
    for c in np.array(centres, np.float32):
        for _ in range(cluster):
            d = rng.integers(-24, 25, 3).astype(np.float32) / np.float32(64)
            d[d == 0] = np.float32(1 / 64)
            e = np.abs(d) * np.float32(0.25) * rng.choice([-1, 1], 3).astype(np.float32)
            tris.append(np.stack([c + d, c - d, c + e]))
    return np.array(tris, np.float32)


@pytest.mark.parametrize("cluster", [17, 40, 200])
def test_large_leaves(gpu, oracle, cluster):
    """triangles whose bounds share one centroid cannot be split (bvh.rs:218-229: degenerate centroid bounds -> leaf),
    so the builder emits leaves with `cluster` primitives whatever max_prims_in_node says; more than 15 do not fit
    the four-box kernel's packed leaf reference and go through its side table.  Every kernel must agree with the
    oracle, including which candidate inside a leaf wins (leaf order)."""
    import os
    P = _centroid_clusters(cluster, 90 + cluster)
    sb = scenes.SceneBuilder()
    m = sb.add_material(scenes.matte((0.5, 0.5, 0.5)))
    sb.add_mesh(P.reshape(-1, 3), np.arange(3 * len(P)).reshape(-1, 3), m)
    big = sb.finish(gpu.bvh_build, max_prims_in_node=4)
    assert big.nodes["n_prims"].max() == cluster
    rays = random_rays(60000, 41, -1.4, 1.4)
    rays["d"][:30000] = -rays["o"][:30000] + rng_targets(30000)  # half of them aimed at the clusters
    rays["d"] /= np.linalg.norm(rays["d"], axis=1)[:, None]
    ds = gpu.DeviceScene(big)
    try:
        for kernel in ("2", "1", "0"):
            os.environ["RSPT_TRACE_KERNEL"] = kernel
            for any_hit in (False, True):
                ref = oracle.trace(big, rays, any_hit=any_hit)
                assert gpu.trace(ds, rays, any_hit=any_hit).tobytes() == ref.tobytes()
        assert (ref["prim"] == 0).sum() > 3000
    finally:
        os.environ.pop("RSPT_TRACE_KERNEL", None)
        ds.close()


def rng_targets(n):
    rng = np.random.default_rng(5)
    centres = np.array([(0, 0, 0)] + [(x, y, z) for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], np.float32)
    return (centres[rng.integers(0, 9, n)] + rng.normal(size=(n, 3)) * 0.1).astype(np.float32)


def test_single_triangle_and_two_triangle_scenes(gpu, oracle):
    """the root is a leaf (one node) / the root's children are both leaves (one four-box record with two empty slots)"""
    for n in (1, 2):
        sb = scenes.SceneBuilder()
        m = sb.add_material(scenes.matte((0.5, 0.5, 0.5)))
        P = np.array([[-1, -1, 0], [1, -1, 0], [0, 1, 0], [2, -1, 1], [4, -1, 1], [3, 1, 1]], np.float32)[: 3 * n]
        sb.add_mesh(P, np.arange(3 * n).reshape(-1, 3), m)
        sc = sb.finish(gpu.bvh_build, max_prims_in_node=1)
        assert len(sc.nodes) == (1 if n == 1 else 3)
        rays = random_rays(20000, 42 + n, -2, 4)
        rays["o"][:, 2] = -3.0
        rays["d"] = np.abs(rays["d"]) * np.array([0.3, 0.3, 1.0], np.float32)
        rays["d"] /= np.linalg.norm(rays["d"], axis=1)[:, None]
        rays["o"][::2, :2] -= 1.5
        ds = gpu.DeviceScene(sc)
        try:
            for any_hit in (False, True):
                got, ref = gpu.trace(ds, rays, any_hit=any_hit), oracle.trace(sc, rays, any_hit=any_hit)
                assert got.tobytes() == ref.tobytes()
            assert (ref["prim"] == 0).sum() > 100
        finally:
            ds.close()


def test_ties_duplicates_and_lattice_geometry(gpu, oracle):
    """equal-distance candidates: every triangle of a lattice of axis-aligned quads exists three times (shuffled, so
    the copies land in different leaves), rays start on lattice points and travel along lattice directions, so hits
    fall on shared edges / vertices / box faces all the time.  The winner among equal t is decided by the reference's
    visiting order and its strict `<` tests; all kernels must reproduce it bit for bit."""
    import os
    rng = np.random.default_rng(2024)
    quads = []
    for k in range(5):
        for i in range(4):
            for j in range(4):
                for axis in range(3):
                    c = np.array([i, j, k], np.float32)
                    u = np.eye(3, dtype=np.float32)[(axis + 1) % 3]; v = np.eye(3, dtype=np.float32)[(axis + 2) % 3]
                    quads.append([c, c + u, c + u + v]); quads.append([c, c + u + v, c + v])
    tris = np.array(quads, np.float32)
    tris = np.concatenate([tris, tris, tris[:, ::-1]])           # duplicates (one copy with the opposite winding)
    tris = tris[rng.permutation(len(tris))]
    sb = scenes.SceneBuilder()
    m = sb.add_material(scenes.matte((0.5, 0.5, 0.5)))
    sb.add_mesh(tris.reshape(-1, 3), np.arange(3 * len(tris)).reshape(-1, 3), m)
    sc = sb.finish(gpu.bvh_build, max_prims_in_node=2)
    n = 80000
    rays = np.zeros(n, abi.RAY_DT)
    rays["o"] = rng.integers(-1, 6, (n, 3)).astype(np.float32) + rng.choice([0.0, 0.5, 0.25], (n, 3)).astype(np.float32)
    d = rng.integers(-2, 3, (n, 3)).astype(np.float32)
    d[(d == 0).all(1)] = (1, 1, 1)
    rays["d"] = d / np.linalg.norm(d, axis=1)[:, None]
    rays["d"][: n // 4] = d[: n // 4]                            # a quarter with unnormalised integer directions
    rays["t_max"] = np.where(rng.random(n) < 0.2, rng.integers(1, 4, n), np.inf).astype(np.float32)
    ds = gpu.DeviceScene(sc)
    try:
        for kernel in ("2", "1", "0"):
            os.environ["RSPT_TRACE_KERNEL"] = kernel
            for any_hit in (False, True):
                ref = oracle.trace(sc, rays, any_hit=any_hit)
                got = gpu.trace(ds, rays, any_hit=any_hit)
                assert got.tobytes() == ref.tobytes(), (kernel, any_hit, int((got["prim"] != ref["prim"]).sum()))
        assert (ref["prim"] == 0).mean() > 0.3
    finally:
        os.environ.pop("RSPT_TRACE_KERNEL", None)
        ds.close()


def test_device_libm_equals_host_libm(gpu, oracle):
    """rspt_libm (the device's sinf / cosf / logf / log2f / expf / acosf / atan2f: glibc's algorithms restated, glibc_libm.h) against the libm
    of THIS host — the one the oracle's values come from — bit for bit: dense sets over the ranges the path produces, 2^24 random arguments
    of the wider domain, and the edge cases (zeros, denormals, quadrant boundaries, domain ends, infinities, NaN)"""
    rng = np.random.default_rng(11)
    dense = lambda lo, hi, step: np.arange(int(np.float32(lo).view(np.uint32)), int(np.float32(hi).view(np.uint32)) + 1, step, dtype=np.uint32).view(np.float32)  # noqa: E731
    rnd = lambda lo, hi: (rng.random(1 << 24, dtype=np.float32) * np.float32(hi - lo) + np.float32(lo)).astype(np.float32)  # noqa: E731
    bits = rng.integers(0, 1 << 32, 1 << 22, dtype=np.uint64).astype(np.uint32).view(np.float32)   # raw bit patterns: every exponent, NaNs, infinities
    edge = np.array([0.0, -0.0, 1e-45, -1e-45, 1e-38, 2.4e-4, 0.5, 0.78539816, 0.7853982, 1.0, -1.0, 1.5707964, 3.1415927, 4.712389, 6.2831855, -3.1415927,
                     88.0, 88.8, -87.4, -103.9, -104.1, 100.0, -119.5, np.inf, -np.inf, np.nan], np.float32)
    args = {"sin": np.concatenate([dense(0.0, 6.2831855, 16), -dense(0.0, 6.2831855, 112), rnd(-119.9, 119.9), edge]),
            "cos": np.concatenate([dense(0.0, 6.2831855, 16), -dense(0.0, 6.2831855, 112), rnd(-119.9, 119.9), edge]),
            "log": np.concatenate([dense(1e-30, 1e30, 64), bits, edge]),
            "log2": np.concatenate([dense(1e-30, 1e30, 64), bits, edge]),
            "exp": np.concatenate([-dense(1e-30, 110.0, 32), dense(1e-30, 90.0, 64), bits, edge]),
            "acos": np.concatenate([dense(0.0, 1.0, 8), -dense(0.0, 1.0, 8), bits, edge])}
    for fn, x in args.items():
        got = gpu.libm(fn, x)
        ref = np.empty_like(x)
        oracle.lib().orc_libm(gpu.LIBM[fn], x.ctypes.data, None, x.size, ref.ctypes.data)
        ok = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
        assert ok.all(), (fn, x[~ok][:4], got[~ok][:4], ref[~ok][:4])
    n = 1 << 25
    y = np.concatenate([rnd(-3.0, 3.0), rnd(-3.0, 3.0)[::-1] * np.float32(1e-6), bits, edge])
    x = np.concatenate([rnd(-3.0, 3.0)[::-1], rnd(-3.0, 3.0), bits[::-1], edge[::-1]])
    got = gpu.libm("atan2", y, x)
    ref = np.empty_like(y)
    oracle.lib().orc_libm(gpu.LIBM["atan2"], y.ctypes.data, x.ctypes.data, y.size, ref.ctypes.data)
    ok = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
    assert ok.all(), ("atan2", y[~ok][:4], x[~ok][:4], got[~ok][:4], ref[~ok][:4])


def test_quantised_shadow_ray_kernel_flags_are_the_references(gpu, oracle, soup, cornell, monkeypatch):
    """round 6: k_trace_w4q (trace_w4q.h: 64-byte records on an 8-bit grid, conservative box tests, the reference's exact test on every leaf box) forced by RSPT_ANY_Q=1 —
    occlusion flags byte-identical to the oracle's BVHAccel::intersect_p on random rays, short shadow-like segments, axis-parallel rays (a zero direction component: the
    axis leaves the grid test and becomes a containment test), rays that start on a triangle's plane, and rays with a huge / tiny direction (reciprocals beyond 2^60)."""
    monkeypatch.setenv("RSPT_ANY_Q", "1")
    for (sc, ds), lo, hi in ((soup, -1.3, 1.3), (cornell, 20.0, 530.0)):
        rng = np.random.default_rng(606)
        sets = [random_rays(150000, 61, lo, hi), random_rays(150000, 62, lo, hi, t_max=0.35 * (hi - lo))]
        ax = random_rays(60000, 63, lo, hi)
        k = rng.integers(0, 3, len(ax))
        d = ax["d"].copy(); d[np.arange(len(ax)), k] = 0.0; d[::3, (k[::3] + 1) % 3] = 0.0    # one zero component, every third ray two
        ax["d"] = d
        sets.append(ax)
        sc_rays = random_rays(60000, 64, lo, hi)
        sc_rays["d"] = (sc_rays["d"] * np.where(rng.uniform(size=(len(sc_rays), 1)) < 0.5, 1e-25, 1e25)).astype(np.float32)   # |1 / d| far beyond 2^60 / far below
        sc_rays["d"][::2, 0] = np.float32(1.0)
        sets.append(sc_rays)
        for rays in sets:
            got, ref = gpu.trace(ds, rays, any_hit=True), oracle.trace(sc, rays, any_hit=True)
            assert got.tobytes() == ref.tobytes()
    monkeypatch.setenv("RSPT_ANY_Q", "0")
    sc, ds = soup
    rays = random_rays(50000, 65, -1.3, 1.3)
    assert gpu.trace(ds, rays, any_hit=True).tobytes() == oracle.trace(sc, rays, any_hit=True).tobytes()
