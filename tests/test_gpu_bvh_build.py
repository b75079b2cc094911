"""-m gpu: the device BVH build (rspt_bvh_build_gpu, bvh_device.h) must reproduce the host build — and with it the
reference's BVHAccel::new — bit for bit: node array (bounds, offsets, counts, axes) and primitive order."""
import numpy as np
import pytest

from rs_pbrt_amd import scenes
from tests.util import small_soup

pytestmark = pytest.mark.gpu


def _tri_soup(n, seed, extent=0.05):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-1, 1, (n, 1, 3)); d = rng.normal(size=(n, 3, 3)) * extent
    P = (c + d).astype(np.float32).reshape(-1, 3)
    return P, np.arange(3 * n, dtype=np.uint32).reshape(-1, 3)


def _same(gpu, P, tri, max_prims):
    nh, oh = gpu.bvh_build(P, tri, max_prims)
    ng, og = gpu.bvh_build_gpu(P, tri, max_prims)
    assert len(nh) == len(ng)
    assert np.array_equal(oh, og)
    assert nh.tobytes() == ng.tobytes()
    return nh


@pytest.mark.parametrize("n,max_prims", [(1, 4), (2, 4), (3, 1), (7, 4), (1000, 4), (1000, 1), (50000, 4), (50000, 255), (300000, 8)])
def test_gpu_build_equals_host_build(gpu, n, max_prims):
    P, tri = _tri_soup(n, 100 + n)
    _same(gpu, P, tri, max_prims)


def test_gpu_build_special_geometry(gpu):
    # Cornell box (shared vertices, axis-aligned quads), lattice duplicates (equal centroids / bucket ties), clusters that
    # cannot be split (degenerate centroid bounds -> big leaves), a chain (exponentially spaced slabs)
    sc = scenes.cornell_box(gpu.bvh_build)
    tri = sc.prims["v"].copy()
    _same(gpu, sc.P, tri, 4)
    from tests.test_gpu_trace import _centroid_clusters
    Pc = _centroid_clusters(40, 7).reshape(-1, 3)
    nodes = _same(gpu, Pc, np.arange(len(Pc), dtype=np.uint32).reshape(-1, 3), 4)
    assert nodes["n_prims"].max() == 40
    quads = []
    for k in range(4):
        for i in range(4):
            for j in range(4):
                c = np.array([i, j, k], np.float32)
                quads += [[c, c + (1, 0, 0), c + (1, 1, 0)], [c, c + (1, 1, 0), c + (0, 1, 0)]]
    T = np.array(quads, np.float32); T = np.concatenate([T, T, T])
    _same(gpu, T.reshape(-1, 3), np.arange(3 * len(T), dtype=np.uint32).reshape(-1, 3), 2)
    xs = (16.0 ** np.arange(30)).astype(np.float32)
    Pch = np.zeros((30, 3, 3), np.float32); Pch[:, :, 0] = xs[:, None]; Pch[:, 0, 1:] = (-4, -4); Pch[:, 1, 1:] = (4, -4); Pch[:, 2, 1:] = (0, 5)
    _same(gpu, Pch.reshape(-1, 3), np.arange(90, dtype=np.uint32).reshape(-1, 3), 1)


def test_gpu_build_million_triangles_and_render(gpu):
    """the bench scene (1 M triangles): identical tree, and a render through a tree built on the device"""
    import time
    sc_h = scenes.triangle_soup(gpu.bvh_build, n_tris=1_000_000)
    # (the scene's primitives are already in BVH order; as an input they are just another triangle list)
    t0 = time.time(); ng, og = gpu.bvh_build_gpu(sc_h.P, sc_h.prims["v"].copy(), 4); t_gpu = time.time() - t0
    t0 = time.time(); nh, oh = gpu.bvh_build(sc_h.P, sc_h.prims["v"].copy(), 4); t_host = time.time() - t0
    assert nh.tobytes() == ng.tobytes() and np.array_equal(oh, og)
    print("1M triangles: gpu build %.3f s, host build %.3f s" % (t_gpu, t_host))
    sc_g = scenes.triangle_soup(gpu.bvh_build_gpu, n_tris=20000, extent=0.03)
    sc_c = scenes.triangle_soup(gpu.bvh_build, n_tris=20000, extent=0.03)
    assert sc_g.nodes.tobytes() == sc_c.nodes.tobytes() and sc_g.prims.tobytes() == sc_c.prims.tobytes()
    rd = scenes.soup_render_desc(res=32, spp=4, max_depth=3)
    ds = gpu.DeviceScene(sc_g)
    try:
        film, st = gpu.render(ds, rd)
    finally:
        ds.close()
    assert st["nan_samples"] == 0 and film[:, 3].min() == 4.0
