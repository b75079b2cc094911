"""-m gpu: the BASELINE configurations at their full scene sizes.  The oracle cannot render a full
1024x1024x256 frame in test time, so parity at full size is checked (a) against the oracle on a crop
window of the full-resolution frame (same Sobol' indices: the sampler depends on the full sample
bounds only through the resolution, which a crop keeps) and (b) through size-independent properties
of the whole frame: idempotence, shard additivity, sample/weight bookkeeping."""
import numpy as np
import pytest

from rs_pbrt_amd import abi, multigpu, scenes
from tests.util import film_rmse

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def soup1m(gpu):
    sc = scenes.triangle_soup(gpu.bvh_build)  # C2: 1 000 002 triangles
    ds = gpu.DeviceScene(sc)
    yield sc, ds
    ds.close()


def test_c2_full_frame_properties(gpu, soup1m):
    sc, ds = soup1m
    assert sc.n_tris == 1_000_002
    rd = scenes.soup_render_desc(res=1024, spp=8, max_depth=8)
    a, st = gpu.render(ds, rd)
    b, _ = gpu.render(ds, rd)
    assert st["samples"] == 1024 * 1024 * 8 and st["nan_samples"] == 0
    assert np.array_equal(a, b)                                   # idempotent, bit for bit
    w = a[:, 3]
    assert (w >= 8).all() and (w == np.round(w)).all()            # box filter: integral weights, >= spp
    assert 0 < (w > 8).sum() < 16384                              # exact-zero film offsets splat into a neighbour (Q22)
    assert np.isfinite(a).all() and a[:, :3].min() >= 0 and a[:, 1].mean() > 0.01
    # tile shards (the multi-GPU decomposition) add up to the frame
    acc = np.zeros_like(a)
    n = 0
    for r in range(8):
        rd_r = scenes.soup_render_desc(res=1024, spp=8, max_depth=8, shard=multigpu.shard_for_rank(r, 8))
        f, s = gpu.render(ds, rd_r)
        acc += f
        n += s["samples"]
        assert 0.08 < s["samples"] / st["samples"] < 0.18          # balanced deal of Morton chunks
    assert n == st["samples"]
    assert np.array_equal(acc[:, 3], a[:, 3]) and np.allclose(acc, a, rtol=1e-6, atol=1e-7)


def test_c2_crop_window_matches_oracle(gpu, oracle, soup1m):
    sc, ds = soup1m
    rd = scenes.soup_render_desc(res=1024, spp=16, max_depth=8, crop=(0.47, 0.53, 0.47, 0.53))  # 62x62 pixels of the 1024^2 frame
    film, st = gpu.render(ds, rd)
    ref = oracle.render(sc, rd, threads=8)
    assert np.array_equal(film[:, 3], ref["film"][:, 3])
    assert film_rmse(film, ref["film"]) < 1e-5
    assert st["samples"] == ref["counters"]["samples"]


def test_c2_rays_against_oracle_on_the_full_bvh(gpu, oracle, soup1m):
    sc, ds = soup1m
    from tests.util import random_rays
    rays = random_rays(50000, 77, -1.2, 1.2)
    for any_hit in (False, True):
        assert gpu.trace(ds, rays, any_hit=any_hit).tobytes() == oracle.trace(sc, rays, any_hit=any_hit).tobytes()


def test_c3_statue_4m_triangles_crop_matches_oracle(gpu, oracle):
    sc = scenes.statue_standin(gpu.bvh_build)  # 4.30 M triangles, smooth normals, plastic, 3 quad lights
    assert sc.n_tris > 4_250_000
    ds = gpu.DeviceScene(sc)
    try:
        rd = scenes.statue_render_desc(spp=8, crop=(0.45, 0.5, 0.4, 0.5))  # 96x108 pixels of the 1920x1080 frame
        film, st = gpu.render(ds, rd)
        ref = oracle.render(sc, rd, threads=8)
        assert np.array_equal(film[:, 3], ref["film"][:, 3])
        assert film_rmse(film, ref["film"]) < 1e-5
        from tests.util import random_rays
        rays = random_rays(20000, 78, -1.5, 1.5)
        assert gpu.trace(ds, rays).tobytes() == oracle.trace(sc, rays).tobytes()
    finally:
        ds.close()
