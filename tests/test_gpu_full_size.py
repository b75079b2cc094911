"""-m gpu: BASELINE.json's C2 at its FULL size (1 M-triangle soup, 1024x1024, sobol 256 spp, path depth 8 =
268 M camera samples) through size-independent properties, plus oracle parity on a crop window of the
same frame at the full sample count (the oracle cannot render the whole frame in test time).

  * every pixel of the full frame carries exactly spp filter weights, no NaN sample, sample count exact;
  * the frame is reproducible (two renders agree);
  * the Morton-tile shards that bench.py deals to N GPUs add up to the unsharded frame (linearity of the
    film: what the RCCL reduce relies on);
  * a 12x8 crop of that frame, rendered as a crop window at 256 spp, agrees with the oracle
    (weights exact, RMSE bar of tests/test_gpu_render.py).  (Its pixel values are not those of the full frame:
    SobolSampler derives its resolution and pixel offsets from the sample bounds, sobol.rs:110-150.)"""
import numpy as np
import pytest

from rs_pbrt_amd import scenes
from tests.util import film_rmse

pytestmark = pytest.mark.gpu
RES, SPP = 1024, 256


def test_c2_full_size_properties_and_crop_parity(gpu, oracle):
    sc = scenes.triangle_soup(gpu.bvh_build_gpu, n_tris=1_000_000)
    ds = gpu.DeviceScene(sc)
    try:
        full, st = gpu.render(ds, scenes.soup_render_desc(res=RES, spp=SPP, max_depth=8))
        again, _ = gpu.render(ds, scenes.soup_render_desc(res=RES, spp=SPP, max_depth=8))
        acc = np.zeros_like(full)
        n = 0
        for r in range(2):
            f, s = gpu.render(ds, scenes.soup_render_desc(res=RES, spp=SPP, max_depth=8, shard=(r, 2, 64)))
            acc += f
            n += s["samples"]
        # a crop window in the middle of the frame (pixels [506, 518) x [508, 516)), full sample count
        crop = (506 / RES, 518 / RES, 508 / RES, 516 / RES)
        rd_crop = scenes.soup_render_desc(res=RES, spp=SPP, max_depth=8, crop=crop)
        win, st_win = gpu.render(ds, rd_crop)
    finally:
        ds.close()
    assert st["samples"] == RES * RES * SPP and st["nan_samples"] == 0
    assert full.shape == (RES * RES, 4)
    assert full[:, 3].min() == SPP and full[:, 3].max() <= SPP + 2  # box filter: own samples, plus exact-zero offsets of a neighbour (Q22)
    assert np.isfinite(full).all() and full[:, :3].min() >= 0.0 and full[:, :3].max() > 0.0
    assert np.array_equal(again[:, 3], full[:, 3]) and np.allclose(again, full, rtol=1e-6, atol=1e-7)
    assert n == RES * RES * SPP
    assert np.array_equal(acc[:, 3], full[:, 3]) and np.allclose(acc, full, rtol=1e-6, atol=1e-6)

    ref = oracle.render(sc, rd_crop, threads=8, want_li=False)
    assert win.shape == ref["film"].shape == (12 * 8, 4)
    assert st_win["samples"] == 12 * 8 * SPP
    assert np.array_equal(win[:, 3], ref["film"][:, 3])
    assert film_rmse(win, ref["film"]) < 1e-5


def test_c1_full_size_properties_and_crop_parity(gpu, oracle):
    """BASELINE C1 (Cornell Box 400x400, sobol 64 spp, depth 5 = 10.24 M samples) the same way"""
    sc = scenes.cornell_box(gpu.bvh_build_gpu)
    ds = gpu.DeviceScene(sc)
    try:
        full, st = gpu.render(ds, scenes.cornell_render_desc(res=400, spp=64))
        acc = np.zeros_like(full)
        for r in range(4):
            acc += gpu.render(ds, scenes.cornell_render_desc(res=400, spp=64, shard=(r, 4, 64)))[0]
        rd_crop = scenes.cornell_render_desc(res=400, spp=64, crop=(0.45, 0.5, 0.25, 0.29))
        win, st_win = gpu.render(ds, rd_crop)
    finally:
        ds.close()
    assert st["samples"] == 400 * 400 * 64 and st["nan_samples"] == 0
    assert full[:, 3].min() == 64 and full[:, 3].max() <= 66
    assert np.isfinite(full).all() and full[:, :3].min() >= 0.0
    assert np.array_equal(acc[:, 3], full[:, 3]) and np.allclose(acc, full, rtol=1e-6, atol=1e-6)
    ref = oracle.render(sc, rd_crop, threads=8, want_li=False)
    assert win.shape == ref["film"].shape and st_win["samples"] == ref["counters"]["samples"]
    assert np.array_equal(win[:, 3], ref["film"][:, 3])
    assert film_rmse(win, ref["film"]) < 1e-5
