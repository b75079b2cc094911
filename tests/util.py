"""Shared helpers for the parity tests."""
import numpy as np

from rs_pbrt_amd import abi, scenes


def random_rays(n, seed, lo, hi, t_max=np.inf):
    rng = np.random.default_rng(seed)
    rays = np.zeros(n, abi.RAY_DT)
    rays["o"] = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    rays["d"] = (d / np.linalg.norm(d, axis=1)[:, None]).astype(np.float32)
    rays["t_max"] = t_max
    rays["id"] = np.arange(n, dtype=np.uint32)
    return rays


def film_rmse(film_a, film_b):
    """per-pixel RMSE over linear RGB of contrib_sum / filter_weight_sum (film.rs:452-462), BASELINE.md §2.4"""
    a, b = scenes.film_to_rgb(film_a), scenes.film_to_rgb(film_b)
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


def small_soup(builder, n=20000, seed=0x5EED5EED):
    return scenes.triangle_soup(builder, n_tris=n, seed=seed, extent=0.03)
