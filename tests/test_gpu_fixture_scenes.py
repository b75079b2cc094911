"""Every scene of the reference-side fixture kit (tools/export_pbrt.py SCENES: what `rs_pbrt` would render for tests/golden/ref_*.npz) through the device:
the library serves each of them — none is handed back with RSPT_E_UNSUPPORTED — and agrees with the oracle the way the fixture check would ask of
the oracle itself (per-sample radiance bit for bit under Sobol' / Halton, filter weights, film)."""
import os
import sys

import numpy as np
import pytest

from rs_pbrt_amd import scenes
from tests.util import film_rmse

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
from export_pbrt import EXTRA, SCENES, camera_of, render_kwargs  # noqa: E402


@pytest.mark.parametrize("name", sorted(SCENES))
def test_fixture_scene_on_the_device_equals_the_oracle(gpu, oracle, name):
    mk, _cam, xres, yres, spp, depth = SCENES[name]
    if name == "cornell_docs":
        xres = yres = 128   # (the full frame is tests/test_gpu_reference_pin.py's)
    sc = mk(gpu.bvh_build, scenes)
    look_at, fov = camera_of(name, scenes)
    rd = scenes.make_render_desc(xres, yres, spp, look_at, fov, max_depth=depth, **render_kwargs(name, scenes))
    extra = EXTRA.get(name, {})
    if extra.get("integrator") == "directlighting":
        ref = oracle.render_integrator(sc, rd, "direct", strategy=extra.get("direct_strategy", "all"), threads=8, want_li=True)
    else:
        ref = oracle.render(sc, rd, threads=8, want_li=True)
    with gpu.DeviceScene(sc) as ds:
        film, st = gpu.render(ds, rd)
        li = gpu.render_samples(ds, rd)[0] if extra.get("sampler", "sobol") in ("sobol", "halton") else None
    assert st["samples"] == ref["counters"]["samples"] and st["nan_samples"] == ref["counters"]["nan_samples"]
    if li is not None:
        assert np.array_equal(li, ref["li"]), "per-sample radiance differs in %d samples" % int((li != ref["li"]).any(axis=2).sum())
    if "filter" in extra:
        assert np.allclose(film[:, 3], ref["film"][:, 3], rtol=1e-5)
    else:
        assert np.array_equal(film[:, 3], ref["film"][:, 3])
    assert film_rmse(film, ref["film"]) < 2e-5
