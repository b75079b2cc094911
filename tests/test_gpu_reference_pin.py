"""-m gpu: the PRODUCT against real rs_pbrt output — the HIP path renders the Cornell box of the reference's documentation
(scenes.cornell_box_docs, tests/test_reference_pin.py has the story) through the C ABI and its film is compared with the reference's own
8-spp PNG byte by byte, next to the usual comparison with the oracle."""
import numpy as np
import pytest

from rs_pbrt_amd import scenes
from tests.test_reference_pin import G, agreement
from tests.util import film_rmse

pytestmark = pytest.mark.gpu


def test_gpu_reproduces_the_references_8spp_png(gpu, oracle):
    sc = scenes.cornell_box_docs(gpu.bvh_build)
    rd = scenes.cornell_docs_render_desc(8)
    ds = gpu.DeviceScene(sc)
    try:
        film, st = gpu.render(ds, rd)
        li = gpu.render_samples(ds, rd)[0]
    finally:
        ds.close()
    exact, w1, w4 = agreement(film, G["spp8"])
    assert exact > 0.93 and w1 > 0.95 and w4 > 0.98, (exact, w1, w4)
    ref = oracle.render(sc, rd, threads=8, want_li=True)
    assert np.array_equal(li, ref["li"])                       # a camera with a mirror in it (`Scale -1 1 1`): every sample bit for bit as the oracle's
    assert np.array_equal(film[:, 3], ref["film"][:, 3]) and film_rmse(film, ref["film"]) < 1e-7
