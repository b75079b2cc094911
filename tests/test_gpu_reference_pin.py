"""-m gpu: the PRODUCT against real rs_pbrt output — the HIP path renders the Cornell box of the reference's documentation
(scenes.cornell_box_docs, tests/test_reference_pin.py has the story) through the C ABI and its film is compared with the reference's own
8-spp PNG byte by byte, next to the usual comparison with the oracle."""
import numpy as np
import pytest

from rs_pbrt_amd import scenes
from tests.test_reference_pin import G, agreement, to_u8
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
    assert exact > 0.94 and w1 > 0.957 and w4 > 0.985, (exact, w1, w4)
    # the strict bar (tests/test_reference_pin.py, last test): where the reference's picture carries no extra contribution the PRODUCT's bytes are the reference's
    ours, refpng = to_u8(scenes.film_to_rgb(film)).reshape(500, 500, 3), G["spp8"].astype(np.int32)
    no_extra = (refpng - ours).max(-1) <= 0
    assert no_extra.mean() > 0.94 and (ours == refpng).all(-1)[no_extra].mean() >= 0.998 and ((ours - refpng).max(-1) > 1).mean() < 0.0005
    ref = oracle.render(sc, rd, threads=8, want_li=True)
    assert np.array_equal(li, ref["li"])                       # a camera with a mirror in it (`Scale -1 1 1`): every sample bit for bit as the oracle's
    assert np.array_equal(film[:, 3], ref["film"][:, 3]) and film_rmse(film, ref["film"]) < 1e-7


def test_gpu_reproduces_the_references_256spp_png(gpu):
    """the whole 256-spp frame (64 M paths, 55 ms on the MI355X; the oracle needs 35 s on 8 cores, so the CPU suite looks at every sixth tile):
    measured 0.7104 of the pixels byte-equal, 0.9888 within 1 / 255, 1.0000 within 4 / 255 (profiles/r03_reference_pin_gpu.txt)"""
    ds = gpu.DeviceScene(scenes.cornell_box_docs(gpu.bvh_build))
    try:
        film, _ = gpu.render(ds, scenes.cornell_docs_render_desc(256))
    finally:
        ds.close()
    exact, w1, w4 = agreement(film, G["spp256"])
    assert exact > 0.65 and w1 > 0.98 and w4 > 0.9995, (exact, w1, w4)
