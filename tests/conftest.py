"""pytest configuration: `-m gpu` tests need a gfx950 device and call the HIP path through the
C ABI (rs_pbrt_amd.lib); everything else runs on the CPU.  oracle/ is imported only from tests."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (gfx950); run with -m gpu")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def gpu():
    """Initialised librspt on device 0.  Fails (does not skip) when the HIP extension or the
    device is missing: a silent fallback would void the parity claims."""
    from rs_pbrt_amd import lib
    lib.init(0)
    yield lib
    lib.shutdown()
