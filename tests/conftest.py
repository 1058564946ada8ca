import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun); everything else runs on CPU")


@pytest.fixture(scope="session")
def gpu_lib():
    """The HIP library bound to cuda:0 -- GPU tests fail (not skip) when it is missing."""
    from gonomics_amd import _lib
    L = _lib.lib()
    assert L.gnx_device_count() > 0, "no HIP device visible"
    _lib.check(L.gnx_init(0, 0))
    return _lib
