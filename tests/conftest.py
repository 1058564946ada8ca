import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The suites' batches are small (the oracle has to finish in seconds) and most of them are there to exercise the affine fast path, which the
# library's routing gives to batches of >= 3072 one-block reads only (smaller ones are faster on the general path, round 4).  So the suites
# run with that one rule off; tests/test_gpu_parity.py::test_small_batch_routing checks the rule itself, tools/switch_matrix.sh runs the
# suites with it on (GNX_FP_SMALL=0).
os.environ.setdefault("GNX_FP_SMALL", "1")


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
