import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Routing of small batches.  The library gives batches of < 3 072 one-block reads to the general path (round 4: what a loop of single
# align.AffineGap calls goes through); the suites' batches are small -- the oracle has to finish in seconds -- and most of them exist to
# exercise the affine fast path.  So:
#   * the parity suites of tests/test_gpu_parity.py run TWICE, ids "shipped" (GNX_FP_SMALL unset: the routing a user gets) and "fp_small"
#     (GNX_FP_SMALL=1: the rule off, the headline's kernels on every batch);
#   * every other test runs with the rule off, as before;
#   * a GNX_FP_SMALL set from OUTSIDE (tools/switch_matrix.sh) wins: no parametrisation, the given value everywhere.
_OUTER_FP_SMALL = os.environ.get("GNX_FP_SMALL")
_DUAL_MODULES = ("test_gpu_parity",)
_DUAL_EXCEPT = ("test_small_batch_routing",)  # (sets the switch itself, both ways)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun); everything else runs on CPU")


def pytest_generate_tests(metafunc):
    if _OUTER_FP_SMALL is None and metafunc.module.__name__ in _DUAL_MODULES and metafunc.function.__name__ not in _DUAL_EXCEPT and "routing" in metafunc.fixturenames:
        metafunc.parametrize("routing", ["shipped", "fp_small"], indirect=True)


@pytest.fixture(autouse=True)
def routing(request, monkeypatch):
    """which small-batch routing the test runs under: "shipped" (the library's own) or "fp_small" (GNX_FP_SMALL=1)"""
    which = getattr(request, "param", "fp_small")
    if _OUTER_FP_SMALL is None:
        if which == "fp_small":
            monkeypatch.setenv("GNX_FP_SMALL", "1")
        else:
            monkeypatch.delenv("GNX_FP_SMALL", raising=False)
    return which


@pytest.fixture(scope="session")
def gpu_lib():
    """The HIP library bound to cuda:0 -- GPU tests fail (not skip) when it is missing."""
    from gonomics_amd import _lib
    L = _lib.lib()
    assert L.gnx_device_count() > 0, "no HIP device visible"
    _lib.check(L.gnx_init(0, 0))
    return _lib
