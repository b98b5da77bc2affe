import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("markers", "slow: CPU tests that take many minutes (heavy fixture regeneration); they run only with "
                                       "MGLD_SLOW=1 (or `-m slow`), so that the default `-m \"not gpu\"` suite stays at a few minutes")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("MGLD_SLOW") == "1" or "slow" in (config.getoption("-m") or ""):
        return
    skip = pytest.mark.skip(reason="slow: set MGLD_SLOW=1 (or run `-m slow`) to regenerate the heavy fixtures")
    for it in items:
        if "slow" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def hip():
    """The ctypes binding with the library loaded; GPU tests call the product path through the C ABI only."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from mgld_vsr_amd import hip as _hip
    _hip.lib()
    _hip._test_ws = _hip.ensure_workspace()   # process-lifetime split-K scratch
    return _hip


@pytest.fixture(autouse=True)
def _restore_workspace(request):
    """tests may unregister the split-K scratch; put it back after every GPU test"""
    yield
    if "hip" in request.fixturenames:
        from mgld_vsr_amd import hip as _hip
        if getattr(_hip, "_test_ws", None) is not None:
            _hip.ensure_workspace()
