import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        from gmmvi_amd import _lib
        return _lib.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly rather than silently skip.
    if config.getoption("-m") and "not gpu" in config.getoption("-m"):
        return
    if not any("gpu" in item.keywords for item in items):
        return
    if not _gpu_available():
        skip = pytest.mark.skip(reason="no HIP device visible")
        for item in items:
            if "gpu" in item.keywords and config.getoption("-m") != "gpu":
                item.add_marker(skip)


@pytest.fixture
def rng():
    return np.random.default_rng(1234)
