import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "graph-embeddings_amd"))   # import geglove  (product host mirror)
sys.path.insert(0, os.path.join(REPO, "oracle"))                  # import oracle   (checker; tests only)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))    # import helpers


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        from geglove import capi
        return capi.lib().ge_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """GPU tests FAIL (not skip) when the HIP library or the device is missing."""
    from geglove import capi
    n = capi.lib().ge_device_count()
    assert n > 0, "no gfx950 device visible: %s" % capi.lib().ge_last_error().decode()
    return n
