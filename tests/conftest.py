import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "graph-embeddings_amd"))   # import geglove  (product host mirror)
sys.path.insert(0, os.path.join(REPO, "oracle"))                  # import oracle   (checker; tests only)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))    # import helpers


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _build_if_missing()


def _build_if_missing():
    """The shared libraries are build products (git-ignored).  A fresh checkout builds them once, here;
    on the GPU box they arrive pre-built with the snapshot (hipcc cross-compiles gfx950 without a GPU)."""
    import subprocess
    pkg = os.path.join(REPO, "graph-embeddings_amd")
    need = [(os.path.join(pkg, "lib", "libgeglove.so"), os.path.join(pkg, "csrc")),
            (os.path.join(pkg, "lib", "libgehost.so"), os.path.join(pkg, "host")),
            (os.path.join(REPO, "oracle", "libge_oracle.so"), os.path.join(REPO, "oracle"))]
    for lib, src in need:
        if not os.path.exists(lib):
            subprocess.check_call(["make", "-C", src, "-j4"])


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

# rank-thread tests: a rank that never reaches an exchange aborts its group after three minutes instead of holding the run
os.environ.setdefault("GE_LOCAL_GROUP_TIMEOUT_S", "180")
