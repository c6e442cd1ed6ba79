import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


@pytest.fixture(scope="session")
def gpu_pkg(pkg):
    """The package with the HIP library loaded and a device present — fails loudly otherwise."""
    L = pkg.capi.lib()
    assert L.tbnav_device_count() > 0, "gpu test selected but no HIP device is visible"
    return pkg
