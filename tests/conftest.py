import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return graft.load_package()


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def gpu_pkg(pkg):
    """The product package with the HIP library loaded and a device present; GPU tests fail (not skip) without it."""
    pkg.lib()
    assert pkg.lib().apds_device_count() > 0, "no HIP device: -m gpu tests must run on the GPU box"
    return pkg
