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
    # torch first: its wheel brings its own HIP runtime, and a process in which libapds_hip.so has already initialised the system
    # one leaves torch without devices ("No HIP GPUs are available"); loaded in this order both use the same runtime, as in bench.py
    import torch
    assert torch.cuda.is_available(), "no HIP device: -m gpu tests must run on the GPU box"
    torch.cuda.init()
    pkg.lib()
    assert pkg.lib().apds_device_count() > 0, "no HIP device: -m gpu tests must run on the GPU box"
    return pkg
