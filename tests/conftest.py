import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """Without a GPU the `gpu`-marked tests are SKIPPED (not failed), so a CPU run of the whole suite separates real
    regressions from a missing device.  On a GPU box nothing is skipped: the ops raise if the HIP library is absent."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (torch.cuda.is_available() is False)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def hostsim():
    """g++ build of trase_amd/csrc/gs_math.h behind a C shim (tests/hostsim/hostsim.cpp)."""
    from tests.util import build_hostsim
    return build_hostsim()
