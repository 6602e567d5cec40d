import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running pin of the oracle against a reference test configuration")


@pytest.fixture(scope="session")
def o32():
    import oracle
    return oracle.get("f32")


@pytest.fixture(scope="session")
def o64():
    import oracle
    return oracle.get("f64")


@pytest.fixture(scope="session")
def hip():
    """The product library; on a GPU box a missing/unloadable library is a hard failure."""
    import uammd_amd
    uammd_amd.load()
    return uammd_amd
