import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running pin of the oracle against a reference test configuration")


def pytest_collection_modifyitems(config, items):
    """`slow` tests (minutes each: the oracle pinned at the reference's own test sizes) run with UAMMD_RUN_SLOW=1; the default
    CPU suite keeps their reduced-size twins and finishes in a few minutes."""
    if os.environ.get("UAMMD_RUN_SLOW", "0") not in ("", "0"):
        return
    skip = pytest.mark.skip(reason="slow pin of the oracle: set UAMMD_RUN_SLOW=1")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)


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
