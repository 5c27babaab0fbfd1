import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def orc():
    import orc as _orc
    _orc.lib()
    return _orc


@pytest.fixture(scope="session")
def ctx():
    """GPU context through the C ABI.  No fallback: a missing library or GPU is a failure."""
    import bzip2_amd
    c = bzip2_amd.Context(device=0)
    yield c
    c.close()
