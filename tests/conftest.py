import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def O():
    from _libs import oracle
    return oracle()


@pytest.fixture(scope="session")
def ga():
    """The product library through its C ABI; initialises the GPU (gpu tests only)."""
    import libgoldilocks_amd as ga
    ga.lib()
    return ga
