import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def built():
    """The in-tree libraries; built on demand when missing (the .so files are git-ignored)."""
    import __graft_entry__ as g
    from particlesmc_amd import _lib
    from oracle import oracle as O
    if not os.path.exists(_lib.SO_PATH):
        g.build()
    O.build()
    return True


@pytest.fixture(scope="session")
def gpu(built):
    import particlesmc_amd as pmc
    if pmc.device_count() < 1:
        pytest.fail("no HIP device visible: the -m gpu tests must run on a GPU box")
    return pmc
