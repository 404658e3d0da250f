import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Compile the HIP library, the host side and the oracle once per session (hipcc cross-compiles on CPU)."""
    import __graft_entry__ as g
    g.build()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    return O


@pytest.fixture(scope="session")
def capi():
    from conga_amd import capi as c
    c.load()
    return c
