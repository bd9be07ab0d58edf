import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure).  Built on demand with g++."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    pyoracle.load()
    return pyoracle


@pytest.fixture(scope="session")
def gpu_ctx():
    """One device context for the whole GPU session.  No fallback: fails loudly without the HIP library."""
    from rivulus_amd import capi
    ctx = capi.Context(0)
    yield ctx
    ctx.close()
