import os
import sys

import pytest

try:  # PyTorch ships its own ROCm runtime: load it before libclsplace.so pulls in the system's, whatever test runs first
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the host libraries (generator, C oracle, and the HIP library if it is
    missing -- hipcc cross-compiles gfx950 without a GPU)."""
    import __graft_entry__ as g

    g.build(only_missing=True)
