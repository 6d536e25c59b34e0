import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return os.path.exists("/dev/kfd")


@pytest.fixture(scope="session")
def have_gpu():
    return _have_gpu()


@pytest.fixture(scope="session", autouse=True)
def _build_once():
    """Make sure the oracle and (if hipcc is present) the HIP library exist."""
    import __graft_entry__ as g
    g.build(quiet=True)
