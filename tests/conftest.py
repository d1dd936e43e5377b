import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def have_gpu():
    return _have_gpu()


@pytest.fixture(scope="session", autouse=True)
def _torch_hip_runtime_first():
    """torch bundles its own libamdhip64; when a GPU is present initialise it before
    libssx_hip.so (which links /opt/rocm's) creates its contexts, the order bench.py uses too."""
    if _have_gpu():
        import torch
        torch.cuda.init()
    yield
