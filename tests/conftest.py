import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A plain `pytest` on a host without a GPU skips the GPU tests instead of failing them (the product path has no
    CPU fallback, so they cannot run there)."""
    gpu_items = [it for it in items if "gpu" in it.keywords]
    if not gpu_items:
        return
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU: the mel-inversion path has no CPU fallback")
    for it in gpu_items:
        it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
