import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "fp8-quantization_amd")
for p in (ROOT, PKG, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # gpu-marked tests are never silently skipped on a GPU box; on a CPU-only box they are
    # deselected by `-m "not gpu"`, and skipped (loudly) if someone runs them anyway.
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible (torch.cuda.is_available() is False)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
