import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_collection_modifyitems(config, items):
    """``gpu``-marked tests are skipped (not failed) on a box without a GPU, e.g. a plain ``pytest tests`` in the build container."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (torch.cuda.is_available() is False)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
