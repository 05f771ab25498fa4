import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_available():
    import torch

    return torch.cuda.is_available()


def pytest_collection_modifyitems(config, items):
    # A GPU test must never pass silently on a box without a GPU: without one it errors loudly
    # unless it was deselected with -m "not gpu".
    pass
