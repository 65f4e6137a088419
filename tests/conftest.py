import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


import pytest


@pytest.fixture(autouse=True)
def _drain_gpu_after_test():
    """GPU tests leave no queued work behind: a kernel of a finished test must not run into the next test's freshly allocated tensors"""
    yield
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    except Exception:
        pass
