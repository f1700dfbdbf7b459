import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope='session')
def hip():
    """The loaded C-ABI library on a GPU box; GPU tests must run the HIP path, never a fallback."""
    import torch
    from deep_cbrs_amar_renaissance_amd import capi
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    capi.load()
    return capi


@pytest.fixture(scope='session')
def ml1m_s1():
    """ml1m(s=1): indexed ratings, ids, UI and UIP adjacency (host arrays), built once per session."""
    from tests import helpers
    return helpers.ml1m_indexed(1)
