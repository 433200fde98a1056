import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture
def rng():
    # seed borrowed from the reference's tests/conftest.py:22
    return np.random.default_rng(71892305)


@pytest.fixture(params=[(3,), (), (2, 1)], ids=["b3", "b0", "b21"])
def batch_shape(request):
    """reference tests/conftest.py:39-43"""
    return request.param


def golden(name):
    return np.load(os.path.join(GOLDEN, name))
