import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    """tests/golden/<name>.npz -> dict of torch tensors (made by tests/golden/make_golden.py)."""
    with np.load(os.path.join(GOLDEN, name + '.npz')) as f:
        return {k: torch.from_numpy(f[k]) for k in f.files}


def sub(d, prefix):
    """Strip a 'prefix/' namespace from a golden dict."""
    n = len(prefix)
    return {k[n:]: v for k, v in d.items() if k.startswith(prefix)}


@pytest.fixture(scope='session')
def golden():
    return load_golden


def have_gpu():
    return torch.cuda.is_available()
