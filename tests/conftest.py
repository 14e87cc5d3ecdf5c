import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'ml-pointconvformer_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    """-> dict of torch tensors (ints stay int64, floats fp32) from tests/golden/<name>.npz."""
    z = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    return {k: torch.from_numpy(z[k]) for k in z.files}


def split(blobs, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in blobs.items() if k.startswith(prefix)}


@pytest.fixture(scope='session')
def device():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')
