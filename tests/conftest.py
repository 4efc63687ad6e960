"""pytest configuration: `gpu` marker for tests that need a real MI355X."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a HIP device (run on the MI355X box with -m gpu)')


@pytest.fixture
def rng():
    return np.random.default_rng(12345)


@pytest.fixture(scope='session')
def bb():
    """The HIP block backend; GPU tests fail loudly (not skip) if the extension is missing."""
    import torch
    if not torch.cuda.is_available():
        # `-m gpu` was asked for on a machine without a device: an error, never a silent green run of skips
        pytest.fail('GPU tests selected (-m gpu) but no HIP device is visible: run them on the MI355X box', pytrace=False)
    from cyten_amd.block_backend import HipBlockBackend
    return HipBlockBackend('cuda:0')
