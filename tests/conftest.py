import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parents[1]
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

GOLDEN = REPO / 'tests' / 'golden'


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run by the '
                            'driver with `-m gpu` on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


def gpu_available():
    import torch
    return torch.cuda.is_available()


@pytest.fixture(scope='session')
def gpu():
    """cuda:0, or a hard failure when a `-m gpu` test runs without one (a
    silent skip would hide a missing device on the GPU box)."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail('GPU test selected but torch.cuda.is_available() is False')
    return torch.device('cuda', 0)
