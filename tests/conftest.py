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
    # The oracle runs on torch CPU.  On a small shared build container OpenMP teams get
    # their vCPUs preempted mid-barrier and a 40 s suite turns into 7 minutes; the
    # tensors are small, so one thread is both the fastest and the only predictable
    # choice there.  The GPU box (hundreds of idle cores) keeps a modest team.
    import torch
    torch.set_num_threads(8 if torch.cuda.is_available() else 1)


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
