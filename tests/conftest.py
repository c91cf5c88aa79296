import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


def pytest_sessionstart(session):
    """The CPU oracle's torch ops default to one thread per visible core -- 256 on a GPU box whose share of the host is 16
    cores: oversubscribed, slow and noisy (the same suite took 384 s on one box and 529 s on another).  Cap the pool."""
    import torch
    torch.set_num_threads(int(os.environ.get('W2L_TEST_THREADS', min(16, os.cpu_count() or 1))))
