import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'soft-robot-control_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # torch ships its own HIP runtime; where a test uses torch CUDA tensors next to libsofacontrol_hip (the sharded POD
    # build), torch has to come up first -- once the library's runtime owns the device, torch finds "no HIP GPUs"
    if 'gpu' in (config.getoption('-m') or '') and 'not gpu' not in (config.getoption('-m') or ''):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + '.npz'))
    return load
