import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def golden_meta():
    import json
    with open(os.path.join(GOLDEN, 'meta.json')) as f:
        return json.load(f)


def load_golden(name):
    import numpy as np
    import torch
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    return {k: torch.from_numpy(z[k]) for k in z.files}
