import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)

    return load


def rel_err(x, ref):
    """(max-norm relative error, L2 relative error) — the parity metrics of SURVEY.md §8d."""
    x = np.asarray(x, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert x.shape == ref.shape, (x.shape, ref.shape)
    den_max = max(np.abs(ref).max(), 1e-300)
    den_l2 = max(np.linalg.norm(ref), 1e-300)
    return np.abs(x - ref).max() / den_max, np.linalg.norm(x - ref) / den_l2
