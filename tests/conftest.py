import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

# physical parameter sets used by the golden fixtures (tests/golden/make_golden.py)
PARAM_SETS = {
    "default": (1.0, 1.0, 10.0, 1e-3),
    "realworld": (0.8, 1.2, 10.2, 1e-3),
    "odd": (1.3, 0.7, 4.5, 2.5e-3),
}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    class G(object):
        def __getattr__(self, name):
            return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return G()
