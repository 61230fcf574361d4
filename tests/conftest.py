import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_small():
    import numpy as np
    return np.load(os.path.join(REPO, "tests", "golden", "golden_small.npz"))


@pytest.fixture(scope="session")
def golden_vitb16():
    import numpy as np
    return np.load(os.path.join(REPO, "tests", "golden", "golden_vitb16.npz"))


@pytest.fixture(scope="session")
def golden_vitb32():
    import numpy as np
    return np.load(os.path.join(REPO, "tests", "golden", "golden_vitb32.npz"))


@pytest.fixture(scope="session")
def golden_vitl14():
    import numpy as np
    return np.load(os.path.join(REPO, "tests", "golden", "golden_vitl14_336.npz"))


def oracle_clip():
    """The CPU oracle's `clip` stand-in (tests only)."""
    import importlib
    return importlib.import_module("oracle.clip")
