import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with np.load(os.path.join(ROOT, "tests", "golden", "kernels.npz")) as f:
        return {k: f[k] for k in f.files}


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (our C restatement).  Checker only."""
    from oracle.oracle import Oracle
    return Oracle("port")


@pytest.fixture(scope="session")
def hip():
    """The product library through its C-ABI (ctypes).  Fails loudly if it is missing."""
    from massivedatans_amd import _lib
    return _lib.load()
