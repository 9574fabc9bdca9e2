import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/fccqp_oracle.c via ctypes) -- the checker, never the product."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def capi():
    """ctypes view of the product's C ABI (fcc_qp_amd/libfccqp_hip.so)."""
    from tests import capi as C
    return C
