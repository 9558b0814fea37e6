"""pytest config: registers the `gpu` marker and makes the repo root importable.

`-m "not gpu"` covers the oracle against the reference's known-answer tests, the
host logic and the C-ABI export check; `-m gpu` are the parity tests proper and
call the HIP kernels through the C-ABI (they fail loudly without the .so).
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o
