import os
import sys

import pytest

try:   # PyTorch-ROCm wheels bundle their own HIP runtime: load it before libhnsw_mi355x.so pulls in the
    import torch  # noqa: F401  system one, so that a process ends up with ONE runtime whatever the test order
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU restatement of the reference (test infrastructure)."""
    from oracle import oracle as o
    o.build()
    o.lib()
    return o


def load_golden(name):
    import json
    with open(os.path.join(ROOT, "tests", "golden", name)) as f:
        return json.load(f)
