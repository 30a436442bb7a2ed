import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


def _has_gpu():
    import waves_jl_amd as w
    try:
        return w.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_available():
    return _has_gpu()


@pytest.fixture(scope="session", autouse=True)
def _build_checkers():
    """tests may use the oracle: make sure its C restatement is compiled."""
    import c_oracle
    c_oracle.build()
    yield
