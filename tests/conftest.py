import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "ka9q-radio_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip():
    """libka9q_hip on device 0.  Only GPU tests ask for it; it fails loudly when there is no device."""
    import pyka9q
    L = pyka9q.lib()
    assert L.ka9q_hip_init(0) == 0, pyka9q.last_error()
    return pyka9q


def rel_rms(a, b):
    import numpy as np
    a = np.asarray(a)
    b = np.asarray(b)
    d = np.linalg.norm((a - b).ravel())
    n = np.linalg.norm(b.ravel())
    return float(d / n) if n > 0 else float(d)
