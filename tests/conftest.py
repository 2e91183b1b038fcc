import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def rel_err(a, ref):
    """max|a-ref| / max|ref| — the 'relative to the tensor' error the parity bar is stated in."""
    a = np.asarray(a, np.float64)
    ref = np.asarray(ref, np.float64)
    den = np.max(np.abs(ref))
    return float(np.max(np.abs(a - ref)) / (den if den > 0 else 1.0))


def same_nan_pattern(a, ref):
    return np.array_equal(np.isnan(a), np.isnan(ref))


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle
    oracle.build()
    return oracle
