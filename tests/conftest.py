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


@pytest.fixture(autouse=True)
def _nfp_env_at_test_start():
    """libnfp_hip.so reads its NFP_* A/B switches once; tests flip them with `nfp_switch`.  Start every test from
    the environment as it is now (the previous test's monkeypatch has been undone by then)."""
    from neighbour_feature_pooling_amd import _abi
    if _abi._lib is not None:
        _abi._lib.nfp_reload_env()
    yield


def nfp_switch(monkeypatch, name, value):
    """Set (value=None: unset) one NFP_* switch for the rest of the test and make the library re-read them."""
    from neighbour_feature_pooling_amd import _abi
    if value is None:
        monkeypatch.delenv(name, raising=False)
    else:
        monkeypatch.setenv(name, str(value))
    _abi.load().nfp_reload_env()
