import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


# NFP_TEST_LIB=<file in the package directory>: run the suite on another build of the same sources (the LDS-poison test
# build, tests/test_gpu_tile.py::test_row_band_kernels_with_poisoned_lds).  The C++ autograd nodes link the product
# library, so the Python nodes serve (they call whatever _abi loads).
if os.environ.get("NFP_TEST_LIB"):
    os.environ["NFP_PY_NODES"] = "1"
    from neighbour_feature_pooling_amd import _abi as _abi_for_test_lib
    _abi_for_test_lib.LIB_PATH = os.path.join(ROOT, "neighbour_feature_pooling_amd", os.environ["NFP_TEST_LIB"])
    assert os.path.exists(_abi_for_test_lib.LIB_PATH), _abi_for_test_lib.LIB_PATH


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


def golden_out_shape(g):
    return tuple(g["out"].shape) if "out" in g else tuple(int(v) for v in g["out_shape"])


def assert_matches_golden(out, gx, g, tol, tol_gx=None):
    """`out` / `gx` (numpy, the case's full tensors; gx may be None) against a fixture of tests/golden/.  Small
    tensors are stored in full; large ones as every 97th element plus per-image (gx) / per-map (out) sums and
    absolute sums of the reference's tensor (cases.py: full_limit)."""
    import cases as K
    tol_gx = tol if tol_gx is None else tol_gx

    def sampled(a, sample, s_ref, a_ref, axes, t):
        assert rel_err(a.reshape(-1)[K.gx_sample_index(a.size)], sample) <= t
        s = a.astype(np.float64).sum(axis=axes)
        assert np.max(np.abs(s - s_ref) / a_ref) <= t

    assert tuple(out.shape) == golden_out_shape(g)
    if "out" in g:
        assert same_nan_pattern(out, g["out"])
        assert rel_err(np.nan_to_num(out), np.nan_to_num(g["out"])) <= tol
    else:
        sampled(out, g["out_sample"], g["out_sum"], g["out_abs_sum"], (2, 3), tol)
    if gx is None:
        return
    if "gx" in g:
        assert same_nan_pattern(gx, g["gx"])
        assert rel_err(np.nan_to_num(gx), np.nan_to_num(g["gx"])) <= tol_gx
    else:
        sampled(gx, g["gx_sample"], g["gx_sum"], g["gx_abs_sum"], (1, 2, 3), tol_gx)


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
