"""Pin the CPU oracle (oracle/nfp_oracle.c) against the real reference's outputs.

Every fixture in tests/golden/ was produced by tests/golden/make_golden.py from
/root/reference/models/pooling/nfp.py (torch 2.10 CPU, fp32).  The oracle computes
in double, so the residual is the REFERENCE's own fp32 rounding: bound 2e-6 relative
to the tensor's max (observed <= 5e-7), 10x inside the 1e-5 parity bar.
"""
import numpy as np
import pytest

import cases as K
from conftest import assert_matches_golden, load_golden, rel_err

TOL = 2e-6


@pytest.mark.parametrize("name", [c["name"] for c in K.CASES])
def test_oracle_matches_reference(name, oracle_lib):
    c = K.BY_NAME[name]
    g = load_golden(name)
    x = K.make_input(c)
    out = oracle_lib.forward(x, **c["ctor"])
    go = K.make_grad_out(c, out.shape)
    gx = oracle_lib.backward(x, go, **c["ctor"])
    assert_matches_golden(out, gx, g, TOL)
    if "gx_abs_sum" in g:
        a = np.abs(gx.astype(np.float64)).sum(axis=(1, 2, 3))
        assert np.max(np.abs(a - g["gx_abs_sum"]) / g["gx_abs_sum"]) <= TOL


def pooled_by_oracle(oracle_lib, c):
    """adaptive_avg_pool2d(NFP(x), 1) and its input gradient from the oracle: the mean of the maps, and the backward
    of the maps under grad_out[b,n,:,:] = g[b,n] / (Ho Wo) (the adjoint of the mean)."""
    x = K.make_input(c)
    out = oracle_lib.forward(x, **c["ctor"])
    g = K.make_pool_grad(c, out.shape[1])
    go = np.broadcast_to((g / (out.shape[2] * out.shape[3]))[:, :, None, None], out.shape).astype(np.float32)
    return out.astype(np.float64).mean(axis=(2, 3)), oracle_lib.backward(x, go, **c["ctor"]), g


def assert_pooled_matches_golden(nfpm, gx, g, tol, tol_gx=None):
    tol_gx = tol if tol_gx is None else tol_gx
    assert rel_err(nfpm, g["nfpm"]) <= tol
    if "gx" in g:
        assert rel_err(gx, g["gx"]) <= tol_gx
    else:
        assert rel_err(gx.reshape(-1)[K.gx_sample_index(gx.size)], g["gx_sample"]) <= tol_gx
        s = gx.astype(np.float64).sum(axis=(1, 2, 3))
        assert np.max(np.abs(s - g["gx_sum"]) / g["gx_abs_sum"]) <= tol_gx


@pytest.mark.parametrize("name", [c["name"] for c in K.POOL_CASES])
def test_oracle_matches_reference_pooled(name, oracle_lib):
    """models/texture_pooling.py:251-252, 320-321: only adaptive_avg_pool2d(NFP(feat), 1) is consumed."""
    c = K.POOL_BY_NAME[name]
    nfpm, gx, _ = pooled_by_oracle(oracle_lib, c)
    assert_pooled_matches_golden(nfpm, gx, load_golden(name), TOL)


def test_oracle_strided_input_equals_contiguous(oracle_lib):
    """channels-last strides give the same answer as NCHW (the oracle reads by strides)."""
    import ctypes
    c = K.BY_NAME["geo_cos_nonsquare"]
    x = K.make_input(c)
    B, C, H, W = x.shape
    ref = oracle_lib.forward(x, **c["ctor"])
    xl = np.ascontiguousarray(x.transpose(0, 2, 3, 1))  # NHWC memory
    d = oracle_lib.make_desc(x.shape, strides=(H * W * C, 1, W * C, C), **c["ctor"])
    out = np.empty_like(ref)
    fp = ctypes.POINTER(ctypes.c_float)
    rc = oracle_lib.lib().nfp_oracle_forward(ctypes.byref(d), xl.ctypes.data_as(fp), out.ctypes.data_as(fp))
    assert rc == 0
    assert np.array_equal(out, ref)


@pytest.mark.parametrize("name", ["c1_cos_k3_2x64x14x14", "c2_cos_k3_4x512x7x7", "c3_cos_k3_8x512x2x2",
                                  "c5_l2_k5_4x192x14x14", "geo_cos_stride2_dil2_pad0", "geo_l2_zeros",
                                  "edge_cos_tiny_norms", "edge_cos_dissimilarity"])
def test_unfold_torch_restatement_matches_reference(name):
    """oracle/unfold_torch.py (bench.py's timed CPU baseline) reproduces the reference."""
    import torch
    from oracle.unfold_torch import UnfoldNFP
    c = K.BY_NAME[name]
    g = load_golden(name)
    ctor = dict(c["ctor"])
    x = torch.from_numpy(K.make_input(c)).requires_grad_(True)
    m = UnfoldNFP(c["shape"][1], **ctor)
    out = m(x)
    assert rel_err(out.detach().numpy(), g["out"]) <= 1e-6
    out.backward(torch.from_numpy(K.make_grad_out(c, tuple(out.shape))))
    gx = x.grad.numpy()
    if "gx" in g:
        assert rel_err(gx, g["gx"]) <= 1e-6
    else:
        assert rel_err(gx.reshape(-1)[K.gx_sample_index(gx.size)], g["gx_sample"]) <= 1e-6
