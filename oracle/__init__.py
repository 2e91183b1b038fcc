"""TEST INFRASTRUCTURE — CPU oracle for Neighbourhood Feature Pooling.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product (neighbour_feature_pooling_amd/) never does.

`forward` / `backward` call oracle/libnfp_oracle.so (plain C, double accumulation;
see nfp_oracle.c for the reference file:line of every restated step).
`unfold_torch` is an op-for-op PyTorch restatement of the reference's CPU path, used
as the timed CPU baseline on the GPU box (where /root/reference does not exist).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libnfp_oracle.so")

MEASURES = ["norm", "cosine", "dot", "rmse", "geman", "attention", "emd", "canberra", "hellinger",
            "chisquared1", "chisquared2", "gfc", "pearson", "jeffrey", "squaredchord", "smith", "scs"]
PAD_MODES = ["zeros", "reflect", "replicate", "circular"]


class Desc(ctypes.Structure):
    """Mirror of `struct nfp_desc` in include/nfp.h (kept separate from the product's
    ctypes mirror on purpose: the oracle must not import the product)."""
    _fields_ = [(n, ctypes.c_int32) for n in
                ("B", "C", "H", "W", "R", "pad", "stride", "dilation", "pad_mode", "measure",
                 "similarity", "diff_weights", "dtype")] + \
               [("p", ctypes.c_float), ("eps", ctypes.c_float), ("q_scs", ctypes.c_float)] + \
               [(n, ctypes.c_int64) for n in ("sxB", "sxC", "sxH", "sxW")]


def build(force=False):
    src = os.path.join(_HERE, "nfp_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "nfp.h")
    if (not force and os.path.exists(_SO)
            and os.path.getmtime(_SO) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-B", "libnfp_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int)
        L.nfp_oracle_output_shape.argtypes = [ctypes.POINTER(Desc), ip, ip, ip]
        L.nfp_oracle_forward.argtypes = [ctypes.POINTER(Desc), fp, fp]
        L.nfp_oracle_backward.argtypes = [ctypes.POINTER(Desc), fp, fp, fp]
        _lib = L
    return _lib


def make_desc(shape, R=1, measure="norm", p=1, stride=1, padding=0, dilation=1,
              padding_mode="reflect", similarity=True, eps=1e-6, q_scs=1e-6, strides=None):
    """`measure` is the RAW constructor string: its lower-case form picks the measure
    (nfp.py:21,85-120) while the exact string picks the conv weights (nfp.py:74)."""
    B, C, H, W = shape
    m = measure.lower()
    if m == "sharpened_cosine":
        m = "scs"
    d = Desc()
    d.B, d.C, d.H, d.W = B, C, H, W
    d.R, d.pad, d.stride, d.dilation = R, padding, stride, dilation
    d.pad_mode = PAD_MODES.index(padding_mode)
    d.measure = MEASURES.index(m)
    d.similarity = int(bool(similarity))
    d.diff_weights = int(measure in ("norm", "rmse", "mahalanobis"))
    d.dtype = 0
    d.p, d.eps, d.q_scs = float(p), float(eps), float(q_scs)
    if strides is None:
        strides = (C * H * W, H * W, W, 1)
    d.sxB, d.sxC, d.sxH, d.sxW = strides
    return d


def out_shape(d):
    n, ho, wo = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    rc = lib().nfp_oracle_output_shape(ctypes.byref(d), ctypes.byref(n), ctypes.byref(ho), ctypes.byref(wo))
    if rc:
        raise ValueError(f"oracle: invalid geometry (rc={rc})")
    return d.B, n.value, ho.value, wo.value


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def forward(x, **kw):
    """x: float32 ndarray [B,C,H,W] (contiguous NCHW) -> float32 [B,N,Ho,Wo]."""
    x, xp = _f32(x)
    d = make_desc(x.shape, **kw)
    out = np.empty(out_shape(d), np.float32)
    rc = lib().nfp_oracle_forward(ctypes.byref(d), xp, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    if rc:
        raise ValueError(f"oracle forward rc={rc}")
    return out


def backward(x, grad_out, **kw):
    x, xp = _f32(x)
    d = make_desc(x.shape, **kw)
    go, gop = _f32(grad_out)
    assert go.shape == out_shape(d), (go.shape, out_shape(d))
    gx = np.empty_like(x)
    rc = lib().nfp_oracle_backward(ctypes.byref(d), xp, gop, gx.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    if rc:
        raise ValueError(f"oracle backward rc={rc}")
    return gx
