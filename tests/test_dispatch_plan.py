"""Dispatch rules of libnfp_hip.so, checked WITHOUT a GPU through nfp_plan (include/nfp.h): which kernel serves
which geometry, and that no descriptor the library accepts produces a launch outside the device's limits."""
import ctypes
import random
import re

import pytest

from neighbour_feature_pooling_amd import _abi
from neighbour_feature_pooling_amd.build import build_hip

LDS_MAX = 160 * 1024


@pytest.fixture(scope="module")
def lib():
    build_hip()
    return _abi.load()


def desc(shape, R=1, pad=None, stride=1, dil=1, mode="reflect", measure="cosine", p=2.0, dtype=_abi.F32,
         channels_last=False):
    B, C, H, W = shape
    d = _abi.NfpDesc()
    d.B, d.C, d.H, d.W = B, C, H, W
    d.R, d.pad, d.stride, d.dilation = R, R if pad is None else pad, stride, dil
    d.pad_mode = _abi.PAD_MODES.index(mode)
    d.measure = _abi.measure_id(measure)
    d.similarity, d.diff_weights, d.dtype = 1, int(measure in ("norm", "rmse")), dtype
    d.p, d.eps, d.q_scs = p, 1e-6, 1e-6
    if channels_last:
        d.sxB, d.sxC, d.sxH, d.sxW = C * H * W, 1, W * C, C
    else:
        d.sxB, d.sxC, d.sxH, d.sxW = C * H * W, H * W, W, 1
    return d


def plan(lib, d, backward):
    buf = ctypes.create_string_buffer(1024)
    rc = lib.nfp_plan(ctypes.byref(d), int(backward), buf, len(buf))
    return rc, buf.value.decode()


def launches(text):
    return [(m.group(1), tuple(int(v) for v in m.group(2).split(",")), int(m.group(3)), int(m.group(4)))
            for m in re.finditer(r"(\w+) grid=\(([\d,]+)\) block=(\d+) lds=(\d+)", text)]


def test_plan_does_not_launch_or_disturb_last_variant(lib):
    before, n0 = lib.nfp_last_variant(), lib.nfp_launch_count()
    rc, text = plan(lib, desc((64, 512, 7, 7)), False)
    assert rc == 0 and text.startswith("fwd_band<R1,cos,f32,nchw>x4 | fwd_band grid=(64,4,1)")
    assert lib.nfp_launch_count() == n0 and lib.nfp_last_variant() == before


@pytest.mark.parametrize("d_kw,fwd,bwd", [
    (dict(shape=(64, 512, 7, 7)), "fwd_band<R1,cos,f32,nchw>x4", "bwd_fast<R1,cos,f32,nchw>"),         # headline: 4 row bands
    (dict(shape=(256, 512, 7, 7)), "fwd_band<R1,cos,f32,nchw>x1", "bwd_fast<R1,cos,f32,nchw>"),        # config 4: whole images
    (dict(shape=(256, 192, 14, 14), R=2, measure="norm", dtype=_abi.BF16), "fwd_gram<R2,l2,bf16,nchw>",
     "bwd_fast<R2,l2,bf16,nchw,mfma2>"),                                                                 # config 5
    (dict(shape=(256, 200, 14, 14), R=2, measure="norm", dtype=_abi.BF16), "fwd_band<R2,l2,bf16,nchw>x1",
     "bwd_fast<R2,l2,bf16,nchw>"),                                                                      # C % 16 != 0
    (dict(shape=(8, 512, 7, 7), channels_last=True), "fwd_band<R1,cos,f32,nhwc>x7", "bwd_fast<R1,cos,f32,nhwc>"),
    (dict(shape=(256, 192, 14, 14), R=2, measure="norm", dtype=_abi.BF16, channels_last=True),
     "fwd_gram<R2,l2,bf16,nhwc>", "bwd_fast<R2,l2,bf16,nhwc,mfma2>"),                                  # ViT tokens: matrix cores, second form (all 7 row tiles' weights in LDS, x in 3 chunks)
    (dict(shape=(256, 512, 7, 7), dtype=_abi.BF16), "fwd_gram<R1,cos,bf16,nchw>", "bwd_fast<R1,cos,bf16,nchw>"),   # odd rows, no channel split
    (dict(shape=(64, 512, 7, 7), dtype=_abi.BF16), "fwd_gram<R1,cos,bf16,nchw>", "bwd_fast<R1,cos,bf16,nchw,mfma>"),
    # the matrix-core backward's loop-free phase A: at most four gather rounds of its (up to 1024) threads
    (dict(shape=(4, 64, 16, 16), R=2, measure="norm", dtype=_abi.BF16, channels_last=True),
     "fwd_gram<R2,l2,bf16,nhwc>", "bwd_fast<R2,l2,bf16,nhwc,mfma>"),                                     # 256 x 13 = 3328 entries; Wd of all 8 row tiles does not fit: round 3's form
    (dict(shape=(4, 64, 18, 18), R=2, measure="norm", dtype=_abi.BF16, channels_last=True),
     "fwd_gram<R2,l2,bf16,nhwc>", "bwd_fast<R2,l2,bf16,nhwc>"),                                         # 324 x 13 = 4212: vector kernel
    (dict(shape=(64, 512, 7, 7), measure="gfc", dtype=_abi.BF16, channels_last=True),
     "fwd_band<R1,gfc,bf16,nhwc>x4", "bwd_fast<R1,gfc,bf16,nhwc,mfma>"),
    (dict(shape=(64, 512, 7, 7), measure="norm", p=1.0), "fwd_band<R1,l1,f32,nchw>x4", "bwd_fast<R1,l1,f32,nchw>"),   # reference default p: table kernels since round 3
    (dict(shape=(64, 512, 7, 7), measure="norm", p=3.0), "fwd_pairs", "bwd_gather"),                    # any other order
    (dict(shape=(64, 512, 7, 7), pad=0), "fwd_pairs", "bwd_gather"),                                    # pad != R
    (dict(shape=(64, 512, 7, 7), stride=2), "fwd_pairs", "bwd_gather"),
    (dict(shape=(64, 510, 7, 7)), "fwd_pairs", "bwd_gather"),                                           # C % 4 != 0
    (dict(shape=(4, 64, 7, 7), mode="circular"), "fwd_pairs", "bwd_gather"),
    (dict(shape=(4, 64, 7, 7), measure="jeffrey"), "fwd_band<R1,jeffrey,f32,nchw>x7", "bwd_fast<R1,jeffrey,f32,nchw>"),   # round 4: the shared symmetric-term instantiation
    (dict(shape=(4, 64, 7, 7), measure="smith"), "fwd_pairs", "bwd_gather"),
    (dict(shape=(64, 512, 7, 7), measure="dot"), "fwd_band<R1,dot,f32,nchw>x4", "bwd_fast<R1,dot,f32,nchw>"),   # round 3: on the product kernels
    (dict(shape=(64, 512, 7, 7), measure="gfc"), "fwd_band<R1,gfc,f32,nchw>x4", "bwd_fast<R1,gfc,f32,nchw>"),
    (dict(shape=(64, 512, 7, 7), measure="rmse"), "fwd_band<R1,rmse,f32,nchw>x4", "bwd_fast<R1,rmse,f32,nchw>"),  # ... and the L2 kernels
    (dict(shape=(4, 64, 7, 7), measure="attention"), "fwd_band<R1,dot,f32,nchw>x7+attn_softmax", "bwd_fast<R1,dot,f32,nchw>"),   # round 4: DotProduct's hot kernels under the softmax (float32 maps)
    (dict(shape=(256, 64, 56, 56), measure="attention"), "fwd_tile<R1,dot,f32,nchw>x10+attn_softmax", "bwd_tile<R1,dot,f32,nchw>x10"),
    (dict(shape=(4, 64, 7, 7), measure="attention", dtype=_abi.BF16), "fwd_pairs+attn_softmax", "bwd_gather"),                   # bf16 maps: float32 dots in the scratch, any-geometry kernels
    (dict(shape=(64, 64, 56, 56)), "fwd_tile<R1,cos,f32,nchw>x10", "bwd_tile<R1,cos,f32,nchw>x10"),     # > 512 px: row bands
    (dict(shape=(256, 16, 112, 112)), "fwd_tile<R1,cos,f32,nchw>x19", "bwd_tile<R1,cos,f32,nchw>x19"),
    (dict(shape=(2, 8, 100, 140)), "fwd_tile<R1,cos,f32,nchw>x50", "bwd_tile<R1,cos,f32,nchw>x50"),
    (dict(shape=(2, 8, 20, 300)), "fwd_pairs", "bwd_direct"),                                          # rows too long for a band's threads
    (dict(shape=(256, 64, 56, 56), channels_last=True), "fwd_tile<R1,cos,f32,nhwc>x10", "bwd_tile<R1,cos,f32,nhwc,dense>x10"),   # 256-byte pixels
    (dict(shape=(256, 16, 112, 112), channels_last=True), "fwd_tile<R1,cos,f32,nhwc>x19", "bwd_tile<R1,cos,f32,nhwc>x19"),
    (dict(shape=(2, 8, 100, 140), measure="chisquared2"), "fwd_pairs", "bwd_gather_banded"),          # tables > LDS
    # round 4: Geman-McClure / Canberra / squared chord / chi-squared 1 share one instantiation of the row-band kernels, any map size
    (dict(shape=(2, 8, 100, 140), measure="canberra"), "fwd_tile<R1,canberra,f32,nchw>x50", "bwd_tile<R1,canberra,f32,nchw>x50"),
    (dict(shape=(256, 64, 56, 56), measure="geman", dtype=_abi.BF16, channels_last=True), "fwd_tile<R1,geman,bf16,nhwc>x10", "bwd_tile<R1,geman,bf16,nhwc,dense>x10"),
    (dict(shape=(64, 512, 7, 7), measure="chisquared1"), "fwd_band<R1,chisq1,f32,nchw>x4", "bwd_fast<R1,chisq1,f32,nchw>"),   # up to 512 pixels: the table kernels
    (dict(shape=(256, 192, 14, 14), R=2, measure="squaredchord"), "fwd_band<R2,sqchord,f32,nchw>x1", "bwd_tile<R2,sqchord,f32,nchw>x2"),   # (backward: row bands from 14 x 14 up)
    (dict(shape=(64, 512, 7, 7), measure="chisquared2"), "fwd_pairs", "bwd_gather"),                  # not symmetric: any-geometry kernels
    # round 4: the class default Norm p = 1 (nfp.py:16) and EMD = the same sum (nfp.py:207-216) on the row-band kernels
    (dict(shape=(2, 8, 100, 140), measure="emd"), "fwd_tile<R1,l1,f32,nchw>x50", "bwd_tile<R1,l1,f32,nchw>x50"),
    (dict(shape=(256, 64, 56, 56), measure="norm", p=1.0), "fwd_tile<R1,l1,f32,nchw>x10", "bwd_tile<R1,l1,f32,nchw>x10"),
    (dict(shape=(64, 512, 7, 7), measure="emd"), "fwd_band<R1,l1,f32,nchw>x4", "bwd_fast<R1,l1,f32,nchw>"),
    # round 4: k = 5 float32 NCHW on 14 x 14 and larger: forward on the tables, BACKWARD on the row-band kernel (measured faster)
    (dict(shape=(256, 192, 14, 14), R=2, measure="norm"), "fwd_band<R2,l2,f32,nchw>x1", "bwd_tile<R2,l2,f32,nchw>x2"),
    (dict(shape=(256, 192, 14, 14), R=2, measure="norm", channels_last=True), "fwd_band<R2,l2,f32,nhwc>x1", "bwd_fast<R2,l2,f32,nhwc>"),
    (dict(shape=(64, 512, 7, 7), R=2, measure="cosine"), "fwd_band<R2,cos,f32,nchw>x4", "bwd_fast<R2,cos,f32,nchw>"),
    (dict(shape=(2, 16, 64, 64), mode="circular"), "fwd_pairs", "bwd_direct"),                        # wraps: no bands
])
def test_which_kernel_serves_which_call(lib, d_kw, fwd, bwd):
    d = desc(**d_kw)
    rc, text = plan(lib, d, False)
    assert rc == 0, lib.nfp_last_error()
    assert text.split(" | ")[0] == fwd
    rc, text = plan(lib, d, True)
    assert rc == 0, lib.nfp_last_error()
    assert text.split(" | ")[0] == bwd


def test_small_batches_still_fill_the_chip(lib):
    """At B = 64 the backward and the any-measure forward must put >= 256 workgroups on the 256 CUs."""
    for d, backward in ((desc((64, 512, 7, 7)), True), (desc((64, 512, 7, 7), measure="dot"), False),
                        (desc((64, 512, 7, 7), measure="dot"), True), (desc((16, 512, 7, 7)), True)):
        _, text = plan(lib, d, backward)
        name, grid, block, lds = launches(text)[0]
        assert grid[0] * grid[1] * grid[2] >= 256, text


def test_every_accepted_descriptor_launches_within_device_limits(lib):
    """Randomised sweep over geometry / measure / dtype / layout: whatever the library accepts must describe
    launches with LDS <= 160 KiB, <= 1024 threads and grid y/z <= 65535; what it refuses, it refuses with a message."""
    rnd = random.Random(20260)
    measures = [m for m in _abi.MEASURES if m != "scs"]
    seen, accepted = set(), 0
    for _ in range(3000):
        H, W = rnd.choice([1, 2, 3, 5, 7, 9, 14, 16, 28, 33, 56, 64, 100, 150]), rnd.choice([1, 2, 4, 7, 8, 14, 28, 37, 56, 112, 200])
        R, stride, dil = rnd.choice([1, 1, 1, 2, 2, 3, 5]), rnd.choice([1, 1, 1, 2, 3]), rnd.choice([1, 1, 1, 2, 3])
        d = desc((rnd.choice([1, 3, 16, 64, 256, 1024]), rnd.choice([1, 3, 4, 16, 63, 192, 512, 960, 2048]), H, W), R=R,
                 pad=rnd.choice([0, 1, R, R * dil, R * dil + 2]), stride=stride, dil=dil,
                 mode=rnd.choice(_abi.PAD_MODES), measure=rnd.choice(measures), p=rnd.choice([1.0, 2.0, 3.0]),
                 dtype=rnd.choice([_abi.F32, _abi.BF16]), channels_last=rnd.random() < 0.3)
        for backward in (False, True):
            rc, text = plan(lib, d, backward)
            if rc != 0:
                assert rc in (-1, -2) and lib.nfp_last_error()
                continue
            accepted += 1
            ls = launches(text)
            assert ls, text
            for name, grid, block, lds in ls:
                seen.add(name)
                assert lds <= LDS_MAX and 1 <= block <= 1024, text
                assert all(v >= 1 for v in grid) and grid[1] <= 65535 and grid[2] <= 65535, text
    assert accepted > 2000
    assert {"fwd_band", "fwd_gram", "bwd_fast", "fwd_pairs", "bwd_gather", "bwd_gather_banded"} <= seen, seen


def test_pool_supported_is_a_dry_run_of_both_launchers(lib):
    """nfp_pool_supported used to re-derive the backward's table size and forget the forward's slab —
    [8,256,20,20] R=2 and friends were promised and then refused with rc = -2.  It is now a plan-mode run of
    nfp_pool_forward AND nfp_pool_backward, and maps above 512 pixels are served by the row-band kernels: every
    cosine / L2 / dot / rmse "same" map of a sweep up to 30x30 (R = 1, 2) and the MultiStage maps must be supported, with a
    scratch size that covers the bands' partial sums where the row-band kernels serve."""
    lib.nfp_pool_supported.restype = ctypes.c_int
    cases = [((8, 256, 20, 20), 2), ((1, 192, 18, 18), 2), ((1, 960, 16, 16), 2), ((300, 512, 20, 20), 2), ((300, 32, 22, 23), 2)]
    cases += [((3, 64, s, s + d), R) for s in range(6, 31, 4) for d in (0, 1) for R in (1, 2)]
    cases += [((256, 16, 112, 112), 1), ((256, 24, 56, 56), 1), ((256, 40, 28, 28), 1), ((64, 128, 28, 28), 2)]
    for shape, R in cases:
        for measure in ("cosine", "norm", "dot", "rmse"):
            for cl in (False, True):
                d = desc(shape, R=R, measure=measure, channels_last=cl)
                d.ws = 0x1000 if lib.nfp_workspace_bytes(ctypes.byref(d)) > 0 else None
                assert lib.nfp_pool_supported(ctypes.byref(d)) == 1, (shape, R, measure, cl, lib.nfp_last_error())
                ns = lib.nfp_pool_saved_floats(ctypes.byref(d))
                B, C, H, W = shape
                base = B * H * W if measure == "cosine" else 0
                assert ns >= base
                if H * W > 512:
                    assert ns >= base + B * (C + (2 * R + 1) ** 2 - 1)      # at least one band's partial sums per image
    # and what no fused kernel serves says so
    for kw in (dict(measure="emd"), dict(stride=2), dict(mode="circular"), dict(pad=0)):
        d = desc((4, 64, 14, 14), **kw)
        d.ws = None
        assert lib.nfp_pool_supported(ctypes.byref(d)) == 0, kw
