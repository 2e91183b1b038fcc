"""torch.library registration of the NFP op — what keeps the drop-in promise under torch.compile.

The reference's NFP is a chain of ATen ops (nfp.py:132-159): Dynamo traces through it and a model holding it compiles
as one graph.  Here the op is a pair of HIP kernels behind a C ABI; as a `torch.autograd.Function` over ctypes / a pybind
module it is a graph BREAK per call.  Registered as custom ops — `nfp_amd::nfp`, `nfp_amd::nfp_pool` and their backward
ops, each with a fake (shape-only) implementation and an autograd formula — the same kernels sit inside one compiled
graph: `torch.compile(model, fullgraph=True)` works on a model that holds NFPPooling / nfp_pooling.

`functional.nfp` / `functional.nfp_pool` route CUDA tensors here only while Dynamo / export is tracing
(`torch.compiler.is_compiling()`); eager calls keep the C++ autograd nodes (csrc/nfp_torch.cpp), whose host cost is
lower.  The implementations below are the same C-ABI calls the eager nodes make — CUDA only: a CPU tensor runs
`_host.nfp_host`, plain torch ops that Dynamo traces natively (the probes the reference's heads run on CPU dummies,
nfp_heads.py:24-27).  Nothing here touches oracle/.
"""
import ctypes

import torch

from . import _abi

_CFG_SCHEMA = ("int R, str measure, float p, int stride, int padding, int dilation, str padding_mode, bool similarity, "
               "float eps, float q_scs, bool diff_weights, int inner_R")


def _cfg(R, measure, p, stride, padding, dilation, padding_mode, similarity, eps, q_scs, diff_weights, inner_R):
    from .functional import NfpConfig
    return NfpConfig(R=R, measure=measure, p=p, stride=stride, padding=padding, dilation=dilation, padding_mode=padding_mode,
                     similarity=similarity, eps=eps, q_scs=q_scs, diff_weights=diff_weights, inner_R=inner_R)


def cfg_args(cfg):
    """NfpConfig -> the positional arguments of the ops (schema types only: int / float / bool / str)."""
    return (int(cfg.R), str(cfg.measure), float(cfg.p), int(cfg.stride), int(cfg.padding), int(cfg.dilation),
            str(cfg.padding_mode), bool(cfg.similarity), float(cfg.eps), float(cfg.q_scs), bool(cfg.diff_weights),
            int(cfg.inner_R))


def _out_hw(H, W, cfg):
    span = cfg.dilation * (2 * cfg.R) + 1
    return (H + 2 * cfg.padding - span) // cfg.stride + 1, (W + 2 * cfg.padding - span) // cfg.stride + 1


def _saved_floats(shape, dtype, cfg, need_grad):
    """Floats of forward-to-backward state of nfp_forward for this call — a function of the shape and the measure alone
    (include/nfp.h: nfp_saved_floats), so the fake implementation can state it without a GPU."""
    if not need_grad and not (cfg.measure == "attention" and dtype != torch.float32):
        return 0
    d = _abi.NfpDesc()
    d.B, d.C, d.H, d.W = shape
    d.R, d.pad, d.stride, d.dilation = cfg.R, cfg.padding, cfg.stride, cfg.dilation
    d.pad_mode = _abi.PAD_MODES.index(cfg.padding_mode)
    d.measure = _abi.measure_id(cfg.measure)
    d.similarity, d.diff_weights = int(cfg.similarity), int(cfg.diff_weights)
    d.dtype = _abi.F32 if dtype == torch.float32 else _abi.BF16
    d.p, d.eps, d.q_scs = float(cfg.p), float(cfg.eps), float(cfg.q_scs)
    d.sxB, d.sxC, d.sxH, d.sxW = shape[1] * shape[2] * shape[3], shape[2] * shape[3], shape[3], 1
    d.inner_R = int(cfg.inner_R)
    return max(int(_abi.load().nfp_saved_floats(ctypes.byref(d))), 0)


def _pool_saved_bound(shape, dtype, cfg):
    """An upper bound, from the shape alone, of nfp_pool_saved_floats (per-pixel state + every row band's share of the two
    pooled sums: at most H bands of C + N floats per image): what the compiled graph allocates."""
    B, C, H, W = shape
    return _saved_floats(shape, dtype, cfg, True) + B * H * (C + cfg.out_channels)


# ---- nfp: x -> (maps, saved) ------------------------------------------------------------------------------------------
@torch.library.custom_op("nfp_amd::nfp", mutates_args=(), device_types="cuda", schema=f"(Tensor x, {_CFG_SCHEMA}, bool need_grad) -> (Tensor, Tensor)")
def nfp_op(x, R, measure, p, stride, padding, dilation, padding_mode, similarity, eps, q_scs, diff_weights, inner_R, need_grad):
    cfg = _cfg(R, measure, p, stride, padding, dilation, padding_mode, similarity, eps, q_scs, diff_weights, inner_R)
    from . import functional as F
    L = _abi.load()
    xd, layout = F._dense(x)
    d, oshape, _, no_bwd = F._plan(xd, layout, cfg)
    if need_grad and no_bwd is not None:
        raise _abi.NfpUnsupported(f"libnfp_hip: the forward of this call is served but its backward is not ({no_bwd})")
    ns = _saved_floats(tuple(x.shape), x.dtype, cfg, need_grad)
    with F._on_device(x.device):
        out = torch.empty(oshape, dtype=x.dtype, device=x.device)
        saved = torch.empty(max(ns, 0), dtype=torch.float32, device=x.device)
        _abi.check(L.nfp_forward(ctypes.byref(d), xd.data_ptr(), out.data_ptr(), saved.data_ptr() if ns > 0 else None,
                                 F._raw_stream(x.device)))
    return out, saved


@nfp_op.register_fake
def _(x, R, measure, p, stride, padding, dilation, padding_mode, similarity, eps, q_scs, diff_weights, inner_R, need_grad):
    cfg = _cfg(R, measure, p, stride, padding, dilation, padding_mode, similarity, eps, q_scs, diff_weights, inner_R)
    B, _, H, W = x.shape
    Ho, Wo = _out_hw(H, W, cfg)
    ns = _saved_floats(tuple(x.shape), x.dtype, cfg, need_grad)
    return x.new_empty((B, cfg.out_channels, Ho, Wo)), x.new_empty((ns,), dtype=torch.float32)


@torch.library.custom_op("nfp_amd::nfp_backward", mutates_args=(), device_types="cuda",
                         schema=f"(Tensor x, Tensor out, Tensor saved, Tensor grad_out, {_CFG_SCHEMA}) -> Tensor")
def nfp_backward_op(x, out, saved, grad_out, R, measure, p, stride, padding, dilation, padding_mode, similarity, eps, q_scs,
                    diff_weights, inner_R):
    cfg = _cfg(R, measure, p, stride, padding, dilation, padding_mode, similarity, eps, q_scs, diff_weights, inner_R)
    from . import functional as F
    L = _abi.load()
    xd, layout = F._dense(x)
    d = F._plan(xd, layout, cfg)[0]
    go = grad_out.contiguous()
    if go.dtype != x.dtype:
        go = go.to(x.dtype)
    with F._on_device(x.device):
        gx = torch.empty(xd.shape, dtype=x.dtype, device=x.device,
                         memory_format=torch.channels_last if layout == "nhwc" else torch.contiguous_format)
        _abi.check(L.nfp_backward(ctypes.byref(d), xd.data_ptr(), go.data_ptr(), out.data_ptr(),
                                  saved.data_ptr() if saved.numel() else None, gx.data_ptr(), F._raw_stream(x.device)))
    return gx


@nfp_backward_op.register_fake
def _(x, out, saved, grad_out, *cfg_fields):
    return torch.empty_like(x)


def _nfp_setup(ctx, inputs, output):
    x = inputs[0]
    ctx.cfg_fields = inputs[1:13]
    ctx.save_for_backward(x, output[0], output[1])


def _nfp_bwd(ctx, g_out, g_saved):
    x, out, saved = ctx.saved_tensors
    gx = torch.ops.nfp_amd.nfp_backward(x, out, saved, g_out, *ctx.cfg_fields)
    return (gx,) + (None,) * 13


nfp_op.register_autograd(_nfp_bwd, setup_context=_nfp_setup)


# ---- nfp_pool: x -> (gap, nfpm, maps, saved) — the fused tail of models/NFP_Pooling.py:27-31 ------------------------------
@torch.library.custom_op("nfp_amd::nfp_pool", mutates_args=(), device_types="cuda",
                         schema=f"(Tensor x, {_CFG_SCHEMA}, bool want_gap, bool need_grad) -> (Tensor, Tensor, Tensor, Tensor)")
def nfp_pool_op(x, R, measure, p, stride, padding, dilation, padding_mode, similarity, eps, q_scs, diff_weights, inner_R,
                want_gap, need_grad):
    cfg = _cfg(R, measure, p, stride, padding, dilation, padding_mode, similarity, eps, q_scs, diff_weights, inner_R)
    from . import functional as F
    f32 = dict(dtype=torch.float32, device=x.device)
    if not F.nfp_pool_fused_ok(x, cfg):
        # (calls the fused kernels do not serve: the two means of the maps' own op.  The state buffer has the size the fake
        # implementation states — a function of the shape alone — with the maps' state in front)
        maps, sv = nfp_op(x, *cfg_args(cfg), need_grad)
        saved = torch.empty(_pool_saved_bound(tuple(x.shape), x.dtype, cfg), **f32)
        saved[:sv.numel()] = sv
        gap = x.float().mean((2, 3)) if want_gap else torch.empty(0, x.shape[1], **f32)
        return gap, maps.float().mean((2, 3)), maps if need_grad else maps.new_empty(0), saved
    L = _abi.load()
    xd, layout = F._dense(x)
    d, (B, N, Ho, Wo), _, _ = F._plan(xd, layout, cfg)
    ns = _pool_saved_bound(tuple(x.shape), x.dtype, cfg)
    assert F._pool_saved_floats(xd, layout, cfg, d) <= ns
    with F._on_device(x.device):
        gap = torch.empty(B if want_gap else 0, x.shape[1], **f32)
        nfpm = torch.empty(B, N, **f32)
        out_map = torch.empty((B, N, Ho, Wo) if need_grad else (0,), dtype=x.dtype, device=x.device)
        saved = torch.empty(max(ns, 0), **f32)
        _abi.check(L.nfp_pool_forward(ctypes.byref(d), xd.data_ptr(), gap.data_ptr() if want_gap else None, nfpm.data_ptr(),
                                      out_map.data_ptr() if need_grad else None, saved.data_ptr() if ns > 0 else None,
                                      F._raw_stream(x.device)))
    return gap, nfpm, out_map, saved


@nfp_pool_op.register_fake
def _(x, R, measure, p, stride, padding, dilation, padding_mode, similarity, eps, q_scs, diff_weights, inner_R, want_gap,
      need_grad):
    cfg = _cfg(R, measure, p, stride, padding, dilation, padding_mode, similarity, eps, q_scs, diff_weights, inner_R)
    B, C, H, W = x.shape
    Ho, Wo = _out_hw(H, W, cfg)
    N = cfg.out_channels
    return (x.new_empty((B if want_gap else 0, C), dtype=torch.float32), x.new_empty((B, N), dtype=torch.float32),
            x.new_empty((B, N, Ho, Wo)) if need_grad else x.new_empty((0,)),
            x.new_empty((_pool_saved_bound(tuple(x.shape), x.dtype, cfg),), dtype=torch.float32))


@torch.library.custom_op("nfp_amd::nfp_pool_backward", mutates_args=(), device_types="cuda",
                         schema=f"(Tensor x, Tensor out_map, Tensor saved, Tensor? grad_gap, Tensor grad_nfpm, {_CFG_SCHEMA}) -> Tensor")
def nfp_pool_backward_op(x, out_map, saved, grad_gap, grad_nfpm, R, measure, p, stride, padding, dilation, padding_mode,
                         similarity, eps, q_scs, diff_weights, inner_R):
    cfg = _cfg(R, measure, p, stride, padding, dilation, padding_mode, similarity, eps, q_scs, diff_weights, inner_R)
    from . import functional as F
    P = out_map.shape[2] * out_map.shape[3]
    if not F.nfp_pool_fused_ok(x, cfg):
        go = (grad_nfpm.to(out_map.dtype) / P)[:, :, None, None].expand_as(out_map).contiguous()
        gx = nfp_backward_op(x, out_map, saved, go, *cfg_args(cfg)).float()
        if grad_gap is not None:
            gx = gx + (grad_gap.float() / (x.shape[2] * x.shape[3]))[:, :, None, None]
        return gx.to(x.dtype)
    L = _abi.load()
    xd, layout = F._dense(x)
    d = F._plan(xd, layout, cfg)[0]
    gg = grad_gap.contiguous().float() if grad_gap is not None else None
    gn = grad_nfpm.contiguous().float()
    with F._on_device(x.device):
        gx = torch.empty(xd.shape, dtype=x.dtype, device=x.device,
                         memory_format=torch.channels_last if layout == "nhwc" else torch.contiguous_format)
        _abi.check(L.nfp_pool_backward(ctypes.byref(d), xd.data_ptr(), gg.data_ptr() if gg is not None else None, gn.data_ptr(),
                                       out_map.data_ptr(), saved.data_ptr() if saved.numel() else None, gx.data_ptr(),
                                       F._raw_stream(x.device)))
    return gx


@nfp_pool_backward_op.register_fake
def _(x, out_map, saved, grad_gap, grad_nfpm, *cfg_fields):
    return torch.empty_like(x)


def _pool_setup(ctx, inputs, output):
    ctx.cfg_fields = inputs[1:13]
    ctx.want_gap = inputs[13]
    ctx.save_for_backward(inputs[0], output[2], output[3])


def _pool_bwd(ctx, g_gap, g_nfpm, g_map, g_saved):
    x, out_map, saved = ctx.saved_tensors
    gx = torch.ops.nfp_amd.nfp_pool_backward(x, out_map, saved, g_gap if ctx.want_gap else None, g_nfpm, *ctx.cfg_fields)
    return (gx,) + (None,) * 14


nfp_pool_op.register_autograd(_pool_bwd, setup_context=_pool_setup)
