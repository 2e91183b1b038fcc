"""Functional NFP: `nfp(x, cfg)` — the autograd op behind NFPPooling.forward.

CUDA tensors go through libnfp_hip.so (include/nfp.h) — one fused forward kernel,
one fused backward kernel, saved state = the output map plus one float per input
pixel.  Nothing else can serve a CUDA tensor: if the library is missing or refuses
the configuration the call raises.  CPU tensors use `_host.nfp_host` (see there).
"""
import ctypes
from collections import OrderedDict
from dataclasses import dataclass

import torch

from . import _abi
from ._host import nfp_host


@dataclass(frozen=True)
class NfpConfig:
    """Constructor arguments of NFPPooling that define the op (nfp.py:16-39)."""
    R: int = 1
    measure: str = "norm"          # lower-cased name (nfp.py:21)
    p: float = 1
    stride: int = 1
    padding: int = 0
    dilation: int = 1
    padding_mode: str = "reflect"
    similarity: bool = True
    eps: float = 1e-6
    q_scs: float = 1e-6
    diff_weights: bool = True      # raw measure string in ['norm','rmse','mahalanobis'] (nfp.py:74)
    inner_R: int = 0               # 1 with R = 2: also the maps of radius 1 (padding 1), first in the channel axis —
                                   # the concatenation models/nfp_heads.py:80-118 builds from two layers

    @property
    def kernel_size(self):
        return 2 * self.R + 1

    @property
    def out_channels(self):
        return self.kernel_size ** 2 - 1 + ((2 * self.inner_R + 1) ** 2 - 1 if self.inner_R else 0)


_DTYPES = {torch.float32: _abi.F32, torch.bfloat16: _abi.BF16}

_CPP = None     # the C++ autograd nodes (csrc/nfp_torch.cpp), None until looked for, False when absent


def _cpp_nodes():
    """neighbour_feature_pooling_amd/_nfp_torch.so — the same two autograd nodes as the Python classes below, in C++
    (no interpreter on the launch path).  Optional: NFP_PY_NODES=1 or a missing module selects the Python nodes."""
    global _CPP
    if _CPP is None:
        import importlib.util
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_nfp_torch.so")
        _CPP = False
        if os.environ.get("NFP_PY_NODES") != "1" and os.path.exists(path):
            _abi.load()     # libnfp_hip.so first: the module links against it
            spec = importlib.util.spec_from_file_location(__package__ + "._nfp_torch", path)
            mod = importlib.util.module_from_spec(spec)
            try:
                spec.loader.exec_module(mod)
                if mod.desc_bytes == ctypes.sizeof(_abi.NfpDesc):
                    _CPP = mod
            except (ImportError, OSError):   # built against another torch: the Python nodes serve
                _CPP = False
    return _CPP


def _cpp_call(fn, *args):
    """Run a C++ node; its TORCH_CHECK messages carry the library's error class as a prefix."""
    try:
        return fn(*args)
    except RuntimeError as e:
        msg = str(e).split("\n")[0]
        if "libnfp_hip unsupported: " in msg:
            raise _abi.NfpUnsupported("libnfp_hip: " + msg.split("libnfp_hip unsupported: ", 1)[1]) from None
        if "libnfp_hip error: " in msg:
            raise _abi.NfpError("libnfp_hip: " + msg.split("libnfp_hip error: ", 1)[1]) from None
        raise


def _inner_layout(x):
    """'nchw' / 'nhwc' when every image of x is dense in that order — whatever the batch stride: the kernels take the
    batch stride from the descriptor, so e.g. ViT patch tokens behind a class token (a [B,1+HW,C] buffer viewed as
    [B,C,H,W], texture_pooling.py:181-188) are read in place — else None."""
    _, C, H, W = x.shape
    st = x.stride()

    def matches(canon):
        return all(n == 1 or s == c for n, s, c in zip((C, H, W), st[1:], canon))

    if matches((H * W, W, 1)):
        return "nchw"
    if matches((1, W * C, C)):
        return "nhwc"
    return None


def _canonical_strides(x, layout):
    """Element strides handed to the library (size-1 dimensions carry arbitrary strides in torch)."""
    B, C, H, W = x.shape
    sB = x.stride(0) if B > 1 else C * H * W
    return (sB, H * W, W, 1) if layout == "nchw" else (sB, 1, W * C, C)


def _dense(x):
    """(x, layout): x itself when its images are dense NCHW or channels-last (read in place by strides), otherwise an
    NCHW copy."""
    layout = _inner_layout(x)
    if layout is None or (x.shape[0] > 1 and x.stride(0) < x.shape[1] * x.shape[2] * x.shape[3]):
        return x.contiguous(), "nchw"
    return x, layout


def make_desc(x, cfg, layout=None):
    if x.dtype not in _DTYPES:
        raise _abi.NfpUnsupported(f"NFP HIP kernels take float32 or bfloat16 feature maps, got {x.dtype}")
    if layout is None:
        x, layout = _dense(x)
    d = _abi.NfpDesc()
    d.B, d.C, d.H, d.W = x.shape
    d.R, d.pad, d.stride, d.dilation = cfg.R, cfg.padding, cfg.stride, cfg.dilation
    d.pad_mode = _abi.PAD_MODES.index(cfg.padding_mode)
    d.measure = _abi.measure_id(cfg.measure)
    d.similarity = int(bool(cfg.similarity))
    d.diff_weights = int(bool(cfg.diff_weights))
    d.dtype = _DTYPES[x.dtype]
    d.p, d.eps, d.q_scs = float(cfg.p), float(cfg.eps), float(cfg.q_scs)
    d.sxB, d.sxC, d.sxH, d.sxW = _canonical_strides(x, layout)
    d.sgB = d.C * d.H * d.W          # grad_x is always allocated dense, in x's inner layout
    d.inner_R = int(cfg.inner_R)
    return d


def output_shape(d):
    L = _abi.load()
    n, ho, wo = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    _abi.check(L.nfp_output_shape(ctypes.byref(d), ctypes.byref(n), ctypes.byref(ho), ctypes.byref(wo)))
    return d.B, n.value, ho.value, wo.value


_WORKSPACES = {}            # (device, geometry) -> [uint8 tensor, fill event or None, fill stream, device index] or False
_PENDING_FILLS = []         # the records of _WORKSPACES whose fill event has not been seen complete yet


def _order_after_fills(device):
    """Every launch path calls this while a table fill may still be running (the ordering used to live in
    _workspace(), which a plan-cache hit never reaches — a same-shape call on ANOTHER stream right after the first one
    launched on tables that were not written yet).  A fill is ordered before later work of its own stream; any other
    stream waits for its event on the device (no host synchronisation); a completed event is dropped, so in steady state
    this is one truthiness test of an empty list.  Under a graph capture neither an event query nor a wait on an event
    recorded outside the capture is legal: nothing is done — a capture is preceded by a device synchronisation
    (torch.cuda.graph does one itself), which also completes every fill enqueued before it."""
    if torch.cuda.is_current_stream_capturing():
        return
    cur = None
    for rec in list(_PENDING_FILLS):
        if rec[1] is None or rec[1].query():
            rec[1] = None
            _PENDING_FILLS.remove(rec)
        elif rec[3] == device.index:
            if cur is None:
                cur = _raw_stream(device)
            if cur != rec[2]:
                torch.cuda.current_stream(device).wait_event(rec[1])


def _workspace(d, device):
    """Device pointer of the workspace of `d` (include/nfp.h: nfp_workspace_bytes / nfp_workspace_init: the pooled
    kernels' arrival counters and the geometry's constant tables), or None.  It depends on the geometry only, so every call
    with the same map size and kernel on the same stream shares one buffer.  The fill kernel is enqueued once, on the stream that first needs the tables, followed by an event
    (_order_after_fills).  Under a graph capture no table is built (the fill would only be recorded, not run, and a
    later eager call would read an unfilled buffer): that one call is planned without tables and nothing is cached —
    warm a geometry up before capturing it, as every graph user does anyway."""
    # (one workspace per STREAM: besides the constant tables it holds the arrival counters of the pooled kernels' row bands
    # — include/nfp.h: one pooled launch at a time per workspace; launches on one stream are ordered)
    key = (device.index, d.H, d.W, d.R, d.pad, d.stride, d.dilation, d.pad_mode, d.inner_R, _raw_stream(device))
    rec = _WORKSPACES.get(key)
    if rec is None:
        L = _abi.load()
        nbytes = int(L.nfp_workspace_bytes(ctypes.byref(d)))
        if nbytes < 0:
            return None     # the descriptor itself is refused (bad p / eps ...): say nothing about the GEOMETRY
        if nbytes == 0:
            _WORKSPACES[key] = rec = False
        else:
            if torch.cuda.is_current_stream_capturing():
                import warnings
                warnings.warn("NFP: a geometry first seen inside a graph capture runs without its workspace tables "
                              "(general kernels); call the op once before capturing", RuntimeWarning, stacklevel=4)
                return None
            with _on_device(device):
                ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
                _abi.check(L.nfp_workspace_init(ctypes.byref(d), ws.data_ptr(), _raw_stream(device)))
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(device))
            _WORKSPACES[key] = rec = [ws, ev, _raw_stream(device), device.index]
            _PENDING_FILLS.append(rec)
    if rec is False:
        return None
    return rec[0].data_ptr()


_DESC_TENSORS = {}          # id(descriptor) -> (descriptor, uint8 tensor over its bytes)
_PLANS = OrderedDict()      # least recently used first
_PLANS_MAX = 256


def _plans_get(key):
    plan = _PLANS.get(key)
    if plan is not None:
        _PLANS.move_to_end(key)
    return plan


def _plans_put(key, plan):
    _PLANS[key] = plan
    if len(_PLANS) > _PLANS_MAX:
        _PLANS.popitem(last=False)


def _plan(x, layout, cfg):
    """(descriptor, output shape, saved floats, why-no-backward) for this input signature — cached (LRU): in eager
    mode the host side of a call (a few ctypes round trips) otherwise costs more than the two kernels.  The last
    entry is None when nfp_backward serves the descriptor too, else the library's message: forward and backward
    envelopes differ for a few large maps, and a call that will need a gradient must fail in forward(), not inside
    loss.backward()."""
    key = (tuple(x.shape), x.stride(0), layout, x.dtype, cfg, x.device.index)
    plan = _plans_get(key)
    if _PENDING_FILLS:
        _order_after_fills(x.device)
    if plan is None:
        L = _abi.load()
        d = make_desc(x, cfg, layout)
        d.ws = _workspace(d, x.device)
        cacheable = d.ws is not None or not torch.cuda.is_current_stream_capturing()
        d.cacheable = cacheable     # (a Python attribute of the ctypes object: the pooled-tail caches below ask)
        buf = ctypes.create_string_buffer(1024)
        rc = L.nfp_plan(ctypes.byref(d), 1, buf, len(buf))
        no_bwd = None if rc == 0 else L.nfp_last_error().decode()
        plan = (d, output_shape(d), int(L.nfp_saved_floats(ctypes.byref(d))), no_bwd)
        if cacheable:
            _plans_put(key, plan)
        _DESC_TENSORS[id(d)] = (d, torch.frombuffer(d, dtype=torch.uint8))   # (same memory; for the C++ nodes)
        if len(_DESC_TENSORS) > 4 * _PLANS_MAX:
            live = {id(p[0]) for p in _PLANS.values() if isinstance(p, tuple)}
            for k in [k for k in _DESC_TENSORS if k not in live]:
                del _DESC_TENSORS[k]
    return plan


def _raw_stream(dev):
    """hipStream_t of torch's current stream on `dev` as an int.  torch.cuda.current_stream() builds a Stream
    object through several Python layers (~20 us, more than both kernels' launch cost); the raw accessor that
    Inductor-generated code uses is a single C call."""
    try:
        return torch._C._cuda_getCurrentRawStream(dev.index)
    except AttributeError:  # pragma: no cover - older / newer torch without the private accessor
        return torch.cuda.current_stream(dev).cuda_stream


class _on_device:
    """`with torch.cuda.device(dev)` only when dev is not already current (the common case costs nothing)."""

    def __init__(self, dev):
        self.ctx = None if dev.index == torch.cuda.current_device() else torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)


class _NfpHip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, cfg, need_grad):
        # need_grad: x.requires_grad and grad mode on, sampled by the caller (grad mode is always off in here, and
        # ctx.needs_input_grad ignores torch.no_grad())
        L = _abi.load()
        x, layout = _dense(x)
        d, oshape, ns, no_bwd = _plan(x, layout, cfg)
        if need_grad:
            if no_bwd is not None:
                raise _abi.NfpUnsupported(f"libnfp_hip: the forward of this call is served but its backward is not "
                                          f"({no_bwd}); run it under torch.no_grad() or on a detached input")
        elif not (cfg.measure == "attention" and x.dtype != torch.float32):
            ns = 0      # (bf16 Attention keeps its raw dots in this scratch even without a backward)
        with _on_device(x.device):
            out = torch.empty(oshape, dtype=x.dtype, device=x.device)
            saved = torch.empty(max(ns, 0), dtype=torch.float32, device=x.device)
            stream = _raw_stream(x.device)
            _abi.check(L.nfp_forward(ctypes.byref(d), x.data_ptr(), out.data_ptr(),
                                     saved.data_ptr() if ns > 0 else None, stream))
        ctx.desc = d
        ctx.layout = layout
        ctx.save_for_backward(x, out, saved)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        x, out, saved = ctx.saved_tensors
        L = _abi.load()
        d = ctx.desc
        go = grad_out.contiguous()
        if go.dtype != x.dtype:
            go = go.to(x.dtype)
        with _on_device(x.device):
            # dense, in x's inner layout (desc.sgB): a batch-strided view of x does not get a gradient with gaps
            gx = torch.empty(x.shape, dtype=x.dtype, device=x.device,
                             memory_format=torch.channels_last if ctx.layout == "nhwc" else torch.contiguous_format)
            stream = _raw_stream(x.device)
            _abi.check(L.nfp_backward(ctypes.byref(d), x.data_ptr(), go.data_ptr(), out.data_ptr(),
                                      saved.data_ptr() if saved.numel() else None, gx.data_ptr(), stream))
        return gx, None, None


class _NfpPoolHip(torch.autograd.Function):
    """(gap [B,C], nfpm [B,N]) = fused tail of models/NFP_Pooling.py:27-31 in one pass over x.  want_gap = False: the
    pooled NFP maps alone (texture_pooling.py:251-252, 320-321) — gap comes back empty and its sums are never taken;
    need_grad = False: the maps themselves are not stored either (include/nfp.h, ABI 6)."""

    @staticmethod
    def forward(ctx, x, cfg, want_gap, need_grad):
        L = _abi.load()
        x, layout = _dense(x)
        d, (B, N, Ho, Wo), _, _ = _plan(x, layout, cfg)
        ns = _pool_saved_floats(x, layout, cfg, d)
        with _on_device(x.device):
            gap = torch.empty(B if want_gap else 0, x.shape[1], dtype=torch.float32, device=x.device)
            nfpm = torch.empty(B, N, dtype=torch.float32, device=x.device)
            out_map = torch.empty((B, N, Ho, Wo) if need_grad else (0,), dtype=x.dtype, device=x.device)
            saved = torch.empty(max(ns, 0), dtype=torch.float32, device=x.device)
            stream = _raw_stream(x.device)
            _abi.check(L.nfp_pool_forward(ctypes.byref(d), x.data_ptr(), gap.data_ptr() if want_gap else None,
                                          nfpm.data_ptr(), out_map.data_ptr() if need_grad else None,
                                          saved.data_ptr() if ns > 0 else None, stream))
        ctx.desc = d
        ctx.layout = layout
        ctx.want_gap = want_gap
        ctx.save_for_backward(x, out_map, saved)
        return gap, nfpm

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_gap, g_nfpm):
        x, out_map, saved = ctx.saved_tensors
        L = _abi.load()
        g_gap = g_gap.contiguous().float() if ctx.want_gap else None
        g_nfpm = g_nfpm.contiguous().float()
        with _on_device(x.device):
            gx = torch.empty(x.shape, dtype=x.dtype, device=x.device,
                             memory_format=torch.channels_last if ctx.layout == "nhwc" else torch.contiguous_format)
            stream = _raw_stream(x.device)
            _abi.check(L.nfp_pool_backward(ctypes.byref(ctx.desc), x.data_ptr(),
                                           g_gap.data_ptr() if g_gap is not None else None, g_nfpm.data_ptr(),
                                           out_map.data_ptr(), saved.data_ptr() if saved.numel() else None,
                                           gx.data_ptr(), stream))
        return gx, None, None, None


def _pool_saved_floats(x, layout, cfg, d):
    """nfp_pool_saved_floats of the plan's descriptor (cached with the plans)."""
    key = ("pool_ns", tuple(x.shape), x.stride(0), layout, x.dtype, cfg, x.device.index)
    ns = _plans_get(key)
    if ns is None:
        ns = int(_abi.load().nfp_pool_saved_floats(ctypes.byref(d)))
        if getattr(d, "cacheable", True):   # (a table-less plan made under a graph capture says nothing about later calls)
            _plans_put(key, ns)
    return ns


def nfp_pool_fused_ok(x, cfg):
    """True when the fused GAP + pooled-NFP kernels can serve this call: "same" maps (stride 1, pad = R; above 512 pixels:
    rows of W <= 254 / 142 pixels for k = 3 / 5, any height), cosine / dot / gfc / L2 / rmse, float32 or bfloat16, images dense in NCHW
    or channels-last order.  (The library answers: a dry run of both launchers.)"""
    if not (x.is_cuda and x.dim() == 4 and x.dtype in _DTYPES):
        return False
    layout = _inner_layout(x)
    if layout is None or (x.shape[0] > 1 and x.stride(0) < x.shape[1] * x.shape[2] * x.shape[3]):
        return False
    if layout == "nhwc" and (x.data_ptr() % 16 or (x.stride(0) * x.element_size()) % 16):
        return False          # (the channels-last kernels load 16 bytes per lane)
    key = ("pool", tuple(x.shape), x.stride(0), layout, x.dtype, cfg, x.device.index)
    ok = _plans_get(key)
    if ok is None:
        d = _plan(x, layout, cfg)[0]
        ok = bool(_abi.load().nfp_pool_supported(ctypes.byref(d)))
        if getattr(d, "cacheable", True):
            _plans_put(key, ok)
    return ok


_WARNED_F64 = False


def _amp_input(x):
    """What the reference's op sequence does with low precision on the GPU (measured on an MI355X,
    profiles/r03_c_autocast_probe_reference_ops.jsonl; scripts/probe_autocast.py):
      * under torch.autocast("cuda", ...) F.cosine_similarity and linalg.norm are on autocast's float32 list: the maps
        come back float32 whatever the input type, gradients in the input's type;
      * a float16 tensor outside autocast runs in float16 end to end: float16 maps.
    The kernels store float32 / bfloat16 and always sum in float32, so: under autocast the input is taken as float32
    and the op runs outside autocast (float32 maps, as the reference; no bf16 rounding of the inputs in between, so
    it is the closer of the two to the float32 result); float16 is computed in float32 and the maps rounded to float16
    once.  Returns (tensor to run on, dtype to cast the result to or None)."""
    if x.is_cuda and torch.is_autocast_enabled("cuda"):
        return x.float(), None
    if x.is_cuda and x.dtype == torch.float16:
        return x.float(), torch.float16
    if x.is_cuda and x.dtype == torch.float64:
        # The reference follows the input type (nfp.py:141-159): float64 in, float64 maps.  The kernels compute in
        # float32: the maps agree with the float64 reference to float32 rounding (<= 1e-6 of the tensor's magnitude, inside
        # the 1e-5 parity bar), in the reference's type.  Said once per process, not silently.
        global _WARNED_F64
        if not _WARNED_F64:
            import warnings
            warnings.warn("NFP: float64 feature maps are computed in float32 by the HIP kernels and returned as float64",
                          RuntimeWarning, stacklevel=4)
            _WARNED_F64 = True
        return x.float(), torch.float64
    return x, None


def nfp_pool(x, cfg, want_gap=True):
    """(GAP(x) [B,C], GAP(NFP(x)) [B,N]) — NFP_Pooling.py:27-31.  Fused on the GPU where supported,
    otherwise the same two reductions composed from `nfp` and torch ops.  want_gap = False: (None, GAP(NFP(x))) — the
    channel sums of x are not taken (nfp_pooled)."""
    xin, cast = _amp_input(x)
    if xin is not x:
        with torch.autocast("cuda", enabled=False):
            gap, nfpm = nfp_pool(xin, cfg, want_gap)
        return (gap, nfpm) if cast is None else (None if gap is None else gap.to(cast), nfpm.to(cast))
    if x.is_cuda and torch.compiler.is_compiling() and x.dim() == 4:
        # under torch.compile: the registered custom op (one node in the compiled graph, no graph break) — _ops.py
        from . import _ops
        need_grad = x.requires_grad and torch.is_grad_enabled()
        gap, nfpm, _, _ = torch.ops.nfp_amd.nfp_pool(x, *_ops.cfg_args(cfg), bool(want_gap), bool(need_grad))
        return (gap if want_gap else None), nfpm
    if x.dim() == 4 and nfp_pool_fused_ok(x, cfg):
        need_grad = x.requires_grad and torch.is_grad_enabled()
        try:
            cpp = _cpp_nodes()
            if cpp:
                xd, layout = _dense(x)
                d, oshape, _, _ = _plan(xd, layout, cfg)
                ns = _pool_saved_floats(xd, layout, cfg, d)
                gap, nfpm = _cpp_call(cpp.nfp_pool_apply, xd, _DESC_TENSORS[id(d)][1], list(oshape), max(ns, 0),
                                      layout == "nhwc", bool(want_gap), bool(need_grad))
            else:
                gap, nfpm = _NfpPoolHip.apply(x, cfg, bool(want_gap), bool(need_grad))
            return (gap if want_gap else None), nfpm
        except _abi.NfpUnsupported:
            pass    # (nfp_pool_supported is a dry run of both launchers; should it ever disagree, the composition serves)
    return (x.mean((2, 3)) if want_gap else None), nfp(x, cfg).mean((2, 3))


def nfp_pooled(x, cfg):
    """adaptive_avg_pool2d(NFPPooling(x), 1).flatten(1) — [B, N] float32 — as MobileNetV3_MultiStageNFP and MidNFP consume
    their NFP layers (models/texture_pooling.py:251-252, 320-321): no GAP(x) beside it.  On the GPU one pass over x that
    neither sums the channels of x nor, without a gradient to follow, stores the maps."""
    return nfp_pool(x, cfg, want_gap=False)[1]


def nfp_multi_radius(x, cfg1, cfg2):
    """torch.cat([NFP_R1(x), NFP_R2(x)], dim=1) for two configurations that differ only in R / padding (radii 1 and 2,
    padding = R): what MultiRadiusNFPHead.forward computes with two layers (models/nfp_heads.py:109-110).  On the GPU
    both radii come from ONE pass over x (and one backward pass) where the hot-path kernels serve the map; otherwise
    the two maps are computed one after the other."""
    import dataclasses
    fusable = (x.is_cuda and x.dim() == 4 and cfg1.R == 1 and cfg2.R == 2 and cfg1.padding == 1 and cfg2.padding == 2
               and dataclasses.replace(cfg1, R=2, padding=2) == cfg2 and cfg2.inner_R == 0
               and cfg2.measure in ("cosine", "norm", "dot", "gfc", "rmse", "emd"))   # (norm: p = 1 or 2; else refused below)
    if fusable:
        try:
            return nfp(x, dataclasses.replace(cfg2, inner_R=1))
        except _abi.NfpUnsupported:
            pass
    return torch.cat([nfp(x, cfg1), nfp(x, cfg2)], dim=1)


def nfp(x, cfg):
    """[B,C,H,W] -> [B, k*k-1, H', W'] neighbour-similarity maps (NFPPooling.forward, nfp.py:132-134)."""
    if x.dim() != 4:
        raise RuntimeError(f"NFP expects a 4-D [B,C,H,W] feature map, got {tuple(x.shape)}")
    xin, cast = _amp_input(x)
    if xin is not x:
        with torch.autocast("cuda", enabled=False):
            out = nfp(xin, cfg)
        return out if cast is None else out.to(cast)
    if x.is_cuda and torch.compiler.is_compiling() and cfg.measure != "scs":
        # under torch.compile: the registered custom op (one node in the compiled graph, no graph break) — _ops.py
        from . import _ops
        need_grad = x.requires_grad and torch.is_grad_enabled()
        return torch.ops.nfp_amd.nfp(x, *_ops.cfg_args(cfg), bool(need_grad))[0]
    if x.is_cuda and cfg.measure == "scs":
        # The one measure without a kernel, on purpose: the reference's SharpenedCosine divides
        # (B,N,H,W) by (B,1,N,H,W) and so averages over the BATCH (nfp.py:359-374).  Its exact behaviour
        # is reproduced with torch ops on the tensor's own device, and said out loud; every other
        # measure on a CUDA tensor is served by libnfp_hip.so or raises.
        import warnings
        warnings.warn("NFP measure 'scs' (SharpenedCosine) mixes batch elements in the reference; it runs as "
                      "PyTorch ops on the GPU, not through the HIP kernels", RuntimeWarning, stacklevel=3)
        return nfp_host(x, cfg)
    if x.is_cuda:
        need_grad = x.requires_grad and torch.is_grad_enabled()
        cpp = _cpp_nodes()
        if cpp:
            xd, layout = _dense(x)
            d, oshape, ns, no_bwd = _plan(xd, layout, cfg)
            if need_grad and no_bwd is not None:
                raise _abi.NfpUnsupported(f"libnfp_hip: the forward of this call is served but its backward is not "
                                          f"({no_bwd}); run it under torch.no_grad() or on a detached input")
            if not need_grad and not (cfg.measure == "attention" and x.dtype != torch.float32):
                ns = 0
            return _cpp_call(cpp.nfp_apply, xd, _DESC_TENSORS[id(d)][1], list(oshape), max(ns, 0), layout == "nhwc")
        return _NfpHip.apply(x, cfg, need_grad)
    if cfg.inner_R:
        import dataclasses
        return torch.cat([nfp_host(x, dataclasses.replace(cfg, R=cfg.inner_R, padding=cfg.inner_R, inner_R=0)),
                          nfp_host(x, dataclasses.replace(cfg, inner_R=0))], dim=1)
    return nfp_host(x, cfg)
