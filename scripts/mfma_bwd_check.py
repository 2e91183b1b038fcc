#!/usr/bin/env python3
"""Backward phase B on the matrix cores (bf16) vs the vector kernel vs float64; timing."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from neighbour_feature_pooling_amd import NFPPooling, _abi
from neighbour_feature_pooling_amd._host import nfp_host
from bench import time_kernel_graph
L = _abi.load()
dev = "cuda"
torch.manual_seed(0)
for (B, C, H, W, R, meas, mode, cl) in [(4, 192, 14, 14, 2, "norm", "reflect", True), (4, 192, 14, 14, 2, "cosine", "reflect", False),
                                        (3, 512, 7, 7, 1, "cosine", "reflect", False), (3, 512, 7, 7, 1, "norm", "replicate", True),
                                        (2, 64, 5, 9, 1, "cosine", "zeros", False), (2, 96, 6, 5, 2, "norm", "zeros", True),
                                        (70, 64, 7, 7, 1, "cosine", "reflect", False), (2, 32, 1, 40, 1, "cosine", "replicate", True)]:
    ctor = dict(R=R, measure=meas, padding=R, padding_mode=mode)
    if meas == "norm":
        ctor["p"] = 2
    m = NFPPooling(C, **ctor)
    x = torch.randn(B, C, H, W, device=dev).bfloat16()
    if cl:
        x = x.contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True)
    os.environ["NFP_MFMA"] = "0"
    _abi.load().nfp_reload_env()
    out = m(x)
    go = torch.randn_like(out)
    gv, = torch.autograd.grad(out, x, go, retain_graph=True)
    v0 = L.nfp_last_variant().decode()
    os.environ["NFP_MFMA"] = "1"
    _abi.load().nfp_reload_env()
    gm, = torch.autograd.grad(out, x, go, retain_graph=True)
    v1 = L.nfp_last_variant().decode()
    assert "mfma" in v1 and "mfma" not in v0, (v0, v1)
    x64 = x.detach().double().requires_grad_(True)
    gref, = torch.autograd.grad(nfp_host(x64, m.config), x64, go.double())
    sc = gref.abs().max().item()
    print(f"[{B},{C},{H},{W}] R{R} {meas} {mode} {'nhwc' if cl else 'nchw'}: mfma err {(gm.double()-gref).abs().max().item()/sc:.2e}  "
          f"vector err {(gv.double()-gref).abs().max().item()/sc:.2e}  mfma-vs-vector {(gm.float()-gv.float()).abs().max().item()/sc:.2e}", flush=True)
s = torch.cuda.Stream()
for (B, C, S, R, meas, cl) in [(256, 192, 14, 2, "norm", True), (256, 192, 14, 2, "norm", False), (256, 192, 14, 2, "cosine", True),
                               (64, 512, 7, 1, "cosine", False), (64, 512, 7, 1, "cosine", True), (256, 512, 7, 1, "cosine", False)]:
    ctor = dict(R=R, measure=meas, padding=R)
    if meas == "norm":
        ctor["p"] = 2
    m = NFPPooling(C, **ctor)
    x = torch.randn(B, C, S, S, device=dev).bfloat16()
    if cl:
        x = x.contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True)
    go = torch.randn(B, m.out_channels, S, S, device=dev).bfloat16()
    res = {}
    for env in ("1", "0"):
        os.environ["NFP_MFMA"] = env
        _abi.load().nfp_reload_env()
        with torch.cuda.stream(s):
            out = m(x)
            tb = time_kernel_graph(lambda: torch.autograd.grad(out, x, go, retain_graph=True), 20, s)
            res[env] = (tb, L.nfp_last_variant().decode())
    print(f"[{B},{C},{S},{S}] k{2*R+1} {meas} bf16 {'nhwc' if cl else 'nchw'} bwd: " + "  ".join(f"{v[1]} {v[0]:.2f} us" for v in res.values()), flush=True)
