#!/usr/bin/env python3
"""Hot-path kernels vs the any-measure kernels on the shapes both serve (same process, same device)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from neighbour_feature_pooling_amd import NFPPooling, _abi
from bench import time_kernel_graph
L = _abi.load()
s = torch.cuda.Stream()
for (B, C, S, R, meas, dt) in [(64, 512, 7, 1, "cosine", torch.float32), (64, 512, 7, 1, "norm", torch.float32),
                               (256, 512, 7, 1, "cosine", torch.float32), (1024, 512, 7, 1, "cosine", torch.float32),
                               (256, 192, 14, 2, "norm", torch.bfloat16), (256, 192, 14, 2, "cosine", torch.float32),
                               (256, 960, 7, 1, "cosine", torch.float32), (16, 512, 7, 1, "cosine", torch.float32)]:
    ctor = dict(R=R, measure=meas, padding=R)
    if meas == "norm":
        ctor["p"] = 2
    m = NFPPooling(C, **ctor)
    x = torch.randn(B, C, S, S, device="cuda").to(dt).requires_grad_(True)
    go = torch.randn(B, m.out_channels, S, S, device="cuda").to(dt)
    row = {}
    for tag, env in (("fast", "0"), ("general", "1")):
        os.environ["NFP_FORCE_GENERIC"] = env
        _abi.load().nfp_reload_env()
        with torch.cuda.stream(s):
            o = m(x)
            fv = L.nfp_last_variant().decode()
            tf = time_kernel_graph(lambda: m(x), 20, s)
            tb = time_kernel_graph(lambda: torch.autograd.grad(o, x, go, retain_graph=True), 20, s)
            bv = L.nfp_last_variant().decode()
        row[tag] = (tf, tb, fv.split("<")[0], bv.split("<")[0])
    os.environ["NFP_FORCE_GENERIC"] = "0"
    _abi.load().nfp_reload_env()
    print(f"[{B},{C},{S},{S}] k{2*R+1} {meas} {str(dt).split('.')[1]}: "
          + "  ".join(f"{k}: fwd {v[0]:6.2f} bwd {v[1]:6.2f} ({v[2]}/{v[3]})" for k, v in row.items()), flush=True)
