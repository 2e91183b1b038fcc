#!/usr/bin/env python3
"""Batch / shape sweep of the hot-path kernels: kernel time and achieved algorithmic GB/s."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from neighbour_feature_pooling_amd import NFPPooling, _abi
from bench import time_kernel_graph, algorithmic_bytes
L = _abi.load()
s = torch.cuda.Stream()
rows = []
cfgs = [(B, 512, 7, 1, "cosine", torch.float32) for B in (16, 64, 256, 1024, 4096)]
cfgs += [(256, 512, 2, 1, "cosine", torch.float32), (256, 960, 7, 1, "cosine", torch.float32),
         (64, 192, 14, 2, "norm", torch.bfloat16), (256, 192, 14, 2, "norm", torch.bfloat16),
         (256, 192, 14, 2, "norm", torch.float32), (256, 192, 14, 1, "cosine", torch.float32)]
for B, C, S, R, meas, dt in cfgs:
    ctor = dict(R=R, measure=meas, padding=R)
    if meas == "norm": ctor["p"] = 2
    m = NFPPooling(C, **ctor)
    x = torch.randn(B, C, S, S, device="cuda").to(dt).requires_grad_(True)
    go = torch.randn(B, m.out_channels, S, S, device="cuda").to(dt)
    reps = 50 if B <= 256 else 10
    with torch.cuda.stream(s):
        out = m(x)
        fv = L.nfp_last_variant().decode()
        tf = time_kernel_graph(lambda: m(x), reps, s)
        tb = time_kernel_graph(lambda: torch.autograd.grad(out, x, go, retain_graph=True), reps, s)
    e = 4 if dt == torch.float32 else 2
    fb, bb = algorithmic_bytes(B, C, S * S, m.out_channels, e)
    px = B * S * S
    rows.append(dict(shape=[B, C, S, S], k=2 * R + 1, measure=meas, dtype=str(dt).split(".")[1], fwd_us=round(tf, 2),
                     bwd_us=round(tb, 2), fwd_GBs=round(fb / tf / 1e3), bwd_GBs=round(bb / tb / 1e3),
                     Mpx_s=round(px / (tf + tb), 1), variant=fv))
    print(json.dumps(rows[-1]), flush=True)
