#!/usr/bin/env python3
"""Every measure with a HIP kernel at the headline shape (and config 5's): kernel times of the generic path
next to the hot-path kernels.  usage: python scripts/sweep_measures.py [out.jsonl]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from neighbour_feature_pooling_amd import NFPPooling, _abi
from bench import time_kernel_graph, algorithmic_bytes
L = _abi.load()
s = torch.cuda.Stream()
out_f = open(sys.argv[1], "w") if len(sys.argv) > 1 else None
shapes = [(64, 512, 7, 1), (256, 192, 14, 2), (256, 64, 56, 1)]   # headline, config 5 (f32), a RESNET18_NFP_AT_LAYER map
for B, C, S, R in shapes:
    for meas in _abi.MEASURES:
        if meas == "scs":
            continue
        for p in ((1, 2) if meas == "norm" else (2,)):
            m = NFPPooling(C, R=R, measure=meas, p=p, padding=R)
            x = (torch.rand(B, C, S, S, device="cuda") + 0.05).requires_grad_(True)   # positive: valid for every measure
            go = torch.randn(B, m.out_channels, S, S, device="cuda")
            with torch.cuda.stream(s):
                o = m(x)
                fv = L.nfp_last_variant().decode()
                torch.autograd.grad(o, x, go, retain_graph=True)
                torch.cuda.synchronize()
                bv = L.nfp_last_variant().decode()
                tf = time_kernel_graph(lambda: m(x), 20, s)
                tb = time_kernel_graph(lambda: torch.autograd.grad(o, x, go, retain_graph=True), 20, s)
            fb, bb = algorithmic_bytes(B, C, S * S, m.out_channels, 4)
            row = dict(shape=[B, C, S, S], k=2 * R + 1, measure=meas, p=p, fwd_us=round(tf, 2), bwd_us=round(tb, 2),
                       fwd_GBs=round(fb / tf / 1e3), bwd_GBs=round(bb / tb / 1e3),
                       Mpx_s=round(B * S * S / (tf + tb), 1), variant=fv, bwd_variant=bv)
            print(json.dumps(row), flush=True)
            if out_f:
                out_f.write(json.dumps(row) + "\n"); out_f.flush()
