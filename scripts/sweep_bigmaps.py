#!/usr/bin/env python3
"""Maps above 512 pixels (MobileNetV3_MultiStageNFP, texture_pooling.py:211-268; RESNET18_NFP_AT_LAYER, resnet18.py:410-468):
kernel time and achieved algorithmic GB/s per launch.  usage: python scripts/sweep_bigmaps.py [out.jsonl] [B]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from neighbour_feature_pooling_amd import NFPPooling, _abi
from bench import time_kernel_graph, algorithmic_bytes
L = _abi.load()


def warm(fn, ms=40.0):
    """clocks and caches: run `fn` for ~ms before anything is timed (the first configuration measured after an idle
    moment otherwise reads 10-20 % slow)"""
    import time
    t = time.perf_counter()
    while (time.perf_counter() - t) * 1e3 < ms:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()


s = torch.cuda.Stream()
out_f = open(sys.argv[1], "w") if len(sys.argv) > 1 else None
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
shapes = [(16, 112), (24, 56), (40, 28), (128, 28), (64, 56), (112, 14), (256, 14), (960, 7)]
for C, S in shapes:
    for layout in ("nchw", "nhwc", "nhwc-bf16"):
        m = NFPPooling(C, R=1, measure="cosine", padding=1)
        x = torch.randn(B, C, S, S, device="cuda")
        es = 4
        if layout == "nhwc-bf16":       # (the channels-last bf16 training flow: BASELINE configs 4 / 5)
            x = x.bfloat16()
            es = 2
        if layout != "nchw":
            x = x.contiguous(memory_format=torch.channels_last)
        x.requires_grad_(True)
        go = torch.randn(B, 8, S, S, device="cuda").to(x.dtype)
        with torch.cuda.stream(s):
            o = m(x)
            fv = L.nfp_last_variant().decode()
            torch.autograd.grad(o, x, go, retain_graph=True)
            torch.cuda.synchronize()
            bv = L.nfp_last_variant().decode()
            warm(lambda: m(x))
            tf = time_kernel_graph(lambda: m(x), 10, s)
            tb = time_kernel_graph(lambda: torch.autograd.grad(o, x, go, retain_graph=True), 10, s)
        fb, bb = algorithmic_bytes(B, C, S * S, 8, es)
        row = dict(shape=[B, C, S, S], layout=layout, fwd_us=round(tf, 2), bwd_us=round(tb, 2), fwd_GBs=round(fb / tf / 1e3),
                   bwd_GBs=round(bb / tb / 1e3), fwd_frac=round(fb / tf / 1e3 / 8000, 3), bwd_frac=round(bb / tb / 1e3 / 8000, 3),
                   fwd_variant=fv, bwd_variant=bv)
        print(json.dumps(row), flush=True)
        if out_f:
            out_f.write(json.dumps(row) + "\n"); out_f.flush()
        del x, go, o
        torch.cuda.empty_cache()
