#!/usr/bin/env python3
"""A/B of the row-band launchers' constants (NFP_TILE_WGS / NFP_TILE_LDS_KB / NFP_TILE_CAP, read by nfp_reload_env) on the
MultiStage / at-layer maps: kernel time per setting.  usage: python scripts/tune_tile.py [out.jsonl] [B] [layout]"""
import os, sys, json, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NFP_PY_NODES"] = "1"
import torch
from neighbour_feature_pooling_amd import NFPPooling, _abi
from bench import time_kernel_graph
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
layout = sys.argv[3] if len(sys.argv) > 3 else "nchw"
shapes = [(16, 112), (24, 56), (40, 28), (128, 28), (64, 56)]
settings = [dict(NFP_TILE_WGS=w, NFP_TILE_LDS_KB=l, NFP_TILE_CAP=c) for w, l, c in
            [(512, 78, 1024), (512, 78, 1024), (1024, 78, 1024), (1536, 78, 1024), (2048, 78, 1024), (1024, 78, 320), (2048, 52, 320)]]
from bench import time_graph
for C, S in shapes:
    m = NFPPooling(C, R=1, measure="cosine", padding=1)
    # rotating input sets, more than the Infinity Cache holds: every byte from HBM, as in training
    nset = max(2, min(12, (600 << 20) // (B * C * S * S * 4)))
    xs = []
    for _ in range(nset):
        x = torch.randn(B, C, S, S, device="cuda")
        if layout == "nhwc":
            x = x.contiguous(memory_format=torch.channels_last)
        xs.append(x.requires_grad_(True))
    x = xs[0]
    go = torch.randn(B, 8, S, S, device="cuda")
    for st in settings:
        for k, v in st.items():
            os.environ[k] = str(v)
        L.nfp_reload_env()
        with torch.cuda.stream(s):
            o = m(x)
            fv = L.nfp_last_variant().decode()
            torch.autograd.grad(o, x, go, retain_graph=True)
            torch.cuda.synchronize()
            bv = L.nfp_last_variant().decode()
            warm(lambda: m(x))
            outs = [m(xx) for xx in xs]
            tf = time_graph([(lambda xx=xx: m(xx)) for xx in xs], s)
            tb = time_graph([(lambda oo=oo, xx=xx: torch.autograd.grad(oo, xx, go, retain_graph=True)) for oo, xx in zip(outs, xs)], s)
        row = dict(shape=[B, C, S, S], layout=layout, **st, fwd_us=round(tf, 2), bwd_us=round(tb, 2), fwd=fv, bwd=bv)
        print(json.dumps(row), flush=True)
        if out_f:
            out_f.write(json.dumps(row) + "\n"); out_f.flush()
    del x, go, o, xs, outs
    torch.cuda.empty_cache()
