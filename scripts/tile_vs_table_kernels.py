#!/usr/bin/env python3
"""Maps the table kernels serve (<= 512 pixels): their kernel times beside the row-band kernels' (NFP_TILE_FIRST=1), same
process, alternating.  usage: python scripts/tile_vs_table_kernels.py [out.jsonl]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NFP_PY_NODES"] = "1"
import torch
from neighbour_feature_pooling_amd import NFPPooling, _abi
from bench import time_kernel_graph
L = _abi.load()
s = torch.cuda.Stream()
out_f = open(sys.argv[1], "w") if len(sys.argv) > 1 else None


def warm(fn, ms=30.0):
    import time
    t = time.perf_counter()
    while (time.perf_counter() - t) * 1e3 < ms:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()


cases = [(64, 512, 7, 1, "cosine", "nchw", "f32"), (256, 512, 7, 1, "cosine", "nchw", "f32"), (4096, 512, 7, 1, "cosine", "nchw", "f32"),
         (256, 960, 7, 1, "cosine", "nchw", "f32"), (256, 112, 14, 1, "cosine", "nchw", "f32"), (256, 256, 14, 1, "cosine", "nchw", "f32"),
         (256, 192, 14, 2, "norm", "nchw", "f32"), (256, 192, 14, 2, "norm", "nhwc", "bf16"), (256, 256, 14, 1, "cosine", "nhwc", "bf16"),
         (64, 512, 7, 2, "cosine", "nchw", "f32"),
         # round 4: where do the row-band kernels (bf16 slots since round 4; dense channels-last stores) beat the tables?
         (256, 192, 14, 2, "norm", "nhwc", "f32"), (256, 192, 14, 2, "cosine", "nchw", "f32"), (256, 256, 14, 2, "norm", "nchw", "f32"),
         (256, 112, 14, 2, "cosine", "nchw", "f32"), (256, 192, 14, 1, "cosine", "nhwc", "f32"), (256, 256, 14, 1, "cosine", "nhwc", "f32"),
         (256, 200, 14, 2, "norm", "nhwc", "bf16"), (256, 512, 7, 1, "cosine", "nhwc", "f32"), (256, 960, 7, 1, "cosine", "nhwc", "f32")]
for B, C, S, R, meas, lay, dt in cases:
    ctor = dict(R=R, measure=meas, padding=R)
    if meas == "norm":
        ctor["p"] = 2
    m = NFPPooling(C, **ctor)
    x = torch.randn(B, C, S, S, device="cuda").to(torch.bfloat16 if dt == "bf16" else torch.float32)
    if lay == "nhwc":
        x = x.contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True)
    go = torch.randn(B, m.out_channels, S, S, device="cuda").to(x.dtype)
    for tf_ in ("0", "1", "0", "1"):
        os.environ["NFP_TILE_FIRST"] = tf_
        L.nfp_reload_env()
        with torch.cuda.stream(s):
            o = m(x)
            fv = L.nfp_last_variant().decode()
            torch.autograd.grad(o, x, go, retain_graph=True)
            torch.cuda.synchronize()
            bv = L.nfp_last_variant().decode()
            warm(lambda: m(x))
            tf = time_kernel_graph(lambda: m(x), 20, s)
            tb = time_kernel_graph(lambda: torch.autograd.grad(o, x, go, retain_graph=True), 20, s)
        row = dict(shape=[B, C, S, S], R=R, measure=meas, layout=lay, dtype=dt, tile_first=int(tf_), fwd_us=round(tf, 2), bwd_us=round(tb, 2),
                   fwd=fv, bwd=bv)
        print(json.dumps(row), flush=True)
        if out_f:
            out_f.write(json.dumps(row) + "\n"); out_f.flush()
    del x, go, o
    torch.cuda.empty_cache()
os.environ["NFP_TILE_FIRST"] = "0"
L.nfp_reload_env()
