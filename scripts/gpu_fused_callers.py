#!/usr/bin/env python3
"""Kernel times of the two fused callers of the path at the headline shape (run on the GPU box):
the nfp_pooling tail (GAP(x), GAP(NFP(x)) from one pass; NFP_Pooling.py:27-31) and the multi-radius maps (nfp_heads.py:88-110),
each against the composition of the plain ops it replaces.  One JSON line per case."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import time_kernel_graph
from neighbour_feature_pooling_amd import NFPPooling, MultiRadiusNFPPooling, _abi
from neighbour_feature_pooling_amd.functional import nfp_pool, nfp_pooled

dev = torch.device("cuda", 0)
s = torch.cuda.Stream()
L = _abi.load()


def case(name, fwd, x, grads_of):
    with torch.cuda.stream(s):
        outs = fwd()
        fv = L.nfp_last_variant().decode()
        gos = grads_of(outs)
        torch.autograd.grad(outs, x, gos, retain_graph=True)
        torch.cuda.synchronize()
        bv = L.nfp_last_variant().decode()
        reps = 50 if x.numel() < (1 << 24) else 10
        tf = time_kernel_graph(fwd, reps, s)
        tb = time_kernel_graph(lambda: torch.autograd.grad(outs, x, gos, retain_graph=True), reps, s)
    print(json.dumps({"case": name, "fwd_us": round(tf, 2), "bwd_us": round(tb, 2), "fwd": fv, "bwd": bv}))


# ... and the maps MobileNetV3_MultiStageNFP averages at once (texture_pooling.py:249-252): row-band kernels + pool_fold
for shape, dt, cl, R, meas in (((64, 512, 7, 7), torch.float32, False, 1, "cosine"),
                               ((256, 192, 14, 14), torch.bfloat16, True, 2, "norm"),
                               ((256, 16, 112, 112), torch.float32, False, 1, "cosine"),
                               ((256, 24, 56, 56), torch.float32, False, 1, "cosine"),
                               ((256, 40, 28, 28), torch.float32, False, 1, "cosine"),
                               ((256, 16, 112, 112), torch.bfloat16, True, 1, "cosine"),
                               ((256, 40, 28, 28), torch.bfloat16, True, 1, "cosine")):
    x = torch.randn(*shape, device=dev).to(dt)
    if cl:
        x = x.contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True)
    ctor = dict(R=R, measure=meas, padding=R)
    if meas == "norm":
        ctor["p"] = 2
    m = NFPPooling(shape[1], **ctor)
    tag = f"{list(shape)} {str(dt).split('.')[-1]} {'nhwc' if cl else 'nchw'} k={2 * R + 1} {meas}"
    case(tag + " | fused pooling tail", lambda: nfp_pool(x, m.config), x,
         lambda o: tuple(torch.randn_like(v) for v in o))
    # round 4: the pooled maps ALONE (texture_pooling.py:251-252: no GAP(x)); and without a gradient to follow (no map stores)
    case(tag + " | pooled NFP only (nfp_pooled)", lambda: (nfp_pooled(x, m.config),), x, lambda o: tuple(torch.randn_like(v) for v in o))
    with torch.no_grad(), torch.cuda.stream(s):
        t0 = time_kernel_graph(lambda: nfp_pooled(x, m.config), 50 if x.numel() < (1 << 24) else 10, s)
    print(json.dumps({"case": tag + " | pooled NFP only, no_grad (no map stores)", "fwd_us": round(t0, 2), "fwd": L.nfp_last_variant().decode()}))
    with torch.cuda.stream(s):
        t1 = time_kernel_graph(lambda: m(x), 50 if x.numel() < (1 << 24) else 10, s)
    print(json.dumps({"case": tag + " | plain NFP forward (maps only)", "fwd_us": round(t1, 2), "fwd": L.nfp_last_variant().decode()}))
    case(tag + " | x.mean + nfp + mean (what it replaces)",
         lambda: (x.float().mean((2, 3)), m(x).float().mean((2, 3))), x, lambda o: tuple(torch.randn_like(v) for v in o))
x = torch.randn(64, 512, 7, 7, device=dev, requires_grad=True)
mr = MultiRadiusNFPPooling(512, R_list=(1, 2), measure="cosine")
case("[64, 512, 7, 7] f32 | multi-radius (1, 2), one pass", lambda: mr(x), x, lambda o: torch.randn_like(o))
m1, m2 = NFPPooling(512, R=1, measure="cosine", padding=1), NFPPooling(512, R=2, measure="cosine", padding=2)
case("[64, 512, 7, 7] f32 | cat(nfp_R1, nfp_R2) (what it replaces)", lambda: torch.cat([m1(x), m2(x)], 1), x,
     lambda o: torch.randn_like(o))
