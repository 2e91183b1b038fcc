#!/usr/bin/env python3
"""What rocprofv3 profiles for the large-map kernels: forward + backward of NFP(cosine, k=3) on the MultiStage / at-layer
maps at B = 256 (texture_pooling.py:211-268, resnet18.py:410-468), rotating over a few input sets, nothing else."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from neighbour_feature_pooling_amd import NFPPooling
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
SHAPES = ((16, 112), (24, 56), (40, 28), (128, 28), (64, 56))
if len(sys.argv) > 3:          # one shape only: run_bigmaps_for_rocprof.py B C S [nchw|nhwc|nhwc-bf16]
    SHAPES = ((int(sys.argv[2]), int(sys.argv[3])),)
LAY = sys.argv[4] if len(sys.argv) > 4 else "nchw"
DT = torch.bfloat16 if LAY.endswith("bf16") else torch.float32
for C, S in SHAPES:
    m = NFPPooling(C, R=1, measure="cosine", padding=1)
    xs = [torch.randn(B, C, S, S, device="cuda").to(DT) for _ in range(3)]
    if LAY.startswith("nhwc"):
        xs = [x.contiguous(memory_format=torch.channels_last) for x in xs]
    xs = [x.requires_grad_(True) for x in xs]
    go = torch.randn(B, 8, S, S, device="cuda").to(DT)
    for i in range(9):
        x = xs[i % 3]
        out = m(x)
        torch.autograd.grad(out, x, go)
    torch.cuda.synchronize()
    del xs, go, out
    torch.cuda.empty_cache()
