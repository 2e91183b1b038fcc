#!/usr/bin/env python3
"""Diagnostic: kernel time of fwd_fast with phases compiled out (results are wrong on purpose)."""
import ctypes, os, subprocess, sys, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from neighbour_feature_pooling_amd import build
from bench import time_kernel_graph

def run(mask):
    lib = os.path.join(ROOT, "gpurun_out", f"libnfp_ablate_{mask}.so")
    subprocess.check_call([build.hipcc_path()] + build.HIPCC_FLAGS + ["-w", f"-DNFP_ABLATE={mask}", "-o", lib,
                          os.path.join(build.CSRC, "nfp_hip.hip")])
    code = f"""
import sys; sys.path.insert(0, {ROOT!r})
import torch
from neighbour_feature_pooling_amd import _abi
_abi.LIB_PATH = {lib!r}
from neighbour_feature_pooling_amd import NFPPooling
from bench import time_kernel_graph
m = NFPPooling(512, R=1, measure='cosine', padding=1)
x = torch.randn(64, 512, 7, 7, device='cuda')
s = torch.cuda.Stream()
with torch.cuda.stream(s), torch.no_grad():
    t = time_kernel_graph(lambda: m(x), 50, s)
print(f"ablate mask {mask:2d}: fwd {{t:.2f}} us")
"""
    subprocess.check_call([sys.executable, "-c", code])

for mask in [0, 1, 2 | 1, 4, 8, 1 | 8, 1 | 2 | 8, 1 | 2 | 4 | 8]:
    run(mask)
