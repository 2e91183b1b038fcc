#!/usr/bin/env python3
"""A/B kernel variants on ONE device in one gpurun call: each argument is a set of hipcc -D flags;
every variant is built, then all are timed round-robin (2 rounds) in fresh subprocesses.
    python scripts/ab_flags.py "" "-DNFP_UNROLL_F=2" "-DNFP_FWD_THREADS=1024 -DNFP_RB=2"
Optional env: AB_SHAPE="64,512,7,1,cosine"  (B,C,S,R,measure)
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neighbour_feature_pooling_amd import build

variants = sys.argv[1:] or [""]
libs = []
for i, fl in enumerate(variants):
    lib = os.path.join(ROOT, "gpurun_out", f"libnfp_ab_{i}.so")
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    subprocess.check_call([build.hipcc_path()] + build.HIPCC_FLAGS + ["-w"] + fl.split() + ["-o", lib,
                          os.path.join(build.CSRC, "nfp_hip.hip")])
    libs.append(lib)
B, C, S, R, meas = (os.environ.get("AB_SHAPE") or "64,512,7,1,cosine").split(",")
code = """
import sys; sys.path.insert(0, {root!r})
import torch
from neighbour_feature_pooling_amd import _abi
_abi.LIB_PATH = {lib!r}
from neighbour_feature_pooling_amd import NFPPooling
from bench import time_kernel_graph
ctor = dict(R={R}, measure={meas!r}, padding={R})
if {meas!r} == 'norm': ctor['p'] = 2
m = NFPPooling({C}, **ctor)
x = torch.randn({B}, {C}, {S}, {S}, device='cuda', requires_grad=True)
go = torch.randn({B}, m.out_channels, {S}, {S}, device='cuda')
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    out = m(x)
    tf = time_kernel_graph(lambda: m(x), 50, s)
    tb = time_kernel_graph(lambda: torch.autograd.grad(out, x, go, retain_graph=True), 50, s)
print(f"[{{{tag!r}:40s}}] fwd {{tf:6.2f}} us   bwd {{tb:6.2f}} us   sum {{tf+tb:6.2f}}")
"""
for rnd in range(2):
    for fl, lib in zip(variants, libs):
        subprocess.check_call([sys.executable, "-c", code.format(root=ROOT, lib=lib, R=R, meas=meas, C=C, B=B, S=S,
                                                                 tag=fl or "(default)")])
