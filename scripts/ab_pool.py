#!/usr/bin/env python3
"""A/B of the fused pooling tail on large maps across the libraries of neighbour_feature_pooling_amd/ab/manifest.json
(scripts/ab_flags.py --build): forward / backward kernel time per library, fresh subprocess each."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = """
import os, sys; sys.path.insert(0, {root!r})
os.environ["NFP_PY_NODES"] = "1"
import torch
from neighbour_feature_pooling_amd import _abi
_abi.LIB_PATH = {lib!r}
from neighbour_feature_pooling_amd import NFPPooling
from neighbour_feature_pooling_amd.functional import nfp_pool
from bench import time_kernel_graph
s = torch.cuda.Stream()
for shape in ((256, 16, 112, 112), (256, 24, 56, 56), (256, 40, 28, 28), (256, 64, 56, 56)):
    m = NFPPooling(shape[1], R=1, measure="cosine", padding=1)
    x = torch.randn(*shape, device="cuda", requires_grad=True)
    with torch.cuda.stream(s):
        o = nfp_pool(x, m.config)
        fv = _abi.load().nfp_last_variant().decode()
        gos = tuple(torch.randn_like(v) for v in o)
        torch.autograd.grad(o, x, gos, retain_graph=True)
        torch.cuda.synchronize()
        for _ in range(30): nfp_pool(x, m.config)
        torch.cuda.synchronize()
        tf = time_kernel_graph(lambda: nfp_pool(x, m.config), 10, s)
        tb = time_kernel_graph(lambda: torch.autograd.grad(o, x, gos, retain_graph=True), 10, s)
    print(f"[{{{tag!r}:24s}}] {{list(shape)}} pooled fwd {{tf:7.2f}} us  bwd {{tb:7.2f}} us  {{fv}}")
"""
for rnd in range(2):
    for v in json.load(open(os.path.join(ROOT, "neighbour_feature_pooling_amd", "ab", "manifest.json"))):
        code = CODE.format(root=ROOT, lib=os.path.join(ROOT, v["lib"]), tag=v["flags"] or "(default)")
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
        sys.stdout.write(r.stdout if r.returncode == 0 else r.stderr[-800:])
        sys.stdout.flush()
