import os, sys
sys.path.insert(0, os.getcwd())
import torch
from neighbour_feature_pooling_amd import NFPPooling, _abi
from bench import time_kernel_graph
L = _abi.load(); s = torch.cuda.Stream()
for B in (64, 256, 1024, 4096):
    m = NFPPooling(512, R=1, measure="cosine", padding=1)
    x = torch.randn(B, 512, 7, 7, device="cuda", requires_grad=True); go = torch.randn(B, 8, 7, 7, device="cuda")
    with torch.cuda.stream(s):
        o = m(x); fv = L.nfp_last_variant().decode()
        torch.autograd.grad(o, x, go, retain_graph=True); torch.cuda.synchronize(); bv = L.nfp_last_variant().decode()
        tf = time_kernel_graph(lambda: m(x), 20, s); tb = time_kernel_graph(lambda: torch.autograd.grad(o, x, go, retain_graph=True), 20, s)
    print(B, round(tf, 2), round(tb, 2), fv, bv, flush=True)
