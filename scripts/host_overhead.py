#!/usr/bin/env python3
"""Where does the eager host time of one NFP forward + backward go?  (cProfile over 3000 steps.)"""
import cProfile, pstats, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from neighbour_feature_pooling_amd import NFPPooling
m = NFPPooling(512, R=1, measure="cosine", padding=1)
x = torch.randn(64, 512, 7, 7, device="cuda", requires_grad=True)
go = torch.randn(64, 8, 7, 7, device="cuda")


def step():
    out = m(x)
    torch.autograd.grad(out, x, go)


for _ in range(200):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3000):
    step()
torch.cuda.synchronize()
print(f"eager: {(time.perf_counter() - t0) / 3000 * 1e6:.1f} us per fwd+bwd")
t0 = time.perf_counter()
with torch.no_grad():
    for _ in range(3000):
        m(x)
torch.cuda.synchronize()
print(f"eager forward only (no grad): {(time.perf_counter() - t0) / 3000 * 1e6:.1f} us")
pr = cProfile.Profile()
pr.enable()
for _ in range(3000):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
