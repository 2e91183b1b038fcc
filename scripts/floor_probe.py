#!/usr/bin/env python3
"""What does the chip allow at this size?  Plain torch kernels over the same 6.4 MB tensor."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import time_kernel_graph
dev = torch.device("cuda:0")
s = torch.cuda.Stream()
for B in (64, 256, 1024):
    x = torch.randn(B, 512, 7, 7, device=dev)
    y = torch.empty_like(x)
    with torch.cuda.stream(s):
        t_copy = time_kernel_graph(lambda: y.copy_(x), 50, s)
        t_sum = time_kernel_graph(lambda: x.sum(), 50, s)
        t_sum1 = time_kernel_graph(lambda: x.sum(dim=1), 50, s)
        t_mul = time_kernel_graph(lambda: torch.mul(x, 2.0, out=y), 50, s)
        t_empty = time_kernel_graph(lambda: y[:1, :1, :1, :1].zero_(), 50, s)
    mb = x.numel() * 4 / 1e6
    print(f"B={B} ({mb:.1f} MB): copy {t_copy:.2f} us ({2*mb/t_copy*1e-3:.2f} TB/s)  mul {t_mul:.2f} us  "
          f"sum {t_sum:.2f} us  sum(dim=1) {t_sum1:.2f} us ({mb/t_sum1*1e-3:.2f} TB/s)  tiny-kernel {t_empty:.2f} us")
