#!/usr/bin/env python3
"""Diagnostic (never the product build): compile libnfp_hip_diag.so with -DNFP_STAMPS and print,
per kernel, the median workgroup's time between phase stamps.  Read SHARES, not totals.
    python scripts/diag_stamps.py [B C S R measure]
"""
import ctypes, os, subprocess, sys
os.environ["NFP_PY_NODES"] = "1"   # the C++ autograd nodes link the product library; the diagnostic one is loaded by ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from neighbour_feature_pooling_amd import _abi, build

# built in-tree (git-ignored *.so) so that a copy compiled in the CPU container travels to the GPU box
diag = os.path.join(ROOT, "neighbour_feature_pooling_amd", "libnfp_hip_diag.so")
if not os.path.exists(diag) or os.path.getmtime(diag) < build._newest_source_mtime():
    build.compile_hip(diag, ["-DNFP_STAMPS"])
if "--build-only" in sys.argv:
    sys.exit(0)
_abi.LIB_PATH = diag
L = _abi.load()
L.nfp_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
from neighbour_feature_pooling_amd import NFPPooling

B, C, S, R = [int(v) for v in (sys.argv[1:5] or [64, 512, 7, 1])]
meas = sys.argv[5] if len(sys.argv) > 5 else "cosine"
dev = torch.device("cuda:0")
ctor = dict(R=R, measure=meas, padding=R)
if meas == "norm":
    ctor["p"] = 2
m = NFPPooling(C, **ctor)
DT = torch.bfloat16 if os.environ.get("DIAG_BF16") == "1" else torch.float32
x = torch.randn(B, C, S, S, device=dev).to(DT)
if os.environ.get("DIAG_NHWC") == "1":
    x = x.contiguous(memory_format=torch.channels_last)
x.requires_grad_(True)
go = torch.randn(B, m.out_channels, S, S, device=dev).to(DT)
buf = torch.zeros(8192 * 16 * 2, dtype=torch.int64, device=dev)
assert L.nfp_debug_set_stamp_buffer(buf.data_ptr()) == 0


def report(name, nwg, nst):
    if nwg == 0:
        print(f"{name}: no stamps in this kernel")
        return
    a = buf.cpu().numpy().reshape(-1, 16, 2)[:nwg, :max(nst, 7)]
    clk, wall = a[..., 0].astype(np.float64), a[..., 1].astype(np.float64)
    clk_all = clk; clk, wall = clk[:, :nst], wall[:, :nst]
    t0 = wall[:, 0].min()
    start = (wall[:, 0] - t0) * 10.0          # ns after the first workgroup started
    end = (wall[:, nst - 1] - t0) * 10.0
    dclk = np.diff(clk, axis=1)
    dwall = np.diff(wall, axis=1) * 10.0
    tot_clk, tot_ns = clk[:, -1] - clk[:, 0], (wall[:, -1] - wall[:, 0]) * 10.0
    print(f"{name}: {nwg} workgroups; WG start spread {start.max():.0f} ns; last WG ends at {end.max():.0f} ns; "
          f"per-WG body median {np.median(tot_ns):.0f} ns = {np.median(tot_clk):.0f} clk "
          f"(~{np.median(tot_clk) / max(np.median(tot_ns), 1) :.2f} GHz)")
    full = buf.cpu().numpy().reshape(-1, 16, 2)[:nwg, :, 0].astype(np.float64)
    extra = [i for i in range(nst, 16) if np.median(full[:, i]) > 0]
    if extra:
        print("   extra stamps (clk after stamp 0): " + ", ".join(f"{i}: {np.median(full[:, i] - full[:, 0]):.0f}" for i in extra))
    for i in range(nst - 1):
        print(f"   phase {i}->{i + 1}: median {np.median(dclk[:, i]):8.0f} clk  {np.median(dwall[:, i]):7.0f} ns   "
              f"max {dclk[:, i].max():8.0f} clk")


for it in range(3):
    for _ in range(20):   # warm clocks/caches
        out = m(x)
        torch.autograd.grad(out, x, go, retain_graph=True)
    torch.cuda.synchronize()
    buf.zero_()
    out = m(x)
    torch.cuda.synchronize()
    var = L.nfp_last_variant().decode()
    if it == 2:
        nwg = int((buf.view(-1, 16, 2)[:, 0, 0] != 0).sum().item())
        report(var, nwg, 6)
    buf.zero_()
    torch.autograd.grad(out, x, go)
    torch.cuda.synchronize()
    var = L.nfp_last_variant().decode()
    if it == 2:
        nwg = int((buf.view(-1, 16, 2)[:, 0, 0] != 0).sum().item())
        report(var, nwg, 7)
