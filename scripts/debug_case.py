#!/usr/bin/env python3
"""Debug helper: one (B C H W R measure mode layout dtype) case on the product path against the float64 formulation —
where the gradients differ and where either side is NaN.  usage: python scripts/debug_case.py B C H W R measure mode nchw|nhwc f32|bf16"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from neighbour_feature_pooling_amd import NFPPooling, _abi
from neighbour_feature_pooling_amd._host import nfp_host
from neighbour_feature_pooling_amd.synth import feature_map
B, C, H, W, R = [int(v) for v in sys.argv[1:6]]
meas, mode, lay, dts = sys.argv[6:10]
dev = torch.device("cuda:0")
ctor = dict(R=R, measure="norm" if meas.startswith("norm") else meas, padding=R, padding_mode=mode)
if meas == "norm":
    ctor["p"] = 2
if meas == "norm1":
    ctor["p"] = 1
m = NFPPooling(C, **ctor)
dt = torch.bfloat16 if dts == "bf16" else torch.float32
seed = int(os.environ.get("SEED", "5"))
x = torch.from_numpy(feature_map((B, C, H, W), seed)).to(dev).to(dt)
if lay == "nhwc":
    x = x.contiguous(memory_format=torch.channels_last)
x.requires_grad_(True)
L = _abi.load()
out = m(x)
fv = L.nfp_last_variant().decode()
go = torch.from_numpy(feature_map(tuple(out.shape), seed + 1)).to(dev).to(dt)
gx, = torch.autograd.grad(out, x, go)
bv = L.nfp_last_variant().decode()
x64 = x.detach().double().contiguous().requires_grad_(True)
ref = nfp_host(x64, m.config)
gref, = torch.autograd.grad(ref, x64, go.double())
a, b = gx.float().cpu().numpy().astype(np.float64), gref.cpu().numpy()
print(fv, bv)
print("NaN in ours", int(np.isnan(a).sum()), "in ref", int(np.isnan(b).sum()), "of", a.size)
d = np.isnan(a) != np.isnan(b)
print("pattern differs at", int(d.sum()))
if d.any():
    idx = np.argwhere(d)
    print(" b:", np.unique(idx[:, 0])[:10], "c:", np.unique(idx[:, 1])[:10], "y:", np.unique(idx[:, 2]), "x:", np.unique(idx[:, 3]))
    for k in idx[:5]:
        print("  ", k, a[tuple(k)], b[tuple(k)])
ok = ~np.isnan(a) & ~np.isnan(b)
print("max err on numbers", np.max(np.abs(a[ok] - b[ok])) / np.max(np.abs(b[ok])))
o, r = out.float().detach().cpu().numpy(), ref.detach().cpu().numpy()
print("out: NaN ours", int(np.isnan(o).sum()), "ref", int(np.isnan(r).sum()), "zeros in ref", int((r == 0).sum()))
