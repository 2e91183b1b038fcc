#!/usr/bin/env python3
"""fwd_gram (matrix cores) against fwd_fast and the float64 formulation, then its kernel time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from neighbour_feature_pooling_amd import NFPPooling, _abi
from neighbour_feature_pooling_amd._host import nfp_host
from bench import time_kernel_graph
L = _abi.load()
dev = "cuda"
torch.manual_seed(0)
worst = 0.0
for (B, C, H, W, R, meas, mode, kind) in [(4, 192, 14, 14, 2, "norm", "reflect", "randn"), (4, 192, 14, 14, 2, "cosine", "reflect", "randn"),
                                          (2, 64, 7, 7, 1, "cosine", "reflect", "randn"), (3, 32, 5, 9, 1, "norm", "replicate", "randn"),
                                          (3, 48, 6, 5, 2, "cosine", "zeros", "relu"), (2, 16, 9, 9, 1, "norm", "zeros", "const"),
                                          (2, 192, 14, 14, 2, "norm", "reflect", "smooth"), (1, 512, 16, 16, 2, "norm", "reflect", "randn")]:
    ctor = dict(R=R, measure=meas, padding=R, padding_mode=mode)
    if meas == "norm":
        ctor["p"] = 2
    m = NFPPooling(C, **ctor)
    x = torch.randn(B, C, H, W, device=dev)
    if kind == "relu":
        x = x.relu()
    if kind == "const":
        x = torch.ones_like(x) * 0.37 + (torch.arange(C, device=dev).view(1, C, 1, 1) * 0.01)
    if kind == "smooth":
        x = x.mean((2, 3), keepdim=True) + 0.01 * x      # neighbours nearly identical: the Gram form's hard case
    x = x.bfloat16()
    if B != 3:
        x = x.contiguous(memory_format=torch.channels_last)      # (the B = 3 cases stay NCHW)
    outs = {}
    for env in ("1", "0"):
        os.environ["NFP_MFMA"] = env
        _abi.load().nfp_reload_env()
        outs[env] = m(x).float()
        var = L.nfp_last_variant().decode()
        assert var.startswith("fwd_gram" if env == "1" else "fwd_fast"), var
    ref = nfp_host(x.double(), m.config).float()
    sc = ref.abs().max().item() + 1e-30
    e_g, e_f = (outs["1"] - ref).abs().max().item() / sc, (outs["0"] - ref).abs().max().item() / sc
    worst = max(worst, e_g)
    print(f"[{B},{C},{H},{W}] R{R} {meas} {mode} {kind}: gram err {e_g:.2e}  fast err {e_f:.2e}  (|ref|max {sc:.3g})", flush=True)
    if kind == "const" and meas == "norm":
        inside = outs["1"][:, :, 1:-1, 1:-1]
        print("   identical neighbours -> max |d| inside:", inside.abs().max().item())
print("worst gram err", worst)
s = torch.cuda.Stream()
for (B, C, S, R, meas, cl) in [(256, 192, 14, 2, "norm", True), (256, 192, 14, 2, "cosine", True), (64, 512, 7, 1, "cosine", True),
                               (256, 192, 14, 2, "norm", False), (64, 512, 7, 1, "cosine", False), (256, 512, 7, 1, "cosine", False)]:
    ctor = dict(R=R, measure=meas, padding=R)
    if meas == "norm":
        ctor["p"] = 2
    m = NFPPooling(C, **ctor)
    x = torch.randn(B, C, S, S, device=dev).bfloat16()
    if cl:
        x = x.contiguous(memory_format=torch.channels_last)
    res = {}
    for env in ("1", "0"):
        os.environ["NFP_MFMA"] = env
        _abi.load().nfp_reload_env()
        with torch.cuda.stream(s), torch.no_grad():
            m(x)
            res[env] = (time_kernel_graph(lambda: m(x), 20, s), L.nfp_last_variant().decode().split("<")[0])
    print(f"[{B},{C},{S},{S}] k{2*R+1} {meas} bf16 {'nhwc' if cl else 'nchw'}: " + "  ".join(f"{v[1]} {v[0]:.2f} us" for v in res.values()), flush=True)
