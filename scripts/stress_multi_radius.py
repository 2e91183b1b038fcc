#!/usr/bin/env python3
"""Randomised stress of the one-pass radius-(1, 2) maps (MultiRadiusNFPPooling, nfp_heads.py:88-110) against the float64
torch formulation of cat([NFP_R1(x), NFP_R2(x)]).  usage: python scripts/stress_multi_radius.py [n] [seed]"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np, torch
from neighbour_feature_pooling_amd import MultiRadiusNFPPooling, _abi
from neighbour_feature_pooling_amd._host import nfp_host
from neighbour_feature_pooling_amd.synth import feature_map
from stress_tile import rel_err


def one_case(rnd, dev):
    H, W = rnd.randint(3, 20), rnd.randint(3, 20)
    C = 4 * rnd.randint(1, 24)
    B = rnd.choice([1, 2, 5, 33, 70, 300, 1100])
    if B * C * H * W > 6_000_000:
        B = 5
    meas = rnd.choice(["cosine", "cosine", "norm", "dot", "gfc", "rmse", "norm1", "emd"])
    mode = rnd.choice(["reflect", "zeros", "replicate"])
    cl, bf = rnd.random() < 0.5, rnd.random() < 0.25
    kw = dict(padding_mode=mode)
    if meas == "norm":
        kw["p"] = 2
    if meas == "norm1":     # the class default (nfp.py:16)
        meas, kw["p"] = "norm", 1
    m = MultiRadiusNFPPooling(C, R_list=(1, 2), measure=meas, **kw)
    dt = torch.bfloat16 if bf else torch.float32
    x = torch.from_numpy(feature_map((B, C, H, W), rnd.randint(0, 1 << 20))).to(dev).to(dt)
    if cl:
        x = x.contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True)
    L = _abi.load()
    out = m(x)
    fv = L.nfp_last_variant().decode()
    go = torch.from_numpy(feature_map(tuple(out.shape), rnd.randint(0, 1 << 20))).to(dev).to(dt)
    gx, = torch.autograd.grad(out, x, go)
    bv = L.nfp_last_variant().decode()
    x64 = x.detach().double().contiguous().requires_grad_(True)
    ref = torch.cat([nfp_host(x64, b.config) for b in m.nfp_blocks], 1)
    gref, = torch.autograd.grad(ref, x64, go.double())
    to, tg = (2e-5, 2e-5) if not bf else (1.5e-2, 3e-2)
    eo, eg = rel_err(out.float().detach().cpu().numpy(), ref.detach().cpu().numpy()), rel_err(gx.float().cpu().numpy(), gref.cpu().numpy())
    desc = f"B{B} C{C} {H}x{W} {meas} {mode} {'nhwc' if cl else 'nchw'} {'bf16' if bf else 'f32'}"
    return eo <= to and eg <= tg, desc, (eo, eg), (fv, bv)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
    dev = torch.device("cuda:0")
    bad, seen = 0, {}
    for i in range(n):
        ok, desc, errs, vs = one_case(rnd, dev)
        k = vs[0].split(",")[0] + " / " + vs[1].split(",")[0]
        seen[k] = seen.get(k, 0) + 1
        if not ok:
            bad += 1
            print("FAIL", desc, ["%.2e" % e for e in errs], vs, flush=True)
        torch.cuda.empty_cache()
    print(f"{n} cases, kernels {seen}, {bad} failed")
