#!/usr/bin/env python3
"""Randomised stress of the table kernels at batches of 1024+ images (quarter-size workgroups, four per CU) against the
float64 torch formulation.  usage: python scripts/stress_big_batch.py [n] [seed]"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from neighbour_feature_pooling_amd import NFPPooling, _abi
from neighbour_feature_pooling_amd._host import nfp_host
from neighbour_feature_pooling_amd.functional import nfp_pool, nfp_pool_fused_ok
from neighbour_feature_pooling_amd.synth import feature_map
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from stress_tile import rel_err


def one_case(rnd, dev):
    R = rnd.choice([1, 1, 2])
    hw = int(os.environ.get("STRESS_HW", "16"))
    H, W = rnd.randint(R + 1, hw), rnd.randint(R + 1, hw)
    if H * W < 4:
        H = W = 3
    C = 4 * rnd.randint(1, 8 if R == 2 else 16)
    lo, hi = [int(v) for v in os.environ.get("STRESS_B", "1024,1400").split(",")]   # (STRESS_B=1,70: the same generator at small batches)
    B = rnd.randint(lo, hi)
    meas = rnd.choice(["cosine", "cosine", "norm", "norm1", "dot", "gfc", "rmse"])
    mode = rnd.choice(["reflect", "zeros", "replicate"])
    cl, bf = rnd.random() < 0.5, rnd.random() < 0.25
    if os.environ.get("STRESS_SYM") == "1":   # the five measures of csrc/nfp_measures.h::kSymTerm (table kernels here)
        meas = rnd.choice(["geman", "canberra", "hellinger", "squaredchord", "chisquared1", "jeffrey"])
        if meas == "hellinger" and (mode == "replicate" or R == 2):   # (a pixel against its own copy: NaN, and how far it travels differs — DESIGN.md section 7)
            mode = "zeros"
    if os.environ.get("STRESS_C32") == "1":   # the matrix-core kernels: bf16, whole 32-channel tiles (NFP_GEMM3=2 / 0 pick the backward's form)
        C, bf = 32 * rnd.randint(1, 8), True
        meas = rnd.choice(["cosine", "norm", "dot", "gfc", "rmse"])
    ctor = dict(R=R, measure="norm" if meas.startswith("norm") else meas, padding=R, padding_mode=mode)
    if meas == "norm":
        ctor["p"] = 2
    if meas == "norm1":
        ctor["p"] = 1
    m = NFPPooling(C, **ctor)
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
    ref = nfp_host(x64, m.config)
    gref, = torch.autograd.grad(ref, x64, go.double(), retain_graph=True)
    to, tg = (2e-5, 2e-5) if not bf else (1.5e-2, 3e-2)
    eo, eg = rel_err(out.float().detach().cpu().numpy(), ref.detach().cpu().numpy()), rel_err(gx.float().cpu().numpy(), gref.cpu().numpy())
    ok = eo <= to and eg <= tg
    ep = egp = 0.0
    pv = ""
    if nfp_pool_fused_ok(x, m.config):
        gap, nfpm = nfp_pool(x, m.config)
        pv = L.nfp_last_variant().decode()
        wg = torch.from_numpy(feature_map((B, C), 11)).to(dev)
        wn = torch.from_numpy(feature_map((B, m.out_channels), 12)).to(dev)
        gp, = torch.autograd.grad((gap * wg).sum() + (nfpm * wn).sum(), x)
        rg, rn = x64.mean((2, 3)), ref.mean((2, 3))
        gpref, = torch.autograd.grad((rg * wg.double()).sum() + (rn * wn.double()).sum(), x64)
        ep = max(rel_err(gap.detach().cpu().numpy(), rg.detach().cpu().numpy()), rel_err(nfpm.detach().cpu().numpy(), rn.detach().cpu().numpy()))
        egp = rel_err(gp.float().cpu().numpy(), gpref.cpu().numpy())
        ok = ok and ep <= to and egp <= tg
    desc = f"B{B} C{C} {H}x{W} R{R} {meas} {mode} {'nhwc' if cl else 'nchw'} {'bf16' if bf else 'f32'}"
    return ok, desc, (eo, eg, ep, egp), (fv, bv, pv)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 99
    rnd = random.Random(seed)
    dev = torch.device("cuda:0")
    bad = 0
    seen = {}
    for i in range(n):
        ok, desc, errs, vs = one_case(rnd, dev)
        key = vs[0].split("<")[0] + "/" + vs[1].split("<")[0] + (",mfma2" if "mfma2" in vs[1] else (",mfma" if "mfma" in vs[1] else ""))
        seen[key] = seen.get(key, 0) + 1
        if not ok:
            bad += 1
            print("FAIL", desc, ["%.2e" % e for e in errs], vs, flush=True)
        torch.cuda.empty_cache()
    print(f"{n} cases, kernels {seen}, {bad} failed")
