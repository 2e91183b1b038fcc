#!/usr/bin/env python3
"""Randomised geometry stress of the row-band kernels (maps above 512 pixels) against the float64 torch formulation:
plain and pooled, both layouts, f32 / bf16, every hot measure, every padding mode.  usage: python scripts/stress_tile.py [n] [seed]"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from neighbour_feature_pooling_amd import NFPPooling, _abi
from neighbour_feature_pooling_amd._host import nfp_host
from neighbour_feature_pooling_amd.functional import nfp_pool, nfp_pool_fused_ok
from neighbour_feature_pooling_amd.synth import feature_map


def rel_err(a, b):
    """max |a - b| / max |b| over the elements where the reference is a number; inf when the NaN patterns differ
    (RMSE at distance 0 — a pixel and its replicated copy — has no subgradient: torch gives NaN, and so must we)."""
    a, b = a.astype(np.float64), b.astype(np.float64)
    na, nb = np.isnan(a), np.isnan(b)
    if not np.array_equal(na, nb):
        return float("inf")
    if nb.all():
        return 0.0
    return float(np.max(np.abs(a[~nb] - b[~nb])) / max(1e-30, np.max(np.abs(b[~nb]))))


def one_case(rnd, dev):
    R = rnd.choice([1, 1, 1, 2])
    wide = os.environ.get("STRESS_WIDE") == "1"   # (rows beyond the row-band kernels' envelope too: whoever serves them)
    W = rnd.randint(2 * R + 2, (150 if R == 2 else 270) if wide else (120 if R == 2 else 200))
    H = rnd.randint(max(R + 1, 513 // W + 1), max(R + 2, min(160, 40000 // W)))
    C = 4 * rnd.randint(1, rnd.choice([4, 12, 40]))
    B = rnd.choice([1, 2, 3, 7, 40, 130]) if H * W * C < 400000 else rnd.choice([1, 2, 3])
    meas = rnd.choice(["cosine", "cosine", "norm", "dot", "gfc", "rmse"])
    mode = rnd.choice(["reflect", "zeros", "replicate"])
    if os.environ.get("STRESS_SYM") == "1":   # the five measures of csrc/nfp_measures.h::kSymTerm
        meas = rnd.choice(["geman", "canberra", "hellinger", "squaredchord", "chisquared1", "jeffrey"])
        if meas == "hellinger" and (mode == "replicate" or R == 2):   # (a pixel against its own copy: NaN, and how far it travels differs — DESIGN.md section 7)
            mode = "zeros"
    cl = rnd.random() < 0.5
    bf = rnd.random() < 0.3
    sim = rnd.random() < 0.8
    ctor = dict(R=R, measure=meas, padding=R, padding_mode=mode, similarity=sim)
    if meas == "norm":
        ctor["p"] = 2
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
    desc = f"B{B} C{C} {H}x{W} R{R} {meas} {mode} {'nhwc' if cl else 'nchw'} {'bf16' if bf else 'f32'} sim={sim}"
    return ok, desc, (eo, eg, ep, egp), (fv, bv, pv)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2026
    rnd = random.Random(seed)
    dev = torch.device("cuda:0")
    bad = tiles = 0
    for i in range(n):
        ok, desc, errs, vs = one_case(rnd, dev)
        tiles += vs[0].startswith("fwd_tile")
        if not ok:
            bad += 1
            print("FAIL", desc, ["%.2e" % e for e in errs], vs, flush=True)
    print(f"{n} cases, {tiles} on the row-band kernels, {bad} failed")
