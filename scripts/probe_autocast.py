#!/usr/bin/env python3
"""SURVEY a7 / VERDICT r2 item 7c: what the reference's op sequence returns on the GPU under autocast and for float16
inputs (the policy of F.cosine_similarity / linalg.norm / conv2d could not be checked without a GPU)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle.unfold_torch import UnfoldNFP
rows = []
for meas, kw in (("cosine", {}), ("norm", {"p": 2})):
    m = UnfoldNFP(64, R=1, measure=meas, padding=1, **kw)
    m.w_comp, m.w_centre = m.w_comp.cuda(), m.w_centre.cuda()
    x32 = torch.randn(2, 64, 7, 7, device="cuda")
    ref = m(x32)
    for ac in (None, torch.bfloat16, torch.float16):
        for xdt in (torch.float32, torch.bfloat16, torch.float16):
            mm = UnfoldNFP(64, R=1, measure=meas, padding=1, **kw)
            mm.w_comp, mm.w_centre = mm.w_comp.cuda(), mm.w_centre.cuda()
            if ac is None:      # module.to(dtype) as a caller would
                mm.w_comp, mm.w_centre = mm.w_comp.to(xdt), mm.w_centre.to(xdt)
            x = x32.to(xdt).requires_grad_(True)
            try:
                if ac is None:
                    out = mm(x)
                else:
                    with torch.autocast("cuda", ac):
                        out = mm(x)
                (gx,) = torch.autograd.grad(out.float().sum(), x)
                err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
                rows.append(dict(measure=meas, autocast=str(ac), x=str(xdt), out=str(out.dtype), grad=str(gx.dtype), rel_err_vs_f32=round(err, 5)))
            except Exception as e:
                rows.append(dict(measure=meas, autocast=str(ac), x=str(xdt), error=f"{type(e).__name__}: {str(e)[:80]}"))
            print(json.dumps(rows[-1]), flush=True)
