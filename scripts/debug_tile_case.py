#!/usr/bin/env python3
"""Debug helper: one case of tests/test_gpu_tile.py, where the outputs differ from the float64 formulation / between runs.
usage: python scripts/debug_tile_case.py B C H W R measure mode nchw|nhwc"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from test_gpu_tile import _run
B, C, H, W, R = [int(v) for v in sys.argv[1:6]]
meas, mode, lay = sys.argv[6:9]
dev = torch.device("cuda:0")
out, gx, ref, gref, fv, bv = None, None, None, None, None, None
try:
    out, gx, ref, gref, fv, bv = _run(B, C, H, W, R, meas, mode, dev, channels_last=(lay == "nhwc"))
except AssertionError as e:
    print("assert:", str(e)[:80])
from neighbour_feature_pooling_amd import NFPPooling, _abi
from neighbour_feature_pooling_amd._host import nfp_host
from neighbour_feature_pooling_amd.synth import feature_map
ctor = dict(R=R, measure=meas, padding=R, padding_mode=mode)
if meas == "norm":
    ctor["p"] = 2
m = NFPPooling(C, **ctor)
x = torch.from_numpy(feature_map((B, C, H, W), 5 * H + W + C)).to(dev)
if lay == "nhwc":
    x = x.contiguous(memory_format=torch.channels_last)
x.requires_grad_(True)
L = _abi.load()
outs = [m(x).detach().cpu().numpy() for _ in range(3)]
print(L.nfp_last_variant().decode())
ref = nfp_host(x.detach().double().contiguous(), m.config).cpu().numpy()
for i, o in enumerate(outs):
    bad = np.argwhere(np.abs(o - ref) > 1e-4 * (1 + np.abs(ref)))
    print("run", i, "bad elements", len(bad), "of", o.size)
    if len(bad):
        print("  b:", np.unique(bad[:, 0])[:10], "n:", np.unique(bad[:, 1]), "y:", np.unique(bad[:, 2])[:40], "x:", np.unique(bad[:, 3])[:40])
        for k in bad[:5]:
            print("   ", k, o[tuple(k)], ref[tuple(k)])
