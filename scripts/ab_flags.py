#!/usr/bin/env python3
"""A/B kernel variants on ONE device.  Two steps, so that no GPU time is spent compiling:

  here (CPU container):  python scripts/ab_flags.py --build "" "-DNFP_BWD_THREADS=448" ...
      builds one library per flag set, in parallel, into neighbour_feature_pooling_amd/ab/ (git-ignored,
      but it travels to the GPU box with the snapshot) and writes ab/manifest.json
  on the GPU box:        python scripts/ab_flags.py --run
      times every library of the manifest round-robin (2 rounds) in fresh subprocesses

Optional env for --run: AB_SHAPE="64,512,7,1,cosine[,bf16][,nhwc]"  (B,C,S,R,measure); AB_COLD=1 adds the rotating-set leg
"""
import concurrent.futures as cf
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neighbour_feature_pooling_amd import build

AB = os.path.join(ROOT, "neighbour_feature_pooling_amd", "ab")
MANIFEST = os.path.join(AB, "manifest.json")


def build_all(variants):
    os.makedirs(AB, exist_ok=True)
    for f in os.listdir(AB):
        os.remove(os.path.join(AB, f))

    def one(i_fl):
        i, fl = i_fl
        lib = os.path.join(AB, f"libnfp_ab_{i}.so")
        return build.compile_hip(lib, ["-w"] + fl.split())

    with cf.ThreadPoolExecutor(max_workers=3) as ex:
        libs = list(ex.map(one, enumerate(variants)))
    json.dump([{"flags": fl, "lib": os.path.relpath(lib, ROOT)} for fl, lib in zip(variants, libs)],
              open(MANIFEST, "w"), indent=1)
    print(f"built {len(libs)} variants into {AB}")


CODE = """
import os, sys; sys.path.insert(0, {root!r})
os.environ["NFP_PY_NODES"] = "1"   # (the C++ nodes link the product library; the Python nodes call whatever _abi loads)
import torch
from neighbour_feature_pooling_amd import _abi
_abi.LIB_PATH = {lib!r}
from neighbour_feature_pooling_amd import NFPPooling
from bench import time_kernel_graph
ctor = dict(R={R}, measure={meas!r}, padding={R})
if {meas!r} == 'norm': ctor['p'] = 2
m = NFPPooling({C}, **ctor)
dt = torch.bfloat16 if {bf16} else torch.float32
x = torch.randn({B}, {C}, {S}, {S}, device='cuda').to(dt)
if {nhwc}: x = x.contiguous(memory_format=torch.channels_last)
x.requires_grad_(True)
go = torch.randn({B}, m.out_channels, {S}, {S}, device='cuda').to(dt)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    out = m(x)
    fv = _abi.load().nfp_last_variant().decode()
    torch.autograd.grad(out, x, go, retain_graph=True)
    torch.cuda.synchronize()
    bv = _abi.load().nfp_last_variant().decode()
    tf = time_kernel_graph(lambda: m(x), 50, s)
    tb = time_kernel_graph(lambda: torch.autograd.grad(out, x, go, retain_graph=True), 50, s)
if os.environ.get("AB_COLD"):   # rotating sets behind a cache flush (bench.py's cold protocol) instead of one resident set
    import bench
    a = bench.parse(["--batch", "{B}", "--channels", "{C}", "--size", "{S}", "--radius", "{R}", "--measure", {meas!r},
                     "--dtype", "bf16" if {bf16} else "f32", "--layout", "nhwc" if {nhwc} else "nchw"])
    w = bench.Workload(a, torch.device("cuda", 0), 0)
    with torch.cuda.stream(s):
        cf_, cb_ = w.kernel_times(s)
        _, cbi = w.kernel_times(s, isolated_backward=True)
    print(f"[{{{tag!r}:44s}}] cold: fwd {{cf_:6.2f}} us   bwd in step {{cb_:6.2f}} us   bwd alone {{cbi:6.2f}} us")
print(f"[{{{tag!r}:44s}}] fwd {{tf:6.2f}} us   bwd {{tb:6.2f}} us   sum {{tf+tb:6.2f}}   {{fv}} / {{bv}}")
"""


def run_all():
    items = json.load(open(MANIFEST))
    shp = (os.environ.get("AB_SHAPE") or "64,512,7,1,cosine").split(",")
    B, C, S, R, meas = shp[:5]
    for rnd in range(2):
        for it in items:
            subprocess.check_call([sys.executable, "-c", CODE.format(
                root=ROOT, lib=os.path.join(ROOT, it["lib"]), R=R, meas=meas, C=C, B=B, S=S,
                bf16="bf16" in shp[5:], nhwc="nhwc" in shp[5:], tag=it["flags"] or "(default)")])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--build":
        build_all(sys.argv[2:] or [""])
    elif len(sys.argv) > 1 and sys.argv[1] == "--run":
        run_all()
    else:
        sys.exit(__doc__)
