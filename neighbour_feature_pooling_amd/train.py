"""Data-parallel train step for backbone+NFP networks (BASELINE.json configs 3-5).

The reference trains on one device through Lightning (Trainer(devices=1), demo.py:404-412).  This
harness reproduces its optimisation recipe — CrossEntropyLoss(label_smoothing=0.05)
(Lightning_Wrapper.py:35), Adam(lr=1e-4) (Lightning_Wrapper.py:69-79, demo.py:461) — and adds what
the reference lacks: one process per GPU, DistributedDataParallel with bucketed gradient all-reduce
over RCCL/xGMI (backend "nccl" on ROCm; "gloo" on CPU for tests).  NFP itself has no parameters, so
it contributes nothing to the all-reduce and needs no find_unused_parameters.

    python -m neighbour_feature_pooling_amd.train --gpus 8 --model resnet18 --batch 256 --image 224 --steps 20
(starts the 8 ranks itself; or under `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 ...`)
"""
import argparse
import json
import os
import time

import torch
import torch.distributed as dist
import torch.nn as nn
from torch.nn.parallel import DistributedDataParallel as DDP

from .models import MultiStageNFPNet, NFPNet
from .nfp import NFPPooling


def build(model="resnet18", num_classes=10, in_chans=3, image=224, nfp=None, device="cpu", dtype=torch.float32):
    if model == "mobilenetv3_multistage":      # texture_pooling.py:211-268: NFP on all five stage outputs
        return MultiStageNFPNet(num_classes=num_classes, num_input_channels=in_chans).to(device=device, dtype=dtype)
    kw = {"img_size": image} if model.startswith("vit") else {}
    net = NFPNet(model, num_classes=num_classes, num_input_channels=in_chans, nfp_layer=nfp, **kw)
    return net.to(device=device, dtype=dtype)


def make_step(net, lr=1e-4):
    """Returns step(x, y) -> loss tensor: forward, CE(ls=0.05), backward (DDP all-reduces here), Adam."""
    crit = nn.CrossEntropyLoss(label_smoothing=0.05)
    opt = torch.optim.Adam(net.parameters(), lr=lr)

    def step(x, y):
        opt.zero_grad(set_to_none=True)
        loss = crit(net(x).float(), y)
        loss.backward()
        opt.step()
        return loss.detach()

    return step, opt


def synthetic_batch(batch, in_chans, image, num_classes, device, dtype, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(batch, in_chans, image, image, generator=g).to(device=device, dtype=dtype)
    y = torch.randint(0, num_classes, (batch,), generator=g).to(device)
    return x, y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="resnet18", choices=["resnet18", "vit_tiny_patch16_224", "mobilenetv3_large_100", "mobilenetv3_multistage"])
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch")
    ap.add_argument("--image", type=int, default=224)
    ap.add_argument("--in-chans", type=int, default=3)
    ap.add_argument("--classes", type=int, default=10)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32")
    ap.add_argument("--nfp-radius", type=int, default=1)
    ap.add_argument("--nfp-measure", default="cosine")
    ap.add_argument("--channels-last", action="store_true", help="NHWC activations and weights: MIOpen's NHWC "
                    "convolutions, and NFP reads the channels-last feature map in place")
    ap.add_argument("--autotune", action="store_true", help="torch.backends.cudnn.benchmark: MIOpen find mode")
    ap.add_argument("--cpu", action="store_true", help="gloo on CPU (plumbing test)")
    ap.add_argument("--backend", default=None, help="override the process-group backend (gloo lets several "
                    "ranks share one GPU for rehearsal; RCCL refuses that)")
    ap.add_argument("--gpus", type=int, default=1, help="N > 1 without a launcher: start N ranks (one per GPU)")
    ap.add_argument("--print-launch", action="store_true", help="with --gpus N: print the launch command and exit")
    a = ap.parse_args()

    from . import parallel
    if a.gpus > 1 and not parallel.launched_by_torchrun():
        import sys
        argv = [v for v in sys.argv[1:] if v != "--print-launch"]
        raise SystemExit(parallel.self_launch(a.gpus, ("-m", __spec__.name if __spec__ else "neighbour_feature_pooling_amd.train"),
                                              argv, backend=a.backend or ("gloo" if a.cpu else "nccl"),
                                              dry_run=a.print_launch))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if parallel.launched_by_torchrun() and a.gpus != 1 and a.gpus != world:   # (as bench.py: ADVICE r3)
        raise SystemExit(f"--gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = a.backend or ("gloo" if a.cpu else "nccl")
    if not a.cpu and backend == "gloo":
        local %= torch.cuda.device_count()
    dev = torch.device("cpu") if a.cpu else torch.device("cuda", local)
    if not a.cpu:
        torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend)
    dtype = torch.float32 if a.dtype == "f32" else torch.bfloat16
    torch.manual_seed(0)  # identical initial weights on every rank (DDP also broadcasts rank 0's)
    C = {"resnet18": 512, "vit_tiny_patch16_224": 192, "mobilenetv3_large_100": 960, "mobilenetv3_multistage": 960}[a.model]
    ctor = dict(R=a.nfp_radius, measure=a.nfp_measure, padding=a.nfp_radius)
    if a.nfp_measure == "norm":
        ctor["p"] = 2
    net = build(a.model, a.classes, a.in_chans, a.image, NFPPooling(C, **ctor), dev, dtype)
    torch.backends.cudnn.benchmark = bool(a.autotune)
    if a.channels_last:
        net = net.to(memory_format=torch.channels_last)
    if world > 1:
        net = DDP(net, device_ids=None if a.cpu else [local], gradient_as_bucket_view=True)
    step, _ = make_step(net)
    x, y = synthetic_batch(a.batch, a.in_chans, a.image, a.classes, dev, dtype, seed=1000 + rank)
    if a.channels_last:
        x = x.contiguous(memory_format=torch.channels_last)
    for _ in range(a.warmup):
        step(x, y)
    if not a.cpu:
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step(x, y)
    if not a.cpu:
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = (time.perf_counter() - t0) / a.steps
    if rank == 0:
        print(json.dumps({"model": a.model, "n_gpus": world, "batch_per_gpu": a.batch, "image": a.image,
                          "dtype": a.dtype, "ms_per_step": round(dt * 1e3, 3),
                          "images_per_s": round(world * a.batch / dt, 1), "loss": round(float(loss), 4),
                          "nfp": f"{a.nfp_measure} R={a.nfp_radius}",
                          "layout": "channels_last" if a.channels_last else "nchw", "autotune": bool(a.autotune)}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
