"""Backbone + NFP networks: shapes, the train recipe, and the DDP step on two gloo
ranks — averaged gradients must equal the single-process gradients on the concatenated batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from neighbour_feature_pooling_amd import NFPPooling
from neighbour_feature_pooling_amd.models import (BACKBONES, MobileNetV3LargeFeatures, NFPNet, ResNet18Features,
                                                  ViTTinyFeatures)
from neighbour_feature_pooling_amd.train import build, make_step, synthetic_batch


def test_backbone_output_shapes_and_sizes():
    with torch.no_grad():
        assert ResNet18Features(13)(torch.randn(2, 13, 64, 64)).shape == (2, 512, 2, 2)       # config 3
        assert ResNet18Features(3)(torch.randn(1, 3, 96, 96)).shape == (1, 512, 3, 3)
        assert ViTTinyFeatures(3, img_size=64)(torch.randn(2, 3, 64, 64)).shape == (2, 17, 192)
        assert MobileNetV3LargeFeatures(3)(torch.randn(1, 3, 64, 64)).shape == (1, 960, 2, 2)
    n = lambda m: sum(p.numel() for p in m.parameters())
    assert 11.1e6 < n(ResNet18Features()) < 11.3e6            # 11.18 M (torchvision/timm resnet18 trunk)
    assert 5.4e6 < n(ViTTinyFeatures()) < 5.6e6               # 5.52 M  (vit_tiny_patch16_224 trunk)
    assert 2.9e6 < n(MobileNetV3LargeFeatures()) < 3.1e6      # 2.97 M  (mobilenetv3_large_100 trunk)


def test_multistage_net_mirrors_the_reference_model():
    """texture_pooling.py::MobileNetV3_MultiStageNFP (211-268): five stage outputs with 16 / 24 / 40 / 112 / 960 channels at
    strides 2 ... 32 (timm's features_only taps), one NFP(cosine, R = 1, padding = 1) each, 5 x 8 pooled values -> Linear(40,
    1280), times GAP(conv_head(last map)), classifier."""
    from neighbour_feature_pooling_amd.models import MultiStageNFPNet
    with torch.no_grad():
        feats = MobileNetV3LargeFeatures(3).forward_stages(torch.randn(1, 3, 224, 224))
    assert [tuple(f.shape[1:]) for f in feats] == [(16, 112, 112), (24, 56, 56), (40, 28, 28), (112, 14, 14), (960, 7, 7)]
    net = MultiStageNFPNet(num_classes=7)
    assert [l.in_channels for l in net.nfps] == [16, 24, 40, 112, 960] and all(l.out_channels == 8 for l in net.nfps)
    assert (net.nfp_proj.in_features, net.nfp_proj.out_features, net.fc.in_features) == (40, 1280, 1280)
    assert not [n for n, p in net.named_parameters() if "nfps." in n and p.requires_grad]   # NFP has no trainable weights
    step, _ = make_step(net)
    x, y = synthetic_batch(2, 3, 64, 7, "cpu", torch.float32, 5)
    l0 = step(x, y)
    assert torch.isfinite(l0)


@pytest.mark.parametrize("name,image,nfp", [
    ("resnet18", 64, None),
    ("vit_tiny_patch16_224", 64, dict(R=2, measure="norm", p=2, padding=2)),  # config 5 geometry: k=5 L2
    ("mobilenetv3_large_100", 64, None),
])
def test_nfpnet_forward_backward(name, image, nfp):
    C = BACKBONES[name].num_features
    layer = NFPPooling(C, **nfp) if nfp else None
    net = build(name, num_classes=7, in_chans=3, image=image, nfp=layer)
    assert net.pool.nfp_proj.in_features == (24 if nfp else 8) and net.pool.nfp_proj.out_features == C
    step, _ = make_step(net)
    x, y = synthetic_batch(4, 3, image, 7, "cpu", torch.float32, 1)
    l0 = step(x, y)
    for _ in range(3):
        l1 = step(x, y)
    assert torch.isfinite(l0) and l1 < l0  # Adam(1e-4) + CE(ls=0.05) makes progress on a fixed batch


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _ddp_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torch.nn.parallel import DistributedDataParallel as DDP
        torch.manual_seed(0)
        net = build("resnet18", num_classes=5, in_chans=3, image=64)
        net.eval()  # BatchNorm in eval mode: batch statistics would legitimately differ per shard
        ddp = DDP(net)
        x, y = synthetic_batch(4, 3, 64, 5, "cpu", torch.float32, seed=77)  # same global batch on both ranks
        lo, hi = rank * 2, rank * 2 + 2
        loss = torch.nn.CrossEntropyLoss(label_smoothing=0.05)(ddp(x[lo:hi]), y[lo:hi])
        loss.backward()  # DDP all-reduces (mean) here
        if rank == 0:
            g = {k: p.grad.numpy() for k, p in net.named_parameters() if p.grad is not None}
            np.savez(os.path.join(out_dir, "ddp.npz"), **{k.replace(".", "/"): v for k, v in g.items()})
    finally:
        dist.destroy_process_group()


def test_ddp_gradients_equal_single_process(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_ddp_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "ddp.npz")
    torch.manual_seed(0)
    net = build("resnet18", num_classes=5, in_chans=3, image=64)
    net.eval()
    x, y = synthetic_batch(4, 3, 64, 5, "cpu", torch.float32, seed=77)
    torch.nn.CrossEntropyLoss(label_smoothing=0.05)(net(x), y).backward()
    n = 0
    for k, p in net.named_parameters():
        ref = p.grad.numpy()
        g = got[k.replace(".", "/")]
        assert np.max(np.abs(g - ref)) <= 1e-5 * max(1e-3, np.max(np.abs(ref))) + 1e-7, k
        n += 1
    assert n > 60 and "pool/nfp_proj/weight" in got.files
    assert not any("nfp_layer" in k for k in got.files)  # NFP has no parameters -> nothing to all-reduce
