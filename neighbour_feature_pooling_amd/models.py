"""Backbone + NFP head networks for the end-to-end configs.

The reference builds its live models from timm (`create_model(..., pretrained=True, num_classes=0,
global_pool='')`, models/texture_pooling.py:153-208) — timm and its weight downloads do not exist
here, so the three feature extractors are written out in plain PyTorch, randomly initialised, with
the architecture hyper-parameters of the published models (torchvision/timm layouts):

    resnet18              -> [B, 512, H/32, W/32]
    vit_tiny_patch16_224  -> tokens [B, 1+N, 192]  (re-gridded as texture_pooling.py:181-188)
    mobilenetv3_large_100 -> [B, 960, H/32, W/32]

`NFPNet` wires `backbone.forward_features -> nfp_pooling -> fc` exactly as
texture_pooling.py::{ResNet18,ViTTiny,MobileNetV3}_NFPPooling do; the NFP layer inside is the HIP-backed
NFPPooling of this package.  These are ordinary nn.Modules (MIOpen/rocBLAS underneath); only NFP is ours.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .pooling import nfp_pooling


# ---- ResNet-18 ---------------------------------------------------------------------------------
class BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.down = None
        if stride != 1 or cin != cout:
            self.down = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)), inplace=True)
        y = self.bn2(self.conv2(y))
        return F.relu(y + (x if self.down is None else self.down(x)), inplace=True)


class ResNet18Features(nn.Module):
    num_features = 512

    def __init__(self, in_chans=3):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(in_chans, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64),
                                  nn.ReLU(inplace=True), nn.MaxPool2d(3, 2, 1))
        cfg, layers, cin = [(64, 1), (128, 2), (256, 2), (512, 2)], [], 64
        for cout, stride in cfg:
            layers += [BasicBlock(cin, cout, stride), BasicBlock(cout, cout, 1)]
            cin = cout
        self.layers = nn.Sequential(*layers)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward_features(self, x):
        return self.layers(self.stem(x))

    forward = forward_features


# ---- ViT-Tiny / 16 -----------------------------------------------------------------------------
class _Block(nn.Module):
    def __init__(self, dim, heads, mlp_ratio=4.0):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.heads = heads
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.fc1 = nn.Linear(dim, int(dim * mlp_ratio))
        self.fc2 = nn.Linear(int(dim * mlp_ratio), dim)

    def forward(self, x):
        B, T, D = x.shape
        qkv = self.qkv(self.norm1(x)).reshape(B, T, 3, self.heads, D // self.heads).permute(2, 0, 3, 1, 4)
        a = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2])
        x = x + self.proj(a.transpose(1, 2).reshape(B, T, D))
        return x + self.fc2(F.gelu(self.fc1(self.norm2(x))))


class ViTTinyFeatures(nn.Module):
    num_features = 192

    def __init__(self, in_chans=3, img_size=224, patch=16, dim=192, depth=12, heads=3):
        super().__init__()
        self.patch_embed = nn.Conv2d(in_chans, dim, patch, patch)
        n = (img_size // patch) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.randn(1, n + 1, dim) * 0.02)
        self.blocks = nn.Sequential(*[_Block(dim, heads) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim, eps=1e-6)

    def forward_features(self, x):
        x = self.patch_embed(x).flatten(2).transpose(1, 2)
        x = torch.cat([self.cls_token.expand(x.shape[0], -1, -1), x], dim=1) + self.pos_embed
        return self.norm(self.blocks(x))  # [B, 1+N, 192]

    forward = forward_features


# ---- MobileNetV3-Large 1.0 -----------------------------------------------------------------------
class _SE(nn.Module):
    def __init__(self, c):
        super().__init__()
        r = _make_divisible(c // 4, 8)
        self.fc1, self.fc2 = nn.Conv2d(c, r, 1), nn.Conv2d(r, c, 1)

    def forward(self, x):
        s = x.mean((2, 3), keepdim=True)
        return x * F.hardsigmoid(self.fc2(F.relu(self.fc1(s))))


def _make_divisible(v, d):
    n = max(d, int(v + d / 2) // d * d)
    return n + d if n < 0.9 * v else n


class _MBConv(nn.Module):
    def __init__(self, cin, k, exp, cout, se, hs, stride):
        super().__init__()
        act = nn.Hardswish if hs else nn.ReLU
        layers = []
        if exp != cin:
            layers += [nn.Conv2d(cin, exp, 1, bias=False), nn.BatchNorm2d(exp), act(inplace=True)]
        layers += [nn.Conv2d(exp, exp, k, stride, k // 2, groups=exp, bias=False), nn.BatchNorm2d(exp), act(inplace=True)]
        if se:
            layers.append(_SE(exp))
        layers += [nn.Conv2d(exp, cout, 1, bias=False), nn.BatchNorm2d(cout)]
        self.body = nn.Sequential(*layers)
        self.res = stride == 1 and cin == cout

    def forward(self, x):
        return x + self.body(x) if self.res else self.body(x)


class MobileNetV3LargeFeatures(nn.Module):
    num_features = 960
    #      k  exp  out  SE     HS   stride
    CFG = [(3, 16, 16, False, False, 1), (3, 64, 24, False, False, 2), (3, 72, 24, False, False, 1),
           (5, 72, 40, True, False, 2), (5, 120, 40, True, False, 1), (5, 120, 40, True, False, 1),
           (3, 240, 80, False, True, 2), (3, 200, 80, False, True, 1), (3, 184, 80, False, True, 1),
           (3, 184, 80, False, True, 1), (3, 480, 112, True, True, 1), (3, 672, 112, True, True, 1),
           (5, 672, 160, True, True, 2), (5, 960, 160, True, True, 1), (5, 960, 160, True, True, 1)]

    def __init__(self, in_chans=3):
        super().__init__()
        layers = [nn.Conv2d(in_chans, 16, 3, 2, 1, bias=False), nn.BatchNorm2d(16), nn.Hardswish(inplace=True)]
        cin = 16
        for k, exp, cout, se, hs, s in self.CFG:
            layers.append(_MBConv(cin, k, exp, cout, se, hs, s))
            cin = cout
        layers += [nn.Conv2d(cin, 960, 1, bias=False), nn.BatchNorm2d(960), nn.Hardswish(inplace=True)]
        self.features = nn.Sequential(*layers)

    def forward_features(self, x):
        return self.features(x)

    forward = forward_features

    # timm's `features_only=True` taps of mobilenetv3_large_100 (what MobileNetV3_MultiStageNFP reads,
    # texture_pooling.py:222-226): the last layer at each stride — 16 ch @ /2, 24 @ /4, 40 @ /8, 112 @ /16, 960 @ /32
    STAGE_ENDS = (3, 5, 8, 14, 20)          # indices into self.features (stem = 0..2, then the 15 blocks, then the 1x1 head)
    STAGE_CHANNELS = (16, 24, 40, 112, 960)

    def forward_stages(self, x):
        feats = []
        for i, layer in enumerate(self.features):
            x = layer(x)
            if i in self.STAGE_ENDS:
                feats.append(x)
        return feats


BACKBONES = {"resnet18": ResNet18Features, "vit_tiny_patch16_224": ViTTinyFeatures,
             "mobilenetv3_large_100": MobileNetV3LargeFeatures}


# ---- backbone + NFP pooling + classifier ---------------------------------------------------------
class NFPNet(nn.Module):
    """texture_pooling.py::{ResNet18,ViTTiny,MobileNetV3}_NFPPooling (153-208) with a local backbone."""

    def __init__(self, backbone="resnet18", num_classes=10, num_input_channels=3, nfp_layer=None, **backbone_kw):
        super().__init__()
        self.backbone = BACKBONES[backbone](in_chans=num_input_channels, **backbone_kw)
        C = self.backbone.num_features
        params = {"num_ftrs": {backbone: C}, "Model_name": backbone, "Dataset": "synthetic",
                  "num_classes": {"synthetic": num_classes}, "input_size": 7}
        self.pool = nfp_pooling(nfp_layer=nfp_layer, Params=params)
        self.fc = nn.Linear(C, num_classes)

    def forward(self, x):
        feats = self.backbone.forward_features(x)
        if feats.dim() == 3:  # ViT tokens -> grid, texture_pooling.py:181-188
            tok = feats[:, 1:]
            B, N, C = tok.shape
            H = W = int(math.isqrt(N))
            # a VIEW with channels-last strides (the reference's reshape copies into NCHW): the NFP kernels read the
            # token matrix in place, and [pixel][channel] is exactly the matrix-core forward's fragment layout
            feats = tok.transpose(1, 2).unflatten(2, (H, W))
        x = self.pool(feats)
        return self.fc(x.view(x.size(0), -1))


class MultiStageNFPNet(nn.Module):
    """texture_pooling.py::MobileNetV3_MultiStageNFP (211-268) with the local backbone: NFP(cosine, R = 1, padding = 1) on all
    five stage outputs (112x112x16 ... 7x7x960 at 224x224), each averaged to 8 values (249-252), the 40 values projected to
    the head width and multiplied into GAP(conv_head(last map)).  The per-stage `F.adaptive_avg_pool2d(NFP(feat), 1)` is the
    pooled half of the fused tail (functional.nfp_pool): on the GPU the maps above 512 pixels run on the row-band kernels
    of csrc/nfp_tile.h, the small ones on the table kernels."""

    def __init__(self, num_classes=10, num_input_channels=3, head_width=1280):
        super().__init__()
        from .nfp import NFPPooling
        self.backbone = MobileNetV3LargeFeatures(in_chans=num_input_channels)
        self.nfps = nn.ModuleList(NFPPooling(in_channels=c, R=1, measure="cosine", padding=1)
                                  for c in MobileNetV3LargeFeatures.STAGE_CHANNELS)
        self.conv_head = nn.Sequential(nn.Conv2d(960, head_width, 1), nn.Hardswish(inplace=True))   # timm: conv_head + act2
        self.nfp_proj = nn.Linear(8 * len(self.nfps), head_width)
        self.fc = nn.Linear(head_width, num_classes)

    def forward(self, x):
        from .functional import nfp_pooled
        feats = self.backbone.forward_stages(x)
        # texture_pooling.py:251-252: F.adaptive_avg_pool2d(nfp(feat), 1) per stage — the pooled maps alone, no GAP(feat)
        v = torch.cat([nfp_pooled(f, layer.config).to(f.dtype) for f, layer in zip(feats, self.nfps)], dim=1)   # [B, 40]
        head = self.conv_head(feats[-1]).mean((2, 3))
        return self.fc(head * self.nfp_proj(v))
