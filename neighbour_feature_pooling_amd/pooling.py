"""nfp_pooling — drop-in for models/NFP_Pooling.py::nfp_pooling (NFP_Pooling.py:5-36).

    y = GAP(x) * Linear_{N->C}( GAP( NFP(x) ) )        [B,C,H,W] -> [B,C]

Same constructor (`nfp_layer`, `Params` dict) and attributes; the NFP layer is the
HIP-backed NFPPooling of this package.
"""
import torch.nn as nn
import torch.nn.functional as F

from .functional import nfp_pool
from .nfp import NFPPooling


class nfp_pooling(nn.Module):
    def __init__(self, nfp_layer=None, Params=None):
        super().__init__()
        # NFP_Pooling.py:9 computes this only when nfp_layer is None, so the reference raises
        # UnboundLocalError at line 23 when given BOTH a layer and Params; here that combination works.
        dense_feature_dim = Params["num_ftrs"][Params["Model_name"]] if Params else 2048
        if nfp_layer is None:
            # NFP_Pooling.py:10-16 — R=1, cosine, padding=1 are hard-coded there
            nfp_layer = NFPPooling(in_channels=dense_feature_dim, R=1, measure='cosine', padding=1,
                                   input_size=Params.get('input_size', 7) if Params else 7)
        self.nfp_layer = nfp_layer
        self.model_name = Params["Model_name"] if Params is not None else None
        self.dataset = Params["Dataset"] if Params is not None else None
        self.num_classes = Params["num_classes"][self.dataset] if Params is not None else None
        self.feature_extraction = Params.get('feature_extraction') if Params is not None else None
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        # NFP_Pooling.py:23 — the projection exists only when Params is given
        self.nfp_proj = nn.Linear(self.nfp_layer.out_channels, dense_feature_dim) if Params else None

    def forward(self, x):
        if isinstance(self.nfp_layer, NFPPooling) and x.dim() == 4 and x.shape[1] == self.nfp_layer.in_channels:
            # both means come out of ONE pass over x on the GPU (nfp_pool_forward / nfp_pool_backward)
            x_avg, x_nfp = nfp_pool(x, self.nfp_layer.config)
            x_avg, x_nfp = x_avg.to(x.dtype), x_nfp.to(x.dtype)
        else:
            x_avg = self.avgpool(x).flatten(1)                               # NFP_Pooling.py:27
            x_nfp = F.adaptive_avg_pool2d(self.nfp_layer(x), 1).flatten(1)   # NFP_Pooling.py:29-31
        if self.nfp_proj is not None:
            x_nfp = self.nfp_proj(x_nfp)                                     # NFP_Pooling.py:32-33
        return x_avg * x_nfp                                                 # NFP_Pooling.py:35
