"""nfp_pooling — drop-in for models/NFP_Pooling.py::nfp_pooling (NFP_Pooling.py:5-36).

    y = GAP(x) * Linear_{N->C}( GAP( NFP(x) ) )        [B,C,H,W] -> [B,C]

Same constructor (`nfp_layer`, `Params` dict) and public attributes.  On the GPU both global
averages come out of ONE pass over x (nfp_pool_forward / nfp_pool_backward in include/nfp.h).
"""
import torch.nn as nn
import torch.nn.functional as F

from .functional import nfp_pool
from .nfp import NFPPooling


def _from_params(params):
    """The five things the reference pulls out of its `Params` dict (NFP_Pooling.py:9,15,18-21)."""
    if params is None:
        return dict(dim=2048, input_size=7, model=None, dataset=None, classes=None, feature_extraction=None)
    model, dataset = params["Model_name"], params["Dataset"]
    return dict(dim=params["num_ftrs"][model], input_size=params.get("input_size", 7), model=model,
                dataset=dataset, classes=params["num_classes"][dataset],
                feature_extraction=params.get("feature_extraction"))


class nfp_pooling(nn.Module):
    def __init__(self, nfp_layer=None, Params=None):
        super().__init__()
        cfg = _from_params(Params)
        # NFP_Pooling.py:9 computes the feature width only when nfp_layer is None, so the reference raises
        # UnboundLocalError at its line 23 when given BOTH a layer and Params; here that combination works.
        if nfp_layer is None:
            # R=1, cosine, padding=1 are hard-coded in the reference (NFP_Pooling.py:10-16)
            nfp_layer = NFPPooling(in_channels=cfg["dim"], R=1, measure='cosine', padding=1,
                                   input_size=cfg["input_size"])
        self.nfp_layer = nfp_layer
        self.model_name, self.dataset = cfg["model"], cfg["dataset"]
        self.num_classes, self.feature_extraction = cfg["classes"], cfg["feature_extraction"]
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        # the N -> C projection exists only when Params is given (NFP_Pooling.py:23)
        self.nfp_proj = nn.Linear(self.nfp_layer.out_channels, cfg["dim"]) if Params else None

    def forward(self, x):
        layer = self.nfp_layer
        if isinstance(layer, NFPPooling) and x.dim() == 4 and x.shape[1] == layer.in_channels:
            pooled_x, pooled_nfp = nfp_pool(x, layer.config)                  # NFP_Pooling.py:27-31, fused
            pooled_x, pooled_nfp = pooled_x.to(x.dtype), pooled_nfp.to(x.dtype)
        else:
            pooled_x = self.avgpool(x).flatten(1)
            pooled_nfp = F.adaptive_avg_pool2d(layer(x), 1).flatten(1)
        if self.nfp_proj is not None:
            pooled_nfp = self.nfp_proj(pooled_nfp)                            # NFP_Pooling.py:32-33
        return pooled_x * pooled_nfp                                          # NFP_Pooling.py:35
